// Host interface of the dense-block solver path (ascent_dense.hip), used by the C ABI in ascent_solver.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include "ascent.h"

namespace ascent {

// bytes of workspace for `batch` problems on K = n_nodes-1 steps
size_t dense_ws_bytes(int K, long batch);
size_t dense_pcr_ws_bytes(int K, long batch);     // with the PCR variant's block images

// Solve (all pointers device pointers; blob / traj layouts of include/ascent.h).  scheme 0/1/2, terminal 0/1.
// The host reads one counter per burst of four rounds.  pcr != 0: the Newton systems are solved by parallel cyclic
// reduction over the nodes (workspace dense_pcr_ws_bytes) instead of the serial Riccati recursion.  Returns ASCENT_OK / ASCENT_E_HIP / ASCENT_E_NOTERM.
int dense_run(const ascent_params *dp, long batch, int K, int scheme, int terminal, double *ws, const double *dguess, int warm,
              int max_iter, double tol, double mu0, double *dtraj, double *dtf, int *dstatus, int *diters, double *dblob,
              hipStream_t stream, char *err, size_t errlen, int pcr = 0, int move_penalty = 0);
// move_penalty != 0: the objective gains params.dcost * sum_k |u_k - u_{k-1}| (the reference's MV DCOST, LO:99; both Newton solvers).

// Parity surface: one Newton step at a caller-supplied iterate (dinertia receives 0 / nonzero), and/or the dense stage
// records of every step as d_eval leaves them, drecords[batch][K][6][64] (grids Ja, Jb, Haa, Hab, Hbb and the vector grid).
int dense_probe(const ascent_params *dp, long batch, int K, int scheme, int terminal, double *ws, const double *diterate,
                const double *dmu, const double *ddw, bool step_too, double *dstep, int *dinertia, double *drecords,
                hipStream_t stream, char *err, size_t errlen, int pcr = 0, int move_penalty = 0);

// Kepler-exact coast arc from every NLP's burnout state (scaled x, y, xdot, ydot: dstate4[4][batch]) to the next apoapsis:
// dcoast[4][nc+1][batch], dtheta2[batch] (duration / T_scale), dapsides[2][batch] (periapsis, apoapsis altitude in m).
int coast_run(const ascent_params *dp, long batch, const double *dstate4, int nc, double *dcoast, double *dtheta2,
              double *dapsides, hipStream_t stream, char *err, size_t errlen);

}  // namespace ascent
