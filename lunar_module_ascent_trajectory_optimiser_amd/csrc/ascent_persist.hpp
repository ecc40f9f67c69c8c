// Host interface of the persistent solver kernel (ascent_persist.hip), used by the C ABI in ascent_solver.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include "ascent.h"

namespace ascent {

// Workspace of one grid level (a multiple of 256 bytes); mp: with the l1 move penalty (five more rows per node and iterate).
size_t persist_ws_bytes(int K, long batch, int mp);

// The whole nested iteration inside the kernel's own layout: levels[0] = the requested grid (nodes), finest first; coarse levels
// are solved to tol_coarse; a level warm-started from the coarsest grid begins at mu_first, later ones at mu_next.  term = 2:
// ascent_opts.terminal 2 (burnout anywhere on the ellipse; on the UNtransformed parameters), 0 otherwise.  Two regions
// alternate between the levels: region 0 at the start of the workspace, region 1 at persist_region1_offset; the total is
// persist_ws_bytes_nested (tests/test_host.py checks that the last NLP of region 1 ends inside it).
size_t persist_region1_offset(const int *levels, long batch, int mp);
size_t persist_ws_bytes_nested(const int *levels, int nlev, long batch, int mp);
size_t persist_level_bytes_used(int K, long batch, int mp);      // bytes the kernels of one level touch from the start of its region
int persist_run_nested(const ascent_params *dp, long batch, int scheme, int form, int mp, int term, const int *levels, int nlev, double *ws, const double *dguess, int warm,
                       int max_iter, double tol, double tol_coarse, double mu0, double mu_first, double mu_next, double *dtraj,
                       double *dtf, int *dstatus, int *diters, double *dblob, hipStream_t stream, char *err, size_t errlen);

// One interior-point round of the same kernel at a caller-supplied iterate, mu and delta_w (parity surface): the Newton step in
// the blob layout, inertia[p] = 1 where the factorisation was refused.
int persist_probe(const ascent_params *dp, long batch, int scheme, int form, int mp, int term, int K, double *ws, const double *diterate, const double *dmu, const double *ddw,
                  double *dstep, int *dinertia, hipStream_t stream, char *err, size_t errlen);

// The node rows (defects, Jacobian and Hessian blocks in the layout of ascent_eval_nodes) that one round of the same kernel
// leaves in LDS for its factorisation sweep, copied out instead of swept.  dzero: `batch` zeros on the device.
int persist_probe_rows(const ascent_params *dp, long batch, int scheme, int form, int K, double *ws, const double *diterate, const double *dzero,
                       double *ddefects, double *djac, double *dhess, hipStream_t stream, char *err, size_t errlen);

// Hermite-Simpson (scheme 2) in the same layout: ascent_hs.hip.  One launch = one grid level of the batch; the node arrays are padded to
// chunks of hs_chunk_nodes(wide) nodes (12: four NLPs per wavefront, 48: one).
void hs_launch_solve(long batch, hipStream_t stream, const ascent_params *dp, int K, int Kp, int nch, int term, int wide, double *w, int max_iter, double tol);
int hs_chunk_nodes(int wide);

}  // namespace ascent
