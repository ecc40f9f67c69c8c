// Host interface of the split pipeline (ascent_pipeline.hip), used by the C ABI in ascent_solver.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include "ascent.h"

namespace ascent {

struct PipelineStats {
  int launches = 0;
};

// bytes of workspace the pipeline needs for `batch` problems on K = n_nodes-1 steps
size_t pipeline_ws_bytes(int K, long batch);

// Solve; all pointers are device pointers.  Synchronises `stream` once per interior-point iteration
// (it reads three counters to steer the lanes' state machines).  Returns ASCENT_OK or ASCENT_E_HIP.
int pipeline_run(const ascent_params *dp, long batch, int K, int scheme, int form, double *ws, const double *dguess, int warm,
                 int max_iter, double tol, double mu0, double *dtraj, double *dtf, int *dstatus, int *diters,
                 double *dblob, hipStream_t stream, PipelineStats *stats, char *err, size_t errlen);

// One round of the pipeline's kernels at a caller-supplied iterate (the parity surface behind ascent_kkt_step /
// ascent_eval_nodes): mu and delta_w per problem; `wide` picks the 16-lane sweeps.  With step_too == false only the
// node evaluation (q_trial_eval) runs.  Output pointers may be null.  All device pointers; asynchronous on `stream`.
int pipeline_probe(const ascent_params *dp, long batch, int K, int scheme, int form, double *ws, const double *diterate,
                   const double *dmu, const double *ddw, bool wide, bool step_too, double *dstep, int *dinertia,
                   double *ddefects, double *djac, double *dhess, hipStream_t stream, char *err, size_t errlen);

}  // namespace ascent
