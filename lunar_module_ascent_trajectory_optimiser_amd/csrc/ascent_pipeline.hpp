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

}  // namespace ascent
