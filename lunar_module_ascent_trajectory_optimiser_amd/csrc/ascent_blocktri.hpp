// Host interface of the generic bordered block-tridiagonal solver (ascent_blocktri.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include "ascent.h"

namespace ascent {

size_t blocktri_ws_bytes(int n, long batch, int algo);

// Device pointers.  dY [batch][n][bs][1+nb] receives T^-1 [rhs | border columns] of the block-tridiagonal part T
// (blocks padded to 16x16; bs <= 16, 1 + nb <= 16).  algo 0 = block elimination serial in the node index (one
// wavefront per system), 1 = parallel cyclic reduction (one wavefront per node and level).  Events bracket the solve
// proper (after packing).  *singular != 0 when a pivot vanished (no pivoting inside blocks).
int blocktri_run(long batch, int n, int bs, int nb, const double *ddiag, const double *dlower, const double *dupper,
                 const double *dborder, const double *drhs, double *ws, double *dY, int algo, int *singular,
                 hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, char *err, size_t errlen);

// doubles per node of the workspace image: blocks L, D, U, R, Dinv, each [4][64] in the MFMA accumulator layout
// (lane l, register q: row (l>>4) + 4q, column l&15)
size_t blocktri_node_doubles();

// PCR on blocks already assembled in that image (buffer a; b = scratch of the same size); asynchronous on `stream`.
int blocktri_pcr_assembled(long batch, int n, int bs, int nb, double *a, double *b, double *dY, int *dflag, hipStream_t stream,
                           char *err, size_t errlen);

}  // namespace ascent
