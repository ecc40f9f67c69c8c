// Generic bordered block-tridiagonal solver on 16x16 blocks (gfx950): the parity surface ascent_kkt_solve of
// include/ascent.h, and the measured answer to "parallel cyclic reduction over the collocation nodes".
//
// One Newton system of the ascent NLP (/root/reference/Launch_Optimiser.py:177, the solve IPOPT/MUMPS does) ordered by
// collocation node is block tridiagonal with 15x15 blocks (7 states, the control, 7 defect multipliers of a node) plus
// border columns (the free final time, LO:39-40/114-123, and the multiplier of the terminal r.v = 0, LO:173).  The
// product solves it by a Riccati recursion in the 7x7 value function (ascent_pipeline.hip, ascent_dense.hip), which is
// serial in the node index.  This file solves the SAME kind of system generically, in two ways:
//   algo 0  block elimination, serial in the node index: one wavefront per system;
//   algo 1  parallel cyclic reduction (PCR): one wavefront per (system, node), log2(n) levels, every level eliminates the
//           couplings to the nodes +-stride away on all nodes at once.
// A 15x15 block padded to 16x16 is exactly one v_mfma_f64_16x16x4_f64 tile: a block product is four MFMA
// instructions.  A block lives in the MFMA accumulator layout for its whole life -- lane l, register q holds element
// (row (l>>4) + 4q, column l&15) -- because that layout is also the B operand of the next MFMA (register kb IS k-block
// kb) and, as the A operand, multiplies by the TRANSPOSE (cdna_hip_programming.md, "an accumulator tile as the next
// MFMA's operand"): X'Y costs no data movement at all, XY one 2-KB transpose of X through LDS.
// FP64 MFMA has the same peak rate as FP64 FMA on MI355X; what it buys here is 16x fewer instructions on a
// latency-bound critical path (one wavefront per block), not throughput.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ascent.h"
#include "ascent_blocktri.hpp"

namespace {

constexpr int WAVE = 64, BS = 16, GRID_D = BS * BS;      // doubles per block
typedef double d4 __attribute__((ext_vector_type(4)));
#define BT_DEV __device__ __forceinline__

struct Blk { d4 v; };     // one 16x16 block, accumulator layout

BT_DEV int brow(int q) { return (threadIdx.x >> 4) + 4 * q; }
BT_DEV int bcol() { return threadIdx.x & 15; }
BT_DEV void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// workspace image of a block: [q][lane] so that every access is one contiguous 512-byte row
BT_DEV Blk bload(const double *p) {
  Blk b;
  _Pragma("unroll") for (int q = 0; q < 4; q++) b.v[q] = p[q * WAVE + threadIdx.x];
  return b;
}
BT_DEV void bstore(double *p, const Blk &b) {
  _Pragma("unroll") for (int q = 0; q < 4; q++) p[q * WAVE + threadIdx.x] = b.v[q];
}
BT_DEV Blk bzero() { Blk b; b.v = d4{0.0, 0.0, 0.0, 0.0}; return b; }
// C = X'Y (+ C0): four MFMAs, no data movement
BT_DEV Blk mmT(const Blk &X, const Blk &Y, const Blk &C0) {
  d4 acc = C0.v;
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X.v[0], Y.v[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X.v[1], Y.v[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X.v[2], Y.v[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X.v[3], Y.v[3], acc, 0, 0, 0);
  Blk c; c.v = acc;
  return c;
}
BT_DEV Blk btr(double *lds, const Blk &X) {      // transpose through LDS
  _Pragma("unroll") for (int q = 0; q < 4; q++) lds[bcol() * BS + brow(q)] = X.v[q];
  wsync();
  Blk t;
  _Pragma("unroll") for (int q = 0; q < 4; q++) t.v[q] = lds[brow(q) * BS + bcol()];
  wsync();
  return t;
}
BT_DEV Blk bneg(const Blk &X) { Blk r; r.v = -X.v; return r; }
// Gauss-Jordan inverse in LDS (no pivoting: the caller's blocks must have non-vanishing leading pivots -- block
// diagonally dominant test systems; KKT blocks ordered (z, u, lambda) with a positive definite (z,u) part).  Only a vanishing
// pivot is reported: a threshold relative to the block's largest entry was tried (round 3, 1e-13) and rejects healthy
// interior-point blocks, whose entries span the barrier's sigma = z/s ~ 1e14 down to pivots of 1e-4; the guard against a
// wrong step is the caller's curvature test along the step (ascent.h: a weaker guarantee than the exact inertia of the Riccati forms).
BT_DEV Blk binv(double *lm, double *lv, const Blk &X, int &bad) {
  double m[4], v[4];
  _Pragma("unroll") for (int q = 0; q < 4; q++) { m[q] = X.v[q]; v[q] = brow(q) == bcol() ? 1.0 : 0.0; }
  for (int k = 0; k < BS; k++) {
    _Pragma("unroll") for (int q = 0; q < 4; q++) { lm[brow(q) * BS + bcol()] = m[q]; lv[brow(q) * BS + bcol()] = v[q]; }
    wsync();
    const double piv = lm[k * BS + k];
    if (!(fabs(piv) > 1e-300)) bad = 1;
    const double ip = 1.0 / piv, mkj = lm[k * BS + bcol()], vkj = lv[k * BS + bcol()];
    _Pragma("unroll") for (int q = 0; q < 4; q++) {
      const int i = brow(q);
      const double mik = lm[i * BS + k];
      if (i == k) { m[q] = mkj * ip; v[q] = vkj * ip; }
      else { const double f = mik * ip; m[q] = fma(-f, mkj, m[q]); v[q] = fma(-f, vkj, v[q]); }
    }
    wsync();
  }
  Blk r;
  _Pragma("unroll") for (int q = 0; q < 4; q++) r.v[q] = v[q];
  return r;
}

// ---- workspace: per node the blocks L (coupling to node i-stride), D, U (to node i+stride), R (right-hand sides as
// columns), Dinv; two buffers for PCR ------------------------------------------------------------------------------------
constexpr int B_L = 0, B_D = 1, B_U = 2, B_R = 3, B_I = 4, NB_ = 5;
__host__ __device__ inline size_t node_doubles() { return (size_t)NB_ * GRID_D; }

// pack the caller's arrays ([batch][n][bs][bs], row-major; border [batch][n][bs][nb]; rhs [batch][n*bs + nb]) into padded
// blocks: identity on the padded diagonal; R = [rhs | border columns]
__global__ __launch_bounds__(WAVE) void bt_pack(long batch, int n, int bs, int nb, const double *diag, const double *lower,
                                                const double *upper, const double *border, const double *rhs, double *ws) {
  const long sys = blockIdx.y;
  const int i = blockIdx.x;
  double *w = ws + ((size_t)sys * n + i) * node_doubles();
  const size_t blk = ((size_t)sys * n + i) * bs * bs;
  Blk L = bzero(), D = bzero(), U = bzero(), R = bzero();
  _Pragma("unroll") for (int q = 0; q < 4; q++) {
    const int r = brow(q), c = bcol();
    if (r < bs && c < bs) {
      D.v[q] = diag[blk + r * bs + c];
      if (i > 0) L.v[q] = lower[blk + r * bs + c];
      if (i < n - 1) U.v[q] = upper[blk + r * bs + c];
    } else if (r == c) {
      D.v[q] = 1.0;
    }
    if (r < bs) {
      if (c == 0) R.v[q] = rhs[(size_t)sys * ((size_t)n * bs + nb) + (size_t)i * bs + r];
      else if (c - 1 < nb) R.v[q] = border[(((size_t)sys * n + i) * bs + r) * nb + (c - 1)];
    }
  }
  bstore(w + B_L * GRID_D, L); bstore(w + B_D * GRID_D, D); bstore(w + B_U * GRID_D, U); bstore(w + B_R * GRID_D, R);
}

__global__ __launch_bounds__(WAVE) void bt_invert(int n, double *ws, int *flag) {
  __shared__ double lm[GRID_D], lv[GRID_D];
  double *w = ws + ((size_t)blockIdx.y * n + blockIdx.x) * node_doubles();
  int bad = 0;
  const Blk Di = binv(lm, lv, bload(w + B_D * GRID_D), bad);
  bstore(w + B_I * GRID_D, Di);
  if (bad && threadIdx.x == 0) atomicOr(flag + blockIdx.y, 1);
}

// one PCR level: node i eliminates its couplings to i-s and i+s
//   a = -L_i Dinv_{i-s}      c = -U_i Dinv_{i+s}
//   D' = D + a U_{i-s} + c L_{i+s}     R' = R + a R_{i-s} + c R_{i+s}     L' = a L_{i-s}     U' = c U_{i+s}
// and inverts the new diagonal block for the next level
__global__ __launch_bounds__(WAVE) void bt_pcr_level(int n, int s, const double *src, double *dst, int *flag) {
  __shared__ double lm[GRID_D], lv[GRID_D];
  const long sys = blockIdx.y;
  const int i = blockIdx.x;
  const double *w = src + ((size_t)sys * n + i) * node_doubles();
  double *o = dst + ((size_t)sys * n + i) * node_doubles();
  Blk D = bload(w + B_D * GRID_D), R = bload(w + B_R * GRID_D), Ln = bzero(), Un = bzero();
  if (i - s >= 0) {
    const double *m = src + ((size_t)sys * n + (i - s)) * node_doubles();
    const Blk Lt = btr(lm, bload(w + B_L * GRID_D));
    const Blk a = bneg(mmT(Lt, bload(m + B_I * GRID_D), bzero()));          // -L Dinv_m
    const Blk at = btr(lm, a);
    D = mmT(at, bload(m + B_U * GRID_D), D);
    R = mmT(at, bload(m + B_R * GRID_D), R);
    Ln = mmT(at, bload(m + B_L * GRID_D), bzero());
  }
  if (i + s < n) {
    const double *pn = src + ((size_t)sys * n + (i + s)) * node_doubles();
    const Blk Ut = btr(lm, bload(w + B_U * GRID_D));
    const Blk c = bneg(mmT(Ut, bload(pn + B_I * GRID_D), bzero()));
    const Blk ct = btr(lm, c);
    D = mmT(ct, bload(pn + B_L * GRID_D), D);
    R = mmT(ct, bload(pn + B_R * GRID_D), R);
    Un = mmT(ct, bload(pn + B_U * GRID_D), bzero());
  }
  int bad = 0;
  const Blk Di = binv(lm, lv, D, bad);
  bstore(o + B_L * GRID_D, Ln); bstore(o + B_D * GRID_D, D); bstore(o + B_U * GRID_D, Un); bstore(o + B_R * GRID_D, R);
  bstore(o + B_I * GRID_D, Di);
  if (bad && threadIdx.x == 0) atomicOr(flag + sys, 1);
}

// X_i = Dinv_i R_i, written in the caller's layout: Y[batch][n][bs][1+nb] (column 0: T^-1 rhs, the rest: T^-1 border)
__global__ __launch_bounds__(WAVE) void bt_finish(int n, int bs, int nb, const double *ws, double *Y) {
  __shared__ double lm[GRID_D];
  const long sys = blockIdx.y;
  const int i = blockIdx.x;
  const double *w = ws + ((size_t)sys * n + i) * node_doubles();
  const Blk X = mmT(btr(lm, bload(w + B_I * GRID_D)), bload(w + B_R * GRID_D), bzero());
  _Pragma("unroll") for (int q = 0; q < 4; q++) {
    const int r = brow(q), c = bcol();
    if (r < bs && c < 1 + nb) Y[(((size_t)sys * n + i) * bs + r) * (1 + nb) + c] = X.v[q];
  }
}

// algo 0: block elimination, one wavefront per system
//   forward:  W = L_i Dinv'_{i-1};  D'_i = D_i - W U_{i-1};  R'_i = R_i - W R'_{i-1};  Dinv'_i
//   backward: X_{n-1} = Dinv' R';   X_i = Dinv'_i (R'_i - U_i X_{i+1})
__global__ __launch_bounds__(WAVE) void bt_thomas(int n, int bs, int nb, double *ws, double *Y, int *flag) {
  __shared__ double lm[GRID_D], lv[GRID_D];
  const long sys = blockIdx.x;
  double *base = ws + (size_t)sys * n * node_doubles();
  int bad = 0;
  Blk Dip = bzero(), Rp = bzero(), Up = bzero();
  for (int i = 0; i < n; i++) {
    double *w = base + (size_t)i * node_doubles();
    Blk D = bload(w + B_D * GRID_D), R = bload(w + B_R * GRID_D);
    if (i > 0) {
      const Blk Lt = btr(lm, bload(w + B_L * GRID_D));
      const Blk Wn = bneg(mmT(Lt, Dip, bzero()));            // -L Dinv'
      const Blk Wt = btr(lm, Wn);
      D = mmT(Wt, Up, D);
      R = mmT(Wt, Rp, R);
    }
    Dip = binv(lm, lv, D, bad);
    Rp = R;
    Up = bload(w + B_U * GRID_D);
    bstore(w + B_I * GRID_D, Dip);
    bstore(w + B_R * GRID_D, R);
  }
  Blk Xn = bzero();
  for (int i = n - 1; i >= 0; i--) {
    const double *w = base + (size_t)i * node_doubles();
    Blk R = bload(w + B_R * GRID_D);
    if (i < n - 1) R = mmT(bneg(btr(lm, bload(w + B_U * GRID_D))), Xn, R);      // R - U X_{i+1}
    Xn = mmT(btr(lm, bload(w + B_I * GRID_D)), R, bzero());
    _Pragma("unroll") for (int q = 0; q < 4; q++) {
      const int r = brow(q), c = bcol();
      if (r < bs && c < 1 + nb) Y[(((size_t)sys * n + i) * bs + r) * (1 + nb) + c] = Xn.v[q];
    }
  }
  if (bad && threadIdx.x == 0) atomicOr(flag + sys, 1);
}

__global__ void bt_any(const int *flags, long batch, int *out) {
  int any = 0;
  for (long q = threadIdx.x; q < batch; q += blockDim.x) any |= flags[q];
  if (any) atomicOr(out, 1);
}

}  // namespace

namespace ascent {

size_t blocktri_ws_bytes(int n, long batch, int algo) {
  return (size_t)batch * n * node_doubles() * sizeof(double) * (algo == 1 ? 2 : 1) + ((size_t)batch + 16) * sizeof(int);
}
size_t blocktri_node_doubles() { return node_doubles(); }

#define BCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { snprintf(err, errlen, "%s: %s", #call, hipGetErrorString(e_)); return ASCENT_E_HIP; } } while (0)

// device pointers; Y [batch][n][bs][1+nb] receives T^-1 [rhs | border]; *singular is set when a pivot vanished
int blocktri_run(long batch, int n, int bs, int nb, const double *ddiag, const double *dlower, const double *dupper,
                 const double *dborder, const double *drhs, double *ws, double *dY, int algo, int *singular,
                 hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, char *err, size_t errlen) {
  const size_t per = (size_t)batch * n * node_doubles();
  int *flag = (int *)((char *)ws + per * sizeof(double) * (algo == 1 ? 2 : 1));     // [batch] per-system flags + 1 summary
  BCHK(hipMemsetAsync(flag, 0, ((size_t)batch + 1) * sizeof(int), stream));
  const dim3 g((unsigned)n, (unsigned)batch);
  hipLaunchKernelGGL(bt_pack, g, dim3(WAVE), 0, stream, batch, n, bs, nb, ddiag, dlower, dupper, dborder, drhs, ws);
  BCHK(hipGetLastError());
  if (ev0) BCHK(hipEventRecord(ev0, stream));
  if (algo == 0) {
    hipLaunchKernelGGL(bt_thomas, dim3((unsigned)batch), dim3(WAVE), 0, stream, n, bs, nb, ws, dY, flag);
  } else {
    double *a = ws, *b = ws + per;
    hipLaunchKernelGGL(bt_invert, g, dim3(WAVE), 0, stream, n, a, flag);
    for (int s = 1; s < n; s *= 2) {
      hipLaunchKernelGGL(bt_pcr_level, g, dim3(WAVE), 0, stream, n, s, (const double *)a, b, flag);
      double *t = a; a = b; b = t;
    }
    hipLaunchKernelGGL(bt_finish, g, dim3(WAVE), 0, stream, n, bs, nb, (const double *)a, dY);
  }
  BCHK(hipGetLastError());
  if (ev1) BCHK(hipEventRecord(ev1, stream));
  hipLaunchKernelGGL(bt_any, dim3(1), dim3(256), 0, stream, (const int *)flag, batch, flag + batch);
  BCHK(hipMemcpyAsync(singular, flag + batch, sizeof(int), hipMemcpyDeviceToHost, stream));
  BCHK(hipStreamSynchronize(stream));
  return ASCENT_OK;
}

// PCR on blocks the caller has already assembled in the workspace image (node_doubles() per node: L, D, U, R, Dinv as
// [4][64] accumulator-layout images; `a` = assembled buffer, `b` = scratch of the same size).  Asynchronous.
// dY [batch][n][bs][1+nb]; dflag [batch]: nonzero where a pivot vanished.
int blocktri_pcr_assembled(long batch, int n, int bs, int nb, double *a, double *b, double *dY, int *dflag, hipStream_t stream,
                           char *err, size_t errlen) {
  const dim3 g((unsigned)n, (unsigned)batch);
  hipLaunchKernelGGL(bt_invert, g, dim3(WAVE), 0, stream, n, a, dflag);
  for (int s = 1; s < n; s *= 2) {
    hipLaunchKernelGGL(bt_pcr_level, g, dim3(WAVE), 0, stream, n, s, (const double *)a, b, dflag);
    double *t = a; a = b; b = t;
  }
  hipLaunchKernelGGL(bt_finish, g, dim3(WAVE), 0, stream, n, bs, nb, (const double *)a, dY);
  BCHK(hipGetLastError());
  return ASCENT_OK;
}

}  // namespace ascent
