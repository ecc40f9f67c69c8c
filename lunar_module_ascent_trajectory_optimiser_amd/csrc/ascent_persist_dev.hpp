// Device-side pieces shared by the persistent kernels (ascent_persist.hip: backward Euler / trapezoid / v1, p_solve; ascent_hs.hip:
// Hermite-Simpson, h_solve): the workspace layout of an NLP, the per-NLP scalar record, 16-lane reductions and broadcasts, what a node
// evaluation loads, the partial sums of the merit function and the KKT error.  Included inside each translation unit's anonymous
// namespace (everything here has internal linkage).
#pragma once
#include <hip/hip_runtime.h>

#include "ascent.h"
#include "ascent_device.hpp"
#include "ascent_tile.hpp"

namespace {
using namespace ascent;

constexpr int NPW = 4, CH = 16;                 // NLPs per wavefront, nodes per chunk
constexpr int O_Z = 0, O_U = 7, O_L = 8, O_ZB = 15;
constexpr int O_LU = 21, O_PP = 22, O_PN = 23, O_ZP = 24, O_ZN = 25;      // move penalty only: lambda_u, p, n, z_p, z_n
// Rows of an NLP's node arrays: two iterate buffers, the step, the feedback gains of the factorisation.  MP = 1: with the l1 move
// penalty (ascent_opts.move_penalty, the reference's angledoubledot.DCOST, Launch_Optimiser.py:99) the control is the eighth
// state of a stage and an iterate carries five more rows per node.
template <int MP>
struct Lay {
  static constexpr int NS = 7 + MP, NIT = 21 + 5 * MP;
  static constexpr int R_IT = 0, R_ST = 2 * NIT, R_KA = 3 * NIT, R_K0 = R_KA + NS, NROWS = R_K0 + 3;
  // LDS stage rows (one chunk): blocks of the factorisation; the forward / adjoint phases reuse the area
  // (MP: the mass row of hT F is a constant of the NLP and the barrier gradients are folded into rz, so that the stage stays within
  //  56 rows -- with 12 output rows and the small arrays 40 672 bytes per wavefront: four wavefronts per CU, as without the penalty)
  static constexpr int S_G = 0, S_E = 8, S_H = 12, S_F = 22, S_C = S_F + 7 - MP, S_RZ = S_C + NS, S_GT = S_RZ + NS, S_SC = S_GT + NS;
  static constexpr int S_ROWS = MP ? 56 : 55;        // (MP: S_SC + 3 = 55 in the factor phase; the forward phase's 7 lanes x 8 rows)
  static constexpr int OUT_ROWS = 11 + MP;
  static_assert(S_SC + (MP ? 3 : 5) <= S_ROWS, "stage rows");
};
constexpr int R_IT = 0;
constexpr int LDW = 65;                         // row stride in doubles: odd, so that the 16 rows a sweep step gathers hit 16 banks
enum {
  X_STATE, X_ITERS, X_STATUS, X_CUR, X_FIRST, X_LS, X_MU, X_NUP, X_DW, X_DWL, X_ALPHA, X_ADU, X_PHI0, X_DM, X_C1, X_SL,
  X_RTH, X_DTH, X_DNU3, X_SIG1, X_SIG2, X_RS1, X_RS2, X_CG1, X_CG2,
  X_ITB,                     // iterations spent on the coarser grids of the nested iteration
  X_PROBE, X_PDW,            // parity probe: one round at the caller's iterate, mu and delta_w, then stop (1: Newton step; 2: the node rows of the factor phase)
  X_TEVAL,                   // the trial point of the next round has been evaluated already (by the adjoint phase)
  X_P,                       // 10 reduced partials of that trial point: rd cinf pmin pmax l1 zsum rth c1 sl mv
  X_PEND = X_P + 9,
  X_MV,                      // move penalty: sum of the slack pairs of the iterate (its part of the objective, without the weight)
  X_S,                       // 10 scalars of the iterate
  X_D = X_S + 10,            // 10 step scalars
  NSCAL = X_D + 10
};
constexpr int NSCAL_PAD = 64;
static_assert(NSCAL <= NSCAL_PAD, "scalar record");
enum { ST_TRIAL = 0, ST_FACTOR = 1, ST_FACTORED = 2, ST_DONE = 3 };

struct PGeo {
  int K, Kp, nch, form, mp, term, wide;      // wide: one NLP per wavefront, 64-node chunks; term: ascent_opts.terminal 2 (burnout anywhere on the ellipse) or 0
  __host__ __device__ int nit() const { return mp ? Lay<1>::NIT : Lay<0>::NIT; }
  __host__ __device__ int r_st() const { return 2 * nit(); }
  __host__ __device__ int nrows() const { return mp ? Lay<1>::NROWS : Lay<0>::NROWS; }
  // (the scalar record padded to 64 doubles: with Kp a multiple of 16 every NLP then starts on a 128-byte line, and so does every row of it --
  //  a chunk row of an NLP is ONE line instead of straddling two)
  __host__ __device__ size_t nlp_doubles() const { return (size_t)nrows() * Kp + NSCAL_PAD; }
};

template <int SRC>
ASC_DEV double bcast16(double v) {
  const long x = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(long, v), 0x150 + SRC, 0xf, 0xf, false);
  return __builtin_bit_cast(double, x);
}
template <int N>
struct OneHot {
  double m[N];
  ASC_DEV explicit OneHot(int role) {
    ASC_UNROLL
    for (int i = 0; i < N; i++) m[i] = (role == i || (i == 0 && role >= N)) ? 1.0 : 0.0;
  }
  ASC_DEV double pick(const double *v) const {
    double r = m[0] * v[0];
    ASC_UNROLL
    for (int i = 1; i < N; i++) r += m[i] * v[i];
    return r;
  }
};
ASC_DEV void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// reductions over the 16 lanes of an NLP (xor strides stay inside the row of 16)
ASC_DEV double gsum16(double v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); return v; }
ASC_DEV double gmax16(double v) { v = fmax(v, __shfl_xor(v, 1)); v = fmax(v, __shfl_xor(v, 2)); v = fmax(v, __shfl_xor(v, 4)); return fmax(v, __shfl_xor(v, 8)); }
ASC_DEV double gmin16(double v) { v = fmin(v, __shfl_xor(v, 1)); v = fmin(v, __shfl_xor(v, 2)); v = fmin(v, __shfl_xor(v, 4)); return fmin(v, __shfl_xor(v, 8)); }
// ... or over the whole wavefront (WIDE: one NLP per wavefront)
template <int WIDE> ASC_DEV double gsumW(double v) { v = gsum16(v); if constexpr (WIDE) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); } return v; }
template <int WIDE> ASC_DEV double gmaxW(double v) { v = gmax16(v); if constexpr (WIDE) { v = fmax(v, __shfl_xor(v, 16)); v = fmax(v, __shfl_xor(v, 32)); } return v; }
template <int WIDE> ASC_DEV double gminW(double v) { v = gmin16(v); if constexpr (WIDE) { v = fmin(v, __shfl_xor(v, 16)); v = fmin(v, __shfl_xor(v, 32)); } return v; }

template <typename PT>      // (a generic or an LDS pointer)
ASC_DEV Scal lds_scal(PT sc, int r0) {
  Scal s;
  s.th = sc[r0 + S_TH]; s.zlt = sc[r0 + S_ZLT]; s.zut = sc[r0 + S_ZUT]; s.s1 = sc[r0 + S_S1]; s.s2 = sc[r0 + S_S2];
  s.zs1 = sc[r0 + S_ZS1]; s.zs2 = sc[r0 + S_ZS2]; s.nu3 = sc[r0 + S_NU3]; s.nu1 = sc[r0 + S_NU1]; s.nu2 = sc[r0 + S_NU2];
  return s;
}
ASC_DEV void put_scal(double *sc, int r0, const Scal &s) {
  sc[r0 + S_TH] = s.th; sc[r0 + S_ZLT] = s.zlt; sc[r0 + S_ZUT] = s.zut; sc[r0 + S_S1] = s.s1; sc[r0 + S_S2] = s.s2;
  sc[r0 + S_ZS1] = s.zs1; sc[r0 + S_ZS2] = s.zs2; sc[r0 + S_NU3] = s.nu3; sc[r0 + S_NU1] = s.nu1; sc[r0 + S_NU2] = s.nu2;
}
ASC_DEV Scal trial_scal(const Der &d, const Scal &s, const Scal &ds, double alpha, double adu, double mu, bool first) {
  Scal t = s;
  if (first) return t;
  t.th += alpha * ds.th; t.s1 += alpha * ds.s1; t.s2 += alpha * ds.s2;
  t.nu3 += alpha * ds.nu3; t.nu1 += alpha * ds.nu1; t.nu2 += alpha * ds.nu2;
  t.zlt = clipz(s.zlt + adu * ds.zlt, t.th - d.tlb, mu);
  t.zut = clipz(s.zut + adu * ds.zut, d.tub - t.th, mu);
  t.zs1 = clipz(s.zs1 + adu * ds.zs1, t.s1, mu);
  t.zs2 = clipz(s.zs2 + adu * ds.zs2, t.s2, mu);
  return t;
}

struct NodeIn {      // what a node evaluation reads: node k of the iterate, the state of node k-1, the multipliers of node k+1
  double z[7], zp[7], l[7], ln[7], zb[6], u;
  double up, lu, lun, pp, pn, zpp, zpn;      // move penalty only: u_{k-1}, lambda_u of nodes k and k+1, the slack pair and its multipliers
};
// (uinit: the control "before node 0" of the movement equations -- the MV's initial value: 0, or -1 where the v1 formulation's
//  angle starts at 0; loads of a STEP pass 0)
template <int MP = 0>
ASC_DEV void load_node(const double *it, int Kp, int K, int k, NodeIn &n, double uinit = 0.0) {
  ASC_UNROLL
  for (int i = 0; i < 7; i++) {
    n.z[i] = it[(O_Z + i) * Kp + k];
    n.l[i] = it[(O_L + i) * Kp + k];
    n.zp[i] = k > 0 ? it[(O_Z + i) * Kp + k - 1] : 0.0;
    n.ln[i] = k + 1 < K ? it[(O_L + i) * Kp + k + 1] : 0.0;
  }
  n.u = it[O_U * Kp + k];
  ASC_UNROLL
  for (int b = 0; b < 6; b++) n.zb[b] = it[(O_ZB + b) * Kp + k];
  if constexpr (MP) {
    n.up = k > 0 ? it[O_U * Kp + k - 1] : uinit;
    n.lu = it[O_LU * Kp + k];
    n.lun = k + 1 < K ? it[O_LU * Kp + k + 1] : 0.0;
    n.pp = it[O_PP * Kp + k]; n.pn = it[O_PN * Kp + k]; n.zpp = it[O_ZP * Kp + k]; n.zpn = it[O_ZN * Kp + k];
  }
}

// Partial sums of the merit function and the KKT error over the nodes a lane evaluates
struct Part {
  double rd, cinf, pmin, pmax, l1, zsum, rth, c1, sl, mv;
  ASC_DEV void clear() { rd = 0.0; cinf = 0.0; pmin = 1e300; pmax = -1e300; l1 = 0.0; zsum = 0.0; rth = 0.0; c1 = 0.0; sl = 0.0; mv = 0.0; }
  template <int MP = 0>
  ASC_DEV void reduce16() {
    rd = gmax16(rd); cinf = gmax16(cinf); pmin = gmin16(pmin); pmax = gmax16(pmax);
    l1 = gsum16(l1); zsum = gsum16(zsum); rth = gsum16(rth); c1 = gsum16(c1); sl = gsum16(sl);
    if constexpr (MP) mv = gsum16(mv);
  }
  template <int MP, int WIDE>
  ASC_DEV void reduceW() {
    rd = gmaxW<WIDE>(rd); cinf = gmaxW<WIDE>(cinf); pmin = gminW<WIDE>(pmin); pmax = gmaxW<WIDE>(pmax);
    l1 = gsumW<WIDE>(l1); zsum = gsumW<WIDE>(zsum); rth = gsumW<WIDE>(rth); c1 = gsumW<WIDE>(c1); sl = gsumW<WIDE>(sl);
    if constexpr (MP) mv = gsumW<WIDE>(mv);
  }
};
struct TrialCtx {       // what the trial point of an NLP needs besides the node data
  double alpha, adu, mlo, mhi, dt, be, hT, dcw;
  bool first;
  Scal stt;
};

#ifdef PERSIST_PROFILE      // diagnostic build (scripts/persist_profile.py): shader cycles per phase, wavefront 0
#define PROF_DECL long long prof_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long pt_ = clock64();
#define PROF(i_) do { const long long t1_ = clock64(); prof_[i_] += t1_ - pt_; pt_ = t1_; } while (0)
#define PROF_END do { if (blockIdx.x == 0 && threadIdx.x == 0) printf("[persist profile] cycles: A %lld | B eval %lld serial %lld flush %lld | F eval %lld serial %lld post %lld | Adj eval %lld serial %lld | rest %lld\n", prof_[0], prof_[1], prof_[2], prof_[3], prof_[4], prof_[5], prof_[6], prof_[7], prof_[8], prof_[9]); } while (0)
#else
#define PROF_DECL
#define PROF(i_) do { } while (0)
#define PROF_END do { } while (0)
#endif

}  // namespace
