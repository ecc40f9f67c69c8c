// Shared device-side plumbing of the solver kernels: the wave-tile view of HBM, row load/store
// helpers, the double-buffered sweep skeletons and small numeric helpers.
#pragma once
#include "ascent_device.hpp"

namespace ascent {

constexpr int WAVE = 64;
typedef __attribute__((address_space(1))) double gdbl;   // HBM (global address space) double


// A pointer / integer that is the same in all 64 lanes, moved to scalar registers so that the
// compiler addresses HBM as  scalar base + lane  (global_load ... s[base:base+1]).
ASC_DEV gdbl *uniform(gdbl *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (gdbl *)(((unsigned long long)hi << 32) | lo);
}
ASC_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int STAGE_ROWS>
struct TileT {  // uniform tile base + this lane; step records of STAGE_ROWS rows
  gdbl *base;
  unsigned lane;
  ASC_DEV explicit TileT(gdbl *tile) : base(uniform(tile)), lane(threadIdx.x) {}
  ASC_DEV TileT(gdbl *tile, unsigned lane_) : base(uniform(tile)), lane(lane_) {}
  ASC_DEV gdbl *st(int k) const { return base + (size_t)k * (STAGE_ROWS * WAVE); }
};
#define ROW(p, r) (p)[(r) * WAVE + t_.lane]
#define ASC_PASS __device__ __noinline__   // one register allocation per pass

template <int N, class TileX>
ASC_DEV void ldn(const TileX &t_, const gdbl *p, int r0, double *v) {
  ASC_UNROLL
  for (int i = 0; i < N; i++) v[i] = ROW(p, r0 + i);
}
template <int N, class TileX>
ASC_DEV void stn(const TileX &t_, gdbl *p, int r0, const double *v) {
  ASC_UNROLL
  for (int i = 0; i < N; i++) ROW(p, r0 + i) = v[i];
}
template <int N>
ASC_DEV void cpy(double *dst, const double *src) {
  ASC_UNROLL
  for (int i = 0; i < N; i++) dst[i] = src[i];
}

// scalars of the iterate / step kept in registers
struct Scal {
  double th, zlt, zut, s1, s2, zs1, zs2, nu3, nu1, nu2;
};

struct ErrParts {  // E(mu) = max(rd/sd, cinf, comp(mu)/sd), comp(mu) from the extreme complementarity products
  double rd, cinf, pmin, pmax, sd;
  ASC_DEV double err(double mu) const {
    const double comp = fmax(fabs(pmax - mu), fabs(pmin - mu));
    return fmax(fmax(rd / sd, cinf), comp / sd);
  }
};

ASC_DEV double clipz(double zv, double dist, double mu) {
  return fmin(fmax(zv, mu / (1e10 * dist)), 1e10 * mu / dist);
}

// ---------------------------------------------------------------------------------------------
// Loop skeleton shared by all passes: the step records are double-buffered in registers (A, B);
// while step k is being computed the loads of the next step are already in flight.  Unrolled by two
// so that the buffers swap roles without register copies.
// ---------------------------------------------------------------------------------------------
#define ASC_SWEEP_BACKWARD(IN, LOAD, BODY)                 \
  {                                                        \
    IN bufA, bufB;                                         \
    LOAD(K - 1, bufA);                                     \
    int k = K - 1;                                         \
    for (; k >= 1; k -= 2) {                               \
      LOAD(k - 1, bufB);                                   \
      BODY(bufA, k);                                       \
      if (k >= 2) LOAD(k - 2, bufA);                       \
      BODY(bufB, k - 1);                                   \
    }                                                      \
    if (k == 0) BODY(bufA, 0);                             \
  }
#define ASC_SWEEP_FORWARD(IN, LOAD, BODY)                  \
  {                                                        \
    IN bufA, bufB;                                         \
    LOAD(0, bufA);                                         \
    int k = 0;                                             \
    for (; k + 1 < K; k += 2) {                            \
      LOAD(k + 1, bufB);                                   \
      BODY(bufA, k);                                       \
      if (k + 2 < K) LOAD(k + 2, bufA);                    \
      BODY(bufB, k + 1);                                   \
    }                                                      \
    if (k == K - 1) BODY(bufA, k);                         \
  }

// Same skeletons with three steps in flight (four register buffers) for passes whose per-step work is
// too short to cover one HBM round trip.
#define ASC_SWEEP_BACKWARD4(IN, LOAD, BODY)                \
  {                                                        \
    IN b0_, b1_, b2_, b3_;                                 \
    int k = K - 1;                                         \
    LOAD(k, b0_);                                          \
    if (k >= 1) LOAD(k - 1, b1_);                          \
    if (k >= 2) LOAD(k - 2, b2_);                          \
    for (; k >= 3; k -= 4) {                               \
      LOAD(k - 3, b3_);                                    \
      BODY(b0_, k);                                        \
      if (k >= 4) LOAD(k - 4, b0_);                        \
      BODY(b1_, k - 1);                                    \
      if (k >= 5) LOAD(k - 5, b1_);                        \
      BODY(b2_, k - 2);                                    \
      if (k >= 6) LOAD(k - 6, b2_);                        \
      BODY(b3_, k - 3);                                    \
    }                                                      \
    if (k >= 0) BODY(b0_, k);                              \
    if (k >= 1) BODY(b1_, k - 1);                          \
    if (k >= 2) BODY(b2_, k - 2);                          \
  }
#define ASC_SWEEP_FORWARD4(IN, LOAD, BODY)                 \
  {                                                        \
    IN b0_, b1_, b2_, b3_;                                 \
    int k = 0;                                             \
    LOAD(0, b0_);                                          \
    if (1 < K) LOAD(1, b1_);                               \
    if (2 < K) LOAD(2, b2_);                               \
    for (; k + 3 < K; k += 4) {                            \
      LOAD(k + 3, b3_);                                    \
      BODY(b0_, k);                                        \
      if (k + 4 < K) LOAD(k + 4, b0_);                     \
      BODY(b1_, k + 1);                                    \
      if (k + 5 < K) LOAD(k + 5, b1_);                     \
      BODY(b2_, k + 2);                                    \
      if (k + 6 < K) LOAD(k + 6, b2_);                     \
      BODY(b3_, k + 3);                                    \
    }                                                      \
    if (k < K) BODY(b0_, k);                               \
    if (k + 1 < K) BODY(b1_, k + 1);                       \
    if (k + 2 < K) BODY(b2_, k + 2);                       \
  }

// The four-buffer skeletons with unconditional prefetches (the step index is clamped at the end of the sweep
// instead of guarding the load): a load under a branch makes the compiler's s_waitcnt placement assume the
// shorter queue and wait for the newest loads.  Used by the 16-lanes-per-NLP sweeps, whose steps are short; the
// one-lane-per-NLP passes measured slower with it (more live registers across the body) and keep the guards.
#define ASC_CLAMP_LO_(k_) ((k_) > 0 ? (k_) : 0)
#define ASC_CLAMP_HI_(k_) ((k_) < K - 1 ? (k_) : K - 1)
// the fence keeps the prefetches where they are written: one full body ahead of each use
#define ASC_SCHED_FENCE_ __builtin_amdgcn_sched_barrier(0)
#define ASC_SWEEP_BACKWARD4U(IN, LOAD, BODY)             \
  {                                                      \
    IN b0_, b1_, b2_, b3_;                               \
    int k = K - 1;                                       \
    LOAD(ASC_CLAMP_LO_(k - 0), b0_); ASC_SCHED_FENCE_;   \
    LOAD(ASC_CLAMP_LO_(k - 1), b1_); ASC_SCHED_FENCE_;   \
    LOAD(ASC_CLAMP_LO_(k - 2), b2_); ASC_SCHED_FENCE_;   \
    for (; k >= 3; k -= 4) {                             \
      LOAD(ASC_CLAMP_LO_(k - 3), b3_); ASC_SCHED_FENCE_; \
      BODY(b0_, k - 0);                                  \
      LOAD(ASC_CLAMP_LO_(k - 4), b0_); ASC_SCHED_FENCE_; \
      BODY(b1_, k - 1);                                  \
      LOAD(ASC_CLAMP_LO_(k - 5), b1_); ASC_SCHED_FENCE_; \
      BODY(b2_, k - 2);                                  \
      LOAD(ASC_CLAMP_LO_(k - 6), b2_); ASC_SCHED_FENCE_; \
      BODY(b3_, k - 3);                                  \
    }                                                    \
    if (k >= 0) BODY(b0_, k - 0);                        \
    if (k >= 1) BODY(b1_, k - 1);                        \
    if (k >= 2) BODY(b2_, k - 2);                        \
  }
#define ASC_SWEEP_FORWARD4U(IN, LOAD, BODY)              \
  {                                                      \
    IN b0_, b1_, b2_, b3_;                               \
    int k = 0;                                           \
    LOAD(ASC_CLAMP_HI_(0), b0_); ASC_SCHED_FENCE_;       \
    LOAD(ASC_CLAMP_HI_(1), b1_); ASC_SCHED_FENCE_;       \
    LOAD(ASC_CLAMP_HI_(2), b2_); ASC_SCHED_FENCE_;       \
    for (; k + 3 < K; k += 4) {                          \
      LOAD(ASC_CLAMP_HI_(k + 3), b3_); ASC_SCHED_FENCE_; \
      BODY(b0_, k + 0);                                  \
      LOAD(ASC_CLAMP_HI_(k + 4), b0_); ASC_SCHED_FENCE_; \
      BODY(b1_, k + 1);                                  \
      LOAD(ASC_CLAMP_HI_(k + 5), b1_); ASC_SCHED_FENCE_; \
      BODY(b2_, k + 2);                                  \
      LOAD(ASC_CLAMP_HI_(k + 6), b2_); ASC_SCHED_FENCE_; \
      BODY(b3_, k + 3);                                  \
    }                                                    \
    if (k + 0 < K) BODY(b0_, k + 0);                     \
    if (k + 1 < K) BODY(b1_, k + 1);                     \
    if (k + 2 < K) BODY(b2_, k + 2);                     \
  }
#define ASC_SWEEP_BACKWARD8U(IN, LOAD, BODY)              \
  {                                                       \
    IN b0_, b1_, b2_, b3_, b4_, b5_, b6_, b7_;            \
    int k = K - 1;                                        \
    LOAD(ASC_CLAMP_LO_(k - 0), b0_); ASC_SCHED_FENCE_;    \
    LOAD(ASC_CLAMP_LO_(k - 1), b1_); ASC_SCHED_FENCE_;    \
    LOAD(ASC_CLAMP_LO_(k - 2), b2_); ASC_SCHED_FENCE_;    \
    LOAD(ASC_CLAMP_LO_(k - 3), b3_); ASC_SCHED_FENCE_;    \
    LOAD(ASC_CLAMP_LO_(k - 4), b4_); ASC_SCHED_FENCE_;    \
    LOAD(ASC_CLAMP_LO_(k - 5), b5_); ASC_SCHED_FENCE_;    \
    LOAD(ASC_CLAMP_LO_(k - 6), b6_); ASC_SCHED_FENCE_;    \
    for (; k >= 7; k -= 8) {                              \
      LOAD(ASC_CLAMP_LO_(k - 7), b7_); ASC_SCHED_FENCE_;  \
      BODY(b0_, k - 0);                                   \
      LOAD(ASC_CLAMP_LO_(k - 8), b0_); ASC_SCHED_FENCE_;  \
      BODY(b1_, k - 1);                                   \
      LOAD(ASC_CLAMP_LO_(k - 9), b1_); ASC_SCHED_FENCE_;  \
      BODY(b2_, k - 2);                                   \
      LOAD(ASC_CLAMP_LO_(k - 10), b2_); ASC_SCHED_FENCE_; \
      BODY(b3_, k - 3);                                   \
      LOAD(ASC_CLAMP_LO_(k - 11), b3_); ASC_SCHED_FENCE_; \
      BODY(b4_, k - 4);                                   \
      LOAD(ASC_CLAMP_LO_(k - 12), b4_); ASC_SCHED_FENCE_; \
      BODY(b5_, k - 5);                                   \
      LOAD(ASC_CLAMP_LO_(k - 13), b5_); ASC_SCHED_FENCE_; \
      BODY(b6_, k - 6);                                   \
      LOAD(ASC_CLAMP_LO_(k - 14), b6_); ASC_SCHED_FENCE_; \
      BODY(b7_, k - 7);                                   \
    }                                                     \
    if (k >= 0) BODY(b0_, k - 0);                         \
    if (k >= 1) BODY(b1_, k - 1);                         \
    if (k >= 2) BODY(b2_, k - 2);                         \
    if (k >= 3) BODY(b3_, k - 3);                         \
    if (k >= 4) BODY(b4_, k - 4);                         \
    if (k >= 5) BODY(b5_, k - 5);                         \
    if (k >= 6) BODY(b6_, k - 6);                         \
  }
#define ASC_SWEEP_FORWARD8U(IN, LOAD, BODY)               \
  {                                                       \
    IN b0_, b1_, b2_, b3_, b4_, b5_, b6_, b7_;            \
    int k = 0;                                            \
    LOAD(ASC_CLAMP_HI_(0), b0_); ASC_SCHED_FENCE_;        \
    LOAD(ASC_CLAMP_HI_(1), b1_); ASC_SCHED_FENCE_;        \
    LOAD(ASC_CLAMP_HI_(2), b2_); ASC_SCHED_FENCE_;        \
    LOAD(ASC_CLAMP_HI_(3), b3_); ASC_SCHED_FENCE_;        \
    LOAD(ASC_CLAMP_HI_(4), b4_); ASC_SCHED_FENCE_;        \
    LOAD(ASC_CLAMP_HI_(5), b5_); ASC_SCHED_FENCE_;        \
    LOAD(ASC_CLAMP_HI_(6), b6_); ASC_SCHED_FENCE_;        \
    for (; k + 7 < K; k += 8) {                           \
      LOAD(ASC_CLAMP_HI_(k + 7), b7_); ASC_SCHED_FENCE_;  \
      BODY(b0_, k + 0);                                   \
      LOAD(ASC_CLAMP_HI_(k + 8), b0_); ASC_SCHED_FENCE_;  \
      BODY(b1_, k + 1);                                   \
      LOAD(ASC_CLAMP_HI_(k + 9), b1_); ASC_SCHED_FENCE_;  \
      BODY(b2_, k + 2);                                   \
      LOAD(ASC_CLAMP_HI_(k + 10), b2_); ASC_SCHED_FENCE_; \
      BODY(b3_, k + 3);                                   \
      LOAD(ASC_CLAMP_HI_(k + 11), b3_); ASC_SCHED_FENCE_; \
      BODY(b4_, k + 4);                                   \
      LOAD(ASC_CLAMP_HI_(k + 12), b4_); ASC_SCHED_FENCE_; \
      BODY(b5_, k + 5);                                   \
      LOAD(ASC_CLAMP_HI_(k + 13), b5_); ASC_SCHED_FENCE_; \
      BODY(b6_, k + 6);                                   \
      LOAD(ASC_CLAMP_HI_(k + 14), b6_); ASC_SCHED_FENCE_; \
      BODY(b7_, k + 7);                                   \
    }                                                     \
    if (k + 0 < K) BODY(b0_, k + 0);                      \
    if (k + 1 < K) BODY(b1_, k + 1);                      \
    if (k + 2 < K) BODY(b2_, k + 2);                      \
    if (k + 3 < K) BODY(b3_, k + 3);                      \
    if (k + 4 < K) BODY(b4_, k + 4);                      \
    if (k + 5 < K) BODY(b5_, k + 5);                      \
    if (k + 6 < K) BODY(b6_, k + 6);                      \
  }

#define ASC_FTB(a, val, dv) do { const double dv_ = (dv); if (dv_ < 0.0) a = fmin(a, -tau * (val) / dv_); } while (0)
// same test with the reciprocal of the distance at hand: alpha <= tau / (-dv/val)
#define ASC_FTBR(amax_inv, ival, dv) amax_inv = fmax(amax_inv, -(dv) * (ival))

ASC_DEV double push_in(double v, double lb, double ub) {
  const double k1 = 1e-2;
  const double pl = fmin(k1 * fmax(1.0, fabs(lb)), k1 * (ub - lb));
  const double pu = fmin(k1 * fmax(1.0, fabs(ub)), k1 * (ub - lb));
  return fmin(fmax(v, lb + pl), ub - pu);
}


}  // namespace ascent
