// Persistent Hermite-Simpson kernel of the batched ascent NLP solver (gfx950): ascent_opts.scheme = 2 in the layout of
// ascent_persist.hip -- ONE launch per grid level, one wavefront owns four NLPs (or one: WIDE) for the whole interior-point solve.
//
// Hermite-Simpson in compressed form with the control held over the step (the report of the reference cites Kelly's tutorial as
// its method source, PDF p3/p25; dynamics /root/reference/Launch_Optimiser.py:114-136):
//     z_m = (z_{k-1}+z_k)/2 + h/8 [f(z_{k-1},u_k) - f(z_k,u_k)],     c_k = z_k - z_{k-1} - h/6 [f(z_{k-1}) + 4 f(z_m) + f(z_k)]
// Both step Jacobians are "dense" 7x7 and a step has a cross Hessian between its two nodes -- the dense-block path
// (ascent_dense.hip) therefore works on 8x8 blocks, one wavefront per NLP, ~1200 instructions per recursion step.  But the blocks
// are products of the SPARSE F = df/dz of the three evaluation points (eps = h^2/12, A = d(ax,ay)/d(x,y), a_. = d(ax,ay)/d(angle|mass)):
//     Jb = I - (h/6) Fb - (h/3) Fm + eps Fm Fb      rows (x,y):       [ I + eps A_b   | -(h/2) I     | eps a_angle^b | 0             | eps a_mass^b ]
//                                                    rows (xdot,ydot): [ -(h/6)A_b-(h/3)A_m | I + eps A_m | -(h/6)a_angle^b-(h/3)a_angle^m | eps a_angle^m | ..mass.. ]
//                                                    rows angle, angledot, mass: unit upper triangular, one entry -(h/2) at (angle, angledot)
//     Ja = -I - (h/6) Fa - (h/3) Fm - eps Fm Fa     (same pattern with the signs of the identity parts flipped)
// so that a solve with Jb' is two 2x2 inverses (of I + eps A_b and of the Schur complement of the (xdot,ydot) block) and a back
// substitution, -Ja' x is 34 multiply-adds, and the cross Hessian of a step is a RANK-4 term through the midpoint's (x, y, angle, mass):
//     1/2 dxi' W dxi,   dxi = La dz_{k-1} + Lb dz_k,   W = -(2h/3) Hessian of lambda_k'(ax,ay) at z_m,   La/Lb: rows of I/2 +- (h/8) F.
// The factorisation sweep keeps the 16-lanes-per-NLP layout of p_solve -- lanes 0-6 one column each of the 7x7 value function, lane 7
// the control's column of the step's 8x8 form in (dz_{k-1}, du_k), lanes 8-10 the three right-hand sides (residual, theta column,
// nu3 column), lanes 11-14 the four rows of Lb Jb^-1 [Ja Ju] that the rank-4 term needs -- which ride on the same instructions:
//     N = Jb^-T (P_k + Q_k) Jb^-1,   T = [Ja Ju]' N [Ja Ju] + Lam' W Lam + diag(0, R),   P_{k-1} = T_zz - T_zu T_uz / T_uu.
// tests/hs_structured.py restates this algorithm in numpy and checks it against a generic sparse LU of the full KKT matrix.
// The forward and adjoint sweeps are the affine recursions of p_solve with HS's node-local matrices (rows of -Jb^-1 Ja from four
// transposed solves per node); everything node-local is evaluated 12 nodes x 4 NLPs (or 48 nodes x 1) at a time and handed to the
// serial steps through LDS, never through HBM.
#include <hip/hip_runtime.h>

// (same reason as in ascent_persist.hip: every phase evaluates the step Jacobians again and must get the same bits)
#pragma clang fp contract(on)

#include <cstdio>
#include <cstdlib>

#include "ascent.h"
#include "ascent_device.hpp"
#include "ascent_tile.hpp"
#include "ascent_persist.hpp"
#include "ascent_persist_dev.hpp"

using namespace ascent;

namespace {

typedef __attribute__((address_space(3))) double ldbl;      // LDS double (the functions below take their pointers with address spaces:
                                                             // generic pointers would make every access a flat instruction)
constexpr int HCH = 12, HCW = 48;     // nodes per chunk: four NLPs x 12, or one NLP x 48 (the LDS stage has 48 columns)
constexpr int LDH = 49;               // row stride of the stage in doubles (odd: the rows a sweep step gathers hit different banks)
// stage rows of the factorisation phase (one chunk)
constexpr int H_GA = 0, H_GM = 8, H_GB = 16, H_EB = 24, H_ES = 28, H_W = 32, H_R0 = 42, H_C = 45, H_JT = 52, H_H = 59, H_RZ = 69,
              H_GT = 76, H_ROWS = 83;
// carry between chunks (descending order): what the last node of a chunk needs of the first step of the chunk above
constexpr int C_PSI = 0, C_GA = 7, C_HA = 14, C_WA = 21, C_OM = 23, C_TG = 27, C_N = 34;   // (C_TG: Ja'lambda of the TRIAL point)


// Der in LDS (16 doubles): the evaluations copy it into registers, the sweeps read single fields
constexpr int DER_N = 16;
ASC_DEV void der_store(double *p, const Der &d) {
  p[0] = d.rho0; p[1] = d.rhof; p[2] = d.vp2; p[3] = d.gam; p[4] = d.thr; p[5] = d.alpha; p[6] = d.mrate; p[7] = d.ms; p[8] = d.M0; p[9] = d.T;
  p[10] = d.aub; p[11] = d.tlb; p[12] = d.tub; p[13] = (double)d.term; p[14] = d.ht; p[15] = d.Et;
}
template <typename PT>
ASC_DEV Der der_load(PT p) {
  Der d;
  d.rho0 = p[0]; d.rhof = p[1]; d.vp2 = p[2]; d.gam = p[3]; d.thr = p[4]; d.alpha = p[5]; d.mrate = p[6]; d.ms = p[7]; d.M0 = p[8]; d.T = p[9];
  d.aub = p[10]; d.tlb = p[11]; d.tub = p[12]; d.term = (int)p[13]; d.ht = p[14]; d.Et = p[15];
  return d;
}

// ---- the three evaluation points of a step -------------------------------------------------------------------------------------
struct HsPts { double Ga[8], Gb[8], Gm[8], fa[7], fb[7], fm[7], Hm[10]; };
// HM: with the Hessian of lx*ax + ly*ay at the midpoint
template <int HM>
ASC_DEV void hs_points(const Der &d, const double *za, const double *zb, double u, double h, double lx, double ly, HsPts &p) {
  double ax, ay, zm[7];
  accel<1>(d, za[IX], za[IY], za[IA], za[IM], 0.0, 0.0, ax, ay, p.Ga, nullptr);
  rhs_f(d, za, u, ax, ay, p.fa);
  accel<1>(d, zb[IX], zb[IY], zb[IA], zb[IM], 0.0, 0.0, ax, ay, p.Gb, nullptr);
  rhs_f(d, zb, u, ax, ay, p.fb);
  const double e8 = 0.125 * h;
  ASC_UNROLL
  for (int i = 0; i < 7; i++) zm[i] = 0.5 * (za[i] + zb[i]) + e8 * (p.fa[i] - p.fb[i]);
  if constexpr (HM) accel<2>(d, zm[IX], zm[IY], zm[IA], zm[IM], lx, ly, ax, ay, p.Gm, p.Hm);
  else accel<1>(d, zm[IX], zm[IY], zm[IA], zm[IM], 0.0, 0.0, ax, ay, p.Gm, nullptr);
  rhs_f(d, zm, u, ax, ay, p.fm);
}
// defect and theta column of the step
ASC_DEV void hs_defect(const HsPts &p, const double *za, const double *zb, double h, double hT, double *c, double *Jth) {
  const double sm = (4.0 / 6.0) * h, h8 = 0.125 * hT;
  double mth[7], t7[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) mth[i] = h8 * (p.fa[i] - p.fb[i]);
  fz_mul(p.Gm, mth, t7);
  ASC_UNROLL
  for (int i = 0; i < 7; i++) {
    const double ws = (1.0 / 6.0) * (p.fa[i] + p.fb[i]) + (4.0 / 6.0) * p.fm[i];
    c[i] = zb[i] - za[i] - h * ws;
    Jth[i] = -hT * ws - sm * t7[i];
  }
}
// Ja'lambda, Jb'lambda and the weights of the two end-point Hessians (the (xdot, ydot) components of sa lam + sm e gm, sb lam - sm e gm)
struct HsDual { double gm[7], ga[7], gb[7], Fal[7], Fbl[7], Fag[7], Fbg[7], wa[2], wb[2]; };
ASC_DEV void hs_dual(const HsPts &p, const double *lam, double h, HsDual &q) {
  const double s6 = (1.0 / 6.0) * h, sm = (4.0 / 6.0) * h, e8 = 0.125 * h;
  fzt_lambda(p.Gm, lam, q.gm);
  fzt_lambda(p.Ga, lam, q.Fal); fzt_lambda(p.Gb, lam, q.Fbl); fzt_lambda(p.Ga, q.gm, q.Fag); fzt_lambda(p.Gb, q.gm, q.Fbg);
  ASC_UNROLL
  for (int i = 0; i < 7; i++) {
    const double Mag = 0.5 * q.gm[i] + e8 * q.Fag[i], Mbg = 0.5 * q.gm[i] - e8 * q.Fbg[i];
    q.ga[i] = -lam[i] - s6 * q.Fal[i] - sm * Mag;
    q.gb[i] = lam[i] - s6 * q.Fbl[i] - sm * Mbg;
  }
  q.wa[0] = s6 * lam[IVX] + sm * e8 * q.gm[IVX]; q.wa[1] = s6 * lam[IVY] + sm * e8 * q.gm[IVY];
  q.wb[0] = s6 * lam[IVX] - sm * e8 * q.gm[IVX]; q.wb[1] = s6 * lam[IVY] - sm * e8 * q.gm[IVY];
}
// the theta column's entries on the rows of the two nodes of the step, and the step's part of the (theta, theta) entry
ASC_DEV void hs_theta(const HsPts &p, const HsDual &q, double h, double hT, double *Hath, double *Hbth, double &Hthth) {
  const double sm = (4.0 / 6.0) * h, e8 = 0.125 * h, h8 = 0.125 * hT;
  const double mq[4] = {h8 * (p.fa[IX] - p.fb[IX]), h8 * (p.fa[IY] - p.fb[IY]), h8 * (p.fa[IA] - p.fb[IA]), h8 * (p.fa[IM] - p.fb[IM])};
  const double *H = p.Hm;      // xx xy xa xm yy ya ym aa am mm
  const double Hmm[4] = {H[0] * mq[0] + H[1] * mq[1] + H[2] * mq[2] + H[3] * mq[3], H[1] * mq[0] + H[4] * mq[1] + H[5] * mq[2] + H[6] * mq[3],
                         H[2] * mq[0] + H[5] * mq[1] + H[7] * mq[2] + H[8] * mq[3], H[3] * mq[0] + H[6] * mq[1] + H[8] * mq[2] + H[9] * mq[3]};
  // rows of Ma / Mb on (x, y, angle, mass): state index -> (index among the four, coefficient)
  const int qi[7] = {0, 1, 0, 1, 2, 2, 3};
  const double ca[7] = {0.5, 0.5, e8, e8, 0.5, e8, 0.5}, cb[7] = {0.5, 0.5, -e8, -e8, 0.5, -e8, 0.5};
  double gmm = 0.0;
  ASC_UNROLL
  for (int i = 0; i < 7; i++) {
    const double Mag = 0.5 * q.gm[i] + e8 * q.Fag[i], Mbg = 0.5 * q.gm[i] - e8 * q.Fbg[i];
    Hath[i] = -hT * ((1.0 / 6.0) * q.Fal[i] + (4.0 / 6.0) * Mag) - sm * (h8 * q.Fag[i] + ca[i] * Hmm[qi[i]]);
    Hbth[i] = -hT * ((1.0 / 6.0) * q.Fbl[i] + (4.0 / 6.0) * Mbg) - sm * (-h8 * q.Fbg[i] + cb[i] * Hmm[qi[i]]);
    gmm += q.gm[i] * (h8 * (p.fa[i] - p.fb[i]));
  }
  const double mHm = mq[0] * Hmm[0] + mq[1] * Hmm[1] + mq[2] * Hmm[2] + mq[3] * Hmm[3];
  Hthth = -2.0 * hT * (4.0 / 6.0) * gmm - sm * mHm;
}

// ---- the step Jacobians as operators ---------------------------------------------------------------------------------------------
// the two 2x2 inverses of a solve with Jb: eb = (I + eps A_b)^-1,  es = (I + eps A_m + (h/2) B21 eb)^-1,  B21 = -(h/6) A_b - (h/3) A_m
ASC_DEV void inv2(double m11, double m12, double m21, double m22, double *E) {
  const double idet = rcp(m11 * m22 - m12 * m21);
  E[0] = m22 * idet; E[1] = -m12 * idet; E[2] = -m21 * idet; E[3] = m11 * idet;
}
ASC_DEV void hs_blocks(const double *Gb, const double *Gm, double h, double *eb, double *es) {
  const double eps = (1.0 / 12.0) * (h * h), h6 = (1.0 / 6.0) * h, h3 = (1.0 / 3.0) * h, hh = 0.5 * h;
  inv2(1.0 + eps * Gb[0], eps * Gb[1], eps * Gb[4], 1.0 + eps * Gb[5], eb);
  const double b0 = -h6 * Gb[0] - h3 * Gm[0], b1 = -h6 * Gb[1] - h3 * Gm[1], b2 = -h6 * Gb[4] - h3 * Gm[4], b3 = -h6 * Gb[5] - h3 * Gm[5];
  // B21 eb
  const double p0 = b0 * eb[0] + b1 * eb[2], p1 = b0 * eb[1] + b1 * eb[3], p2 = b2 * eb[0] + b3 * eb[2], p3 = b2 * eb[1] + b3 * eb[3];
  inv2(1.0 + eps * Gm[0] + hh * p0, eps * Gm[1] + hh * p1, eps * Gm[4] + hh * p2, 1.0 + eps * Gm[5] + hh * p3, es);
}
struct HsJ {
  double eb[4], es[4], b21[4], caq[2], cav[2], cwv[2], cmq[2], cmv[2];      // Jb
  double eAa[4], b21a[4], eAm[4], eaa[2], cava[2], ema[2], cmva[2];        // Ja
  double hh, bu;
};
ASC_DEV void hs_coeffs(const double *Ga, const double *Gm, const double *Gb, const double *eb, const double *es, double h, double bu, HsJ &J) {
  const double eps = (1.0 / 12.0) * (h * h), h6 = (1.0 / 6.0) * h, h3 = (1.0 / 3.0) * h;
  ASC_UNROLL
  for (int i = 0; i < 4; i++) { J.eb[i] = eb[i]; J.es[i] = es[i]; }
  constexpr int ai[4] = {0, 1, 4, 5};
  ASC_UNROLL
  for (int i = 0; i < 4; i++) {
    J.b21[i] = -h6 * Gb[ai[i]] - h3 * Gm[ai[i]];
    J.b21a[i] = -h6 * Ga[ai[i]] - h3 * Gm[ai[i]];
    J.eAa[i] = eps * Ga[ai[i]];
    J.eAm[i] = eps * Gm[ai[i]];
  }
  ASC_UNROLL
  for (int i = 0; i < 2; i++) {
    const int ia = 2 + 4 * i, im = 3 + 4 * i;
    J.caq[i] = eps * Gb[ia]; J.cav[i] = -h6 * Gb[ia] - h3 * Gm[ia]; J.cwv[i] = eps * Gm[ia];
    J.cmq[i] = eps * Gb[im]; J.cmv[i] = -h6 * Gb[im] - h3 * Gm[im];
    J.eaa[i] = eps * Ga[ia]; J.cava[i] = -h6 * Ga[ia] - h3 * Gm[ia];
    J.ema[i] = eps * Ga[im]; J.cmva[i] = -h6 * Ga[im] - h3 * Gm[im];
  }
  J.hh = 0.5 * h; J.bu = bu;
}
// x with Jb' x = a
ASC_DEV void hs_solve_jbt(const HsJ &J, const double *a, double *x) {
  const double t1x = J.eb[0] * a[IX] + J.eb[2] * a[IY], t1y = J.eb[1] * a[IX] + J.eb[3] * a[IY];
  const double r2x = a[IVX] + J.hh * t1x, r2y = a[IVY] + J.hh * t1y;
  const double x2x = J.es[0] * r2x + J.es[2] * r2y, x2y = J.es[1] * r2x + J.es[3] * r2y;
  const double ux = J.b21[0] * x2x + J.b21[2] * x2y, uy = J.b21[1] * x2x + J.b21[3] * x2y;
  const double x1x = t1x - (J.eb[0] * ux + J.eb[2] * uy), x1y = t1y - (J.eb[1] * ux + J.eb[3] * uy);
  const double ra = a[IA] - (J.caq[0] * x1x + J.caq[1] * x1y) - (J.cav[0] * x2x + J.cav[1] * x2y);
  const double rw = a[IW] - (J.cwv[0] * x2x + J.cwv[1] * x2y);
  const double rm = a[IM] - (J.cmq[0] * x1x + J.cmq[1] * x1y) - (J.cmv[0] * x2x + J.cmv[1] * x2y);
  x[IX] = x1x; x[IY] = x1y; x[IVX] = x2x; x[IVY] = x2y; x[IA] = ra; x[IW] = rw + J.hh * ra; x[IM] = rm;
}
// o = -[Ja Ju]' x  (o[0..6]: the states of node k-1; o[7]: the control of the step)
ASC_DEV void hs_apply_j8t(const HsJ &J, const double *x, double *o) {
  o[IX] = x[IX] + (J.eAa[0] * x[IX] + J.eAa[2] * x[IY]) - (J.b21a[0] * x[IVX] + J.b21a[2] * x[IVY]);
  o[IY] = x[IY] + (J.eAa[1] * x[IX] + J.eAa[3] * x[IY]) - (J.b21a[1] * x[IVX] + J.b21a[3] * x[IVY]);
  o[IVX] = J.hh * x[IX] + x[IVX] + (J.eAm[0] * x[IVX] + J.eAm[2] * x[IVY]);
  o[IVY] = J.hh * x[IY] + x[IVY] + (J.eAm[1] * x[IVX] + J.eAm[3] * x[IVY]);
  o[IA] = (J.eaa[0] * x[IX] + J.eaa[1] * x[IY]) - (J.cava[0] * x[IVX] + J.cava[1] * x[IVY]) + x[IA];
  o[IW] = (J.cwv[0] * x[IVX] + J.cwv[1] * x[IVY]) + J.hh * x[IA] + x[IW];
  o[IM] = (J.ema[0] * x[IX] + J.ema[1] * x[IY]) - (J.cmva[0] * x[IVX] + J.cmva[1] * x[IVY]) + x[IM];
  o[7] = J.bu * x[IW];
}
// Rows x, y, xdot, ydot of Jb^-1 (X[i] = Jb^-T e_i) and of Abar = -Jb^-1 Ja (the rows angle, angledot, mass are e_a + (h/2) e_w, e_w, e_m
// and e_a + h e_w, e_w, e_m)
ASC_DEV void hs_rows(const HsJ &J, double X[4][7], double Ab[4][7]) {
  ASC_UNROLL
  for (int r = 0; r < 4; r++) {
    double e[7], o[8];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) e[i] = i == r ? 1.0 : 0.0;
    hs_solve_jbt(J, e, X[r]);
    hs_apply_j8t(J, X[r], o);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) Ab[r][i] = o[i];
  }
}

// ---- the trial point of one chunk ----------------------------------------------------------------------------------------------------
// The trial point x + alpha dx at the nodes of a chunk (iterate ic, step stp), stored into the other iterate buffer, and its pieces of the
// l1 merit function and of the KKT error.  The stationarity row of node k holds Jb_k'lambda_k + Ja_{k+1}'lambda_{k+1}: the second term
// comes from the lane of step k+1, or -- at the end of a chunk -- from the carry the chunk above has left (chunks run downwards).
// (A function of its own -- one register allocation per node-parallel evaluation, see hs_eval_factor below; the scalars of the trial point
//  come from the NLP's record in LDS: iterate X_S, step X_D.)
struct HsTrial { double alpha, adu, mu, dt, hT; bool first; };
#ifdef PERSIST_PROFILE
__device__ unsigned long long g_hsprof[8];
#define HPROF(i_) do { __builtin_amdgcn_sched_barrier(0); const long long t1_ = clock64(); if (blockIdx.x == 0 && threadIdx.x == 0) g_hsprof[i_] += t1_ - hp_; hp_ = t1_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define HPROF_DECL long long hp_ = clock64();
#else
#define HPROF(i_) do { } while (0)
#define HPROF_DECL
#endif
template <int TERM>
ASC_PASS Part hs_trial_chunk(const ldbl *dp, const ldbl *sc, int K, int Kp, int k, bool on, bool lastl, bool firstl, const gdbl *ic, const gdbl *stp,
                             gdbl *in, HsTrial t, bool live, ldbl *carry, Part P) {
  HPROF_DECL
  const Der d = der_load(dp);
  const double be = t.dt * d.alpha, mlo = t.mu * 1e-10, mhi = t.mu * 1e10;
  double ga[7] = {0, 0, 0, 0, 0, 0, 0}, r[7] = {0, 0, 0, 0, 0, 0, 0};
  if (on) {
    const double alpha = t.alpha;
    const int km = k > 0 ? k - 1 : 0;
    double z[7], zp[7], l[7], zb[6];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      z[i] = ic[(O_Z + i) * Kp + k]; l[i] = ic[(O_L + i) * Kp + k];
      const double pv = ic[(O_Z + i) * Kp + km];      // (node k-1 by a clamped index and a select: a conditional load is a branch and a wait each)
      zp[i] = k > 0 ? pv : 0.0;
    }
    double u = ic[O_U * Kp + k];
    ASC_UNROLL
    for (int b = 0; b < 6; b++) zb[b] = ic[(O_ZB + b) * Kp + k];
    double dzb[6] = {0, 0, 0, 0, 0, 0};
    if (!t.first) {
      ASC_UNROLL
      for (int i = 0; i < 7; i++) {
        z[i] += alpha * stp[(O_Z + i) * Kp + k];
        const double dpv = stp[(O_Z + i) * Kp + km];
        zp[i] += k > 0 ? alpha * dpv : 0.0;
        l[i] += alpha * stp[(O_L + i) * Kp + k];
      }
      u += alpha * stp[O_U * Kp + k];
      ASC_UNROLL
      for (int b = 0; b < 6; b++) dzb[b] = stp[(O_ZB + b) * Kp + k];
    }
    const double dist[6] = {z[IA], d.aub - z[IA], z[IM], 1.0 - z[IM], u + 1.0, 1.0 - u};
    ASC_UNROLL
    for (int b = 0; b < 6; b++) {
      const double id = rcp(dist[b]);
      zb[b] = t.first ? zb[b] : fmin(fmax(zb[b] + t.adu * dzb[b], mlo * id), mhi * id);
    }
    if (live) {
      ASC_UNROLL
      for (int i = 0; i < 7; i++) { in[(O_Z + i) * Kp + k] = z[i]; in[(O_L + i) * Kp + k] = l[i]; }
      in[O_U * Kp + k] = u;
      ASC_UNROLL
      for (int b = 0; b < 6; b++) in[(O_ZB + b) * Kp + k] = zb[b];
    }
    HPROF(0);
    HsPts pt;
    hs_points<0>(d, zp, z, u, t.dt, 0.0, 0.0, pt);
    HPROF(1);
    double c[7], Jth[7];
    hs_defect(pt, zp, z, t.dt, t.hT, c, Jth);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      P.c1 += fabs(c[i]);
      P.cinf = fmax(P.cinf, fabs(c[i]));
      P.rth += Jth[i] * l[i];
      P.l1 += fabs(l[i]);
    }
    HPROF(2);
    HsDual q;
    hs_dual(pt, l, t.dt, q);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { ga[i] = q.ga[i]; r[i] = q.gb[i]; }
    HPROF(3);
    r[IA] += zb[1] - zb[0];
    r[IM] += zb[3] - zb[2];
    if (k == K - 1) {
      const Scal stt = trial_scal(d, lds_scal(sc, X_S), lds_scal(sc, X_D), t.alpha, t.adu, t.mu, t.first);
      const Terminal tt = TERM == 2 ? terminal_eval_any(d, z) : terminal_eval(d, z);
      const double e1 = fabs(tt.e3), e2 = fabs(tt.g1 - stt.s1), e3 = fabs(tt.g2 - stt.s2);
      P.cinf = fmax(P.cinf, fmax(e1, fmax(e2, e3)));
      P.c1 += e1 + e2 + e3;
      const double ps = ((stt.th - d.tlb) * (d.tub - stt.th)) * (stt.s1 * stt.s2);
      P.sl += ps > 0.0 ? log(ps) : NAN;
      if constexpr (TERM == 2) {
        double g4[4];
        terminal_grad_any(tt, stt.nu1, stt.nu2, g4);
        r[IX] += g4[0]; r[IY] += g4[1]; r[IVX] += g4[2]; r[IVY] += g4[3];
      } else {
        r[IX] += stt.nu3 * tt.e3g[0] + stt.nu1 * tt.g1g[0];
        r[IY] += stt.nu3 * tt.e3g[1] + stt.nu1 * tt.g1g[1];
        r[IVX] += stt.nu3 * tt.e3g[2] + stt.nu2 * tt.g2g[0];
        r[IVY] += stt.nu3 * tt.e3g[3] + stt.nu2 * tt.g2g[1];
      }
    }
    ASC_UNROLL
    for (int b = 0; b < 6; b++) { const double pr = dist[b] * zb[b]; P.pmin = fmin(P.pmin, pr); P.pmax = fmax(P.pmax, pr); P.zsum += zb[b]; }
    const double pa = dist[0] * dist[1], pm = dist[2] * dist[3], pu = dist[4] * dist[5];
    P.sl += (pa > 0.0 && pm > 0.0 && pu > 0.0) ? log(pa * pm * pu) : NAN;
    const double ruv = -be * l[IW] - zb[4] + zb[5];
    P.rd = fmax(P.rd, fabs(ruv));
  }
  HPROF(4);
  double gn[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) gn[i] = __shfl_down(ga[i], 1);
  if (on) {
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      const double g1 = k == K - 1 ? 0.0 : lastl ? carry[C_TG + i] : gn[i];
      P.rd = fmax(P.rd, fabs(r[i] + g1));
    }
  }
  wsync();
  if (on && firstl) {
    ASC_UNROLL
    for (int i = 0; i < 7; i++) carry[C_TG + i] = ga[i];
  }
  wsync();
  HPROF(5);
  return P;
}

// ---- node-parallel evaluations as functions ---------------------------------------------------------------------------------------
// Each evaluation below keeps 150-200 doubles alive (three evaluation points with first derivatives, the midpoint's Hessian, four
// products with F', both Jacobians' coefficients); inlined into h_solve next to the state the serial sweeps carry across a chunk they
// exceed the 512 registers of a wavefront, and the spills came back one load and one wait at a time (30 dependent scratch round trips per
// evaluation: more stall than arithmetic).  As functions they have a register allocation of their own and the caller's state is saved
// around the call in one burst.  Lane = node k (step k: nodes k-1, k); lastl / firstl: the last / first node of the chunk.

// Factorisation phase: the step's blocks into the LDS stage (column col).  Returns the step's part of the (theta, theta) entry.
ASC_PASS double hs_eval_factor(const ldbl *dp, const gdbl *it, ldbl *stage, ldbl *carry, int K, int Kp, int k, bool on, bool lastl, bool firstl,
                               int col, double h, double hT, double mu, double dw) {
  const Der d = der_load(dp);
  const double bu = h * d.alpha;
  HsPts pt;
  HsDual q;
  double Hath[7] = {0, 0, 0, 0, 0, 0, 0}, Hbth[7], zb6[6], z[7], u = 0.0, lw = 0.0, Hthth = 0.0;
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { q.ga[i] = 0.0; }
  q.wa[0] = 0.0; q.wa[1] = 0.0;
  if (on) {
    double zp[7], l[7];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      z[i] = it[(O_Z + i) * Kp + k]; l[i] = it[(O_L + i) * Kp + k];
      const double pv = it[(O_Z + i) * Kp + (k > 0 ? k - 1 : 0)];
      zp[i] = k > 0 ? pv : 0.0;
    }
    u = it[O_U * Kp + k]; lw = l[IW];
    ASC_UNROLL
    for (int b = 0; b < 6; b++) zb6[b] = it[(O_ZB + b) * Kp + k];
    hs_points<1>(d, zp, z, u, h, l[IVX], l[IVY], pt);
    hs_dual(pt, l, h, q);
    double cc[7], Jth[7];
    hs_defect(pt, zp, z, h, hT, cc, Jth);
    hs_theta(pt, q, h, hT, Hath, Hbth, Hthth);
    double eb[4], es[4];
    hs_blocks(pt.Gb, pt.Gm, h, eb, es);
    const double sm = (4.0 / 6.0) * h;
    ASC_UNROLL
    for (int i = 0; i < 8; i++) {
      stage[(H_GA + i) * LDH + col] = pt.Ga[i]; stage[(H_GM + i) * LDH + col] = pt.Gm[i]; stage[(H_GB + i) * LDH + col] = pt.Gb[i];
    }
    ASC_UNROLL
    for (int i = 0; i < 4; i++) { stage[(H_EB + i) * LDH + col] = eb[i]; stage[(H_ES + i) * LDH + col] = es[i]; }
    ASC_UNROLL
    for (int i = 0; i < 10; i++) stage[(H_W + i) * LDH + col] = -sm * pt.Hm[i];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { stage[(H_C + i) * LDH + col] = cc[i]; stage[(H_JT + i) * LDH + col] = Jth[i]; }
  }
  // from the lane of step k+1 (or the carry of the chunk above): Ja'lambda, the theta column's entries on node k, the end-point weights
  double gan[7], han[7], wan[2];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { gan[i] = __shfl_down(q.ga[i], 1); han[i] = __shfl_down(Hath[i], 1); }
  wan[0] = __shfl_down(q.wa[0], 1); wan[1] = __shfl_down(q.wa[1], 1);
  if (on) {
    if (k == K - 1) {
      ASC_UNROLL
      for (int i = 0; i < 7; i++) { gan[i] = 0.0; han[i] = 0.0; }
      wan[0] = 0.0; wan[1] = 0.0;
    } else if (lastl) {
      ASC_UNROLL
      for (int i = 0; i < 7; i++) { gan[i] = carry[C_GA + i]; han[i] = carry[C_HA + i]; }
      wan[0] = carry[C_WA]; wan[1] = carry[C_WA + 1];
    }
    double Hn[10], Gx[8], t1, t2;
    accel<2>(d, z[IX], z[IY], z[IA], z[IM], q.wb[0] + wan[0], q.wb[1] + wan[1], t1, t2, Gx, Hn);
    const double dist[6] = {z[IA], d.aub - z[IA], z[IM], 1.0 - z[IM], u + 1.0, 1.0 - u};
    double id[6];
    ASC_UNROLL
    for (int b = 0; b < 6; b++) id[b] = rcp(dist[b]);
    ASC_UNROLL
    for (int i = 0; i < 10; i++) Hn[i] = -Hn[i];
    Hn[7] += zb6[0] * id[0] + zb6[1] * id[1];
    Hn[9] += zb6[2] * id[2] + zb6[3] * id[3];
    ASC_UNROLL
    for (int i = 0; i < 10; i++) stage[(H_H + i) * LDH + col] = Hn[i];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      double rz = q.gb[i] + gan[i];
      if (i == IA) rz += mu * (id[1] - id[0]);
      if (i == IM) rz += mu * (id[3] - id[2]);
      stage[(H_RZ + i) * LDH + col] = rz;
      stage[(H_GT + i) * LDH + col] = Hbth[i] + han[i];
    }
    stage[H_R0 * LDH + col] = zb6[4] * id[4] + zb6[5] * id[5] + dw;
    stage[(H_R0 + 1) * LDH + col] = -bu * lw + mu * (id[5] - id[4]);
    stage[(H_R0 + 2) * LDH + col] = -hT * d.alpha * lw;
  }
  wsync();
  if (on && firstl) {
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { carry[C_GA + i] = q.ga[i]; carry[C_HA + i] = Hath[i]; }
    carry[C_WA] = q.wa[0]; carry[C_WA + 1] = q.wa[1];
  }
  wsync();
  return Hthth;
}

// Forward phase: the node-local coefficients of dz_k = M_k dz_{k-1} + v_k (rows 7 i .. 7 i + 5: M[i][0..5], row 7 i + 6: v[i]; lane 6: du) into the
// stage, the mass component by a prefix sum over the nodes (row F_OUT + 6).  carry_m: dz_m of the last node of the chunk before; returns the new one.
constexpr int F_OUT = 49, A_RHS = 42, A_C = 49, A_J = 56;      // (adjoint: rhs, defect and the coefficients of Jb stashed beside the recursion's rows)
template <int WIDE>
ASC_PASS double hs_eval_forward(const ldbl *dp, const gdbl *it, const gdbl *gains, ldbl *stage, int K, int Kp, int kn, bool on, int nl, int col,
                                double h, double hT, double dth, double dnu3, double carry_m) {
  const Der d = der_load(dp);
  const double bu = h * d.alpha;
  double x0m = 0.0, du00 = 0.0, ka[7], rc[7];
  HsJ J;
  if (on) {
    double z[7], zp[7];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      z[i] = it[(O_Z + i) * Kp + kn];
      const double pv = it[(O_Z + i) * Kp + (kn > 0 ? kn - 1 : 0)];
      zp[i] = kn > 0 ? pv : 0.0;
    }
    const double u_ = it[O_U * Kp + kn];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) ka[i] = gains[(size_t)i * Kp + kn];
    du00 = -(gains[(size_t)7 * Kp + kn] + gains[(size_t)8 * Kp + kn] * dth + gains[(size_t)9 * Kp + kn] * dnu3);
    HsPts pt;
    hs_points<0>(d, zp, z, u_, h, 0.0, 0.0, pt);
    double cc[7], Jth[7], eb[4], es[4];
    hs_defect(pt, zp, z, h, hT, cc, Jth);
    hs_blocks(pt.Gb, pt.Gm, h, eb, es);
    hs_coeffs(pt.Ga, pt.Gm, pt.Gb, eb, es, h, bu, J);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) rc[i] = cc[i] + Jth[i] * dth;
    x0m = -rc[IM];
  }
  // dz_m: inclusive prefix sum of x0_m over the nodes of the NLP (the chunk here, the chunks before in carry_m)
  double incl = x0m;
  ASC_UNROLL
  for (int sft = 1; sft < (WIDE ? 64 : 16); sft *= 2) {
    const double t = __shfl_up(incl, sft, WIDE ? 64 : 16);
    if (nl >= sft) incl += t;
  }
  const double dzm_k = carry_m + incl, dzm_p = dzm_k - x0m;
  carry_m += WIDE ? __shfl(incl, HCW - 1) : bcast16<HCH - 1>(incl);
  if (on) {
    constexpr int NC = 6, FS = 7;
    const double du00p = du00 - ka[IM] * dzm_p;
    double X[4][7], Ab[4][7];
    hs_rows(J, X, Ab);
    auto emit = [&](int i, const double *xr_, const double *ar) {       // row i: its row of Jb^-1 and of Abar
      const double bw = bu * xr_[IW];
      double v = bw * du00p + ar[IM] * dzm_p;
      ASC_UNROLL
      for (int j = 0; j < 7; j++) v -= xr_[j] * rc[j];
      ASC_UNROLL
      for (int j = 0; j < 6; j++) stage[(FS * i + j) * LDH + col] = ar[j] - bw * ka[j];
      stage[(FS * i + NC) * LDH + col] = v;
    };
    emit(IX, X[0], Ab[0]); emit(IY, X[1], Ab[1]); emit(IVX, X[2], Ab[2]); emit(IVY, X[3], Ab[3]);
    {
      const double xa[7] = {0, 0, 0, 0, 1.0, 0.5 * h, 0}, aa[7] = {0, 0, 0, 0, 1.0, h, 0};
      emit(IA, xa, aa);
      const double xw[7] = {0, 0, 0, 0, 0, 1.0, 0}, aw[7] = {0, 0, 0, 0, 0, 1.0, 0};
      emit(IW, xw, aw);
    }
    ASC_UNROLL
    for (int j = 0; j < 6; j++) stage[(42 + j) * LDH + col] = -ka[j];
    stage[48 * LDH + col] = du00p;
    stage[(F_OUT + 6) * LDH + col] = dzm_k;
  }
  wsync();
  return carry_m;
}

// Adjoint phase: rhs_k (everything of node k's stationarity row that the forward sweep has fixed), the coefficients of
// psi_{k-1} = Abar_k' (psi_k - rhs_k) into rows 6 i .. 6 i + 5 of the stage, and -- for the part after the sweep -- rhs, the defect and the
// coefficients of Jb into rows A_RHS / A_C / A_J of the lane's column.  Returns c . lambda of the node.
template <int TERM>
ASC_PASS double hs_eval_adjoint(const ldbl *dp, const ldbl *sc, const gdbl *it, const gdbl *stp, ldbl *stage, ldbl *carry, int K, int Kp, int kn,
                                bool on, bool lastl, bool firstl, int col, double h, double hT, double mu, double dw, double dth, double dnu3) {
  const Der d = der_load(dp);
  const double bu = h * d.alpha, e8 = 0.125 * h;
  double ccl = 0.0, ccn[7], rhs[7] = {0, 0, 0, 0, 0, 0, 0};
  double om[4] = {0, 0, 0, 0}, Hath[7] = {0, 0, 0, 0, 0, 0, 0}, Hbth[7];
  HsJ J;
  HsPts pt;
  HsDual q;
  ASC_UNROLL
  for (int i = 0; i < 7; i++) q.ga[i] = 0.0;
  q.wa[0] = 0.0; q.wa[1] = 0.0;
  double z[7], dz[7], zb6[4];
  if (on) {
    double zp[7], l[7], dzp[7];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      const int km = kn > 0 ? kn - 1 : 0;
      z[i] = it[(O_Z + i) * Kp + kn]; l[i] = it[(O_L + i) * Kp + kn]; dz[i] = stp[(O_Z + i) * Kp + kn];
      const double pv = it[(O_Z + i) * Kp + km], dpv = stp[(O_Z + i) * Kp + km];
      zp[i] = kn > 0 ? pv : 0.0; dzp[i] = kn > 0 ? dpv : 0.0;
    }
    const double u = it[O_U * Kp + kn];
    ASC_UNROLL
    for (int b = 0; b < 4; b++) zb6[b] = it[(O_ZB + b) * Kp + kn];
    hs_points<1>(d, zp, z, u, h, l[IVX], l[IVY], pt);
    hs_dual(pt, l, h, q);
    double Jth[7], Hthth, eb[4], es[4];
    hs_defect(pt, zp, z, h, hT, ccn, Jth);
    hs_theta(pt, q, h, hT, Hath, Hbth, Hthth);
    hs_blocks(pt.Gb, pt.Gm, h, eb, es);
    hs_coeffs(pt.Ga, pt.Gm, pt.Gb, eb, es, h, bu, J);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) ccl += ccn[i] * l[i];
    // omega = W (La dz_{k-1} + Lb dz_k) on (x, y, angle, mass)
    const double xi[4] = {0.5 * (dzp[IX] + dz[IX]) + e8 * (dzp[IVX] - dz[IVX]), 0.5 * (dzp[IY] + dz[IY]) + e8 * (dzp[IVY] - dz[IVY]),
                          0.5 * (dzp[IA] + dz[IA]) + e8 * (dzp[IW] - dz[IW]), 0.5 * (dzp[IM] + dz[IM])};
    const double sm = (4.0 / 6.0) * h;
    const double *H = pt.Hm;
    om[0] = -sm * (H[0] * xi[0] + H[1] * xi[1] + H[2] * xi[2] + H[3] * xi[3]);
    om[1] = -sm * (H[1] * xi[0] + H[4] * xi[1] + H[5] * xi[2] + H[6] * xi[3]);
    om[2] = -sm * (H[2] * xi[0] + H[5] * xi[1] + H[7] * xi[2] + H[8] * xi[3]);
    om[3] = -sm * (H[3] * xi[0] + H[6] * xi[1] + H[8] * xi[2] + H[9] * xi[3]);
  }
  double gan[7], han[7], wan[2], omn[4];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { gan[i] = __shfl_down(q.ga[i], 1); han[i] = __shfl_down(Hath[i], 1); }
  wan[0] = __shfl_down(q.wa[0], 1); wan[1] = __shfl_down(q.wa[1], 1);
  ASC_UNROLL
  for (int r = 0; r < 4; r++) omn[r] = __shfl_down(om[r], 1);
  if (on) {
    if (kn == K - 1) {
      ASC_UNROLL
      for (int i = 0; i < 7; i++) { gan[i] = 0.0; han[i] = 0.0; }
      wan[0] = 0.0; wan[1] = 0.0;
      ASC_UNROLL
      for (int r = 0; r < 4; r++) omn[r] = 0.0;
    } else if (lastl) {
      ASC_UNROLL
      for (int i = 0; i < 7; i++) { gan[i] = carry[C_GA + i]; han[i] = carry[C_HA + i]; }
      wan[0] = carry[C_WA]; wan[1] = carry[C_WA + 1];
      ASC_UNROLL
      for (int r = 0; r < 4; r++) omn[r] = carry[C_OM + r];
    }
    double Hn[10], Gx[8], t1, t2;
    accel<2>(d, z[IX], z[IY], z[IA], z[IM], q.wb[0] + wan[0], q.wb[1] + wan[1], t1, t2, Gx, Hn);
    const double id0 = rcp(z[IA]), id1 = rcp(d.aub - z[IA]), id2 = rcp(z[IM]), id3 = rcp(1.0 - z[IM]);
    ASC_UNROLL
    for (int i = 0; i < 10; i++) Hn[i] = -Hn[i];
    Hn[7] += zb6[0] * id0 + zb6[1] * id1;
    Hn[9] += zb6[2] * id2 + zb6[3] * id3;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) rhs[i] = (q.gb[i] + gan[i]) + (Hbth[i] + han[i]) * dth + dw * dz[i];
    rhs[IA] += mu * (id1 - id0);
    rhs[IM] += mu * (id3 - id2);
    rhs[IX] += Hn[0] * dz[IX] + Hn[1] * dz[IY] + Hn[2] * dz[IA] + Hn[3] * dz[IM];
    rhs[IY] += Hn[1] * dz[IX] + Hn[4] * dz[IY] + Hn[5] * dz[IA] + Hn[6] * dz[IM];
    rhs[IA] += Hn[2] * dz[IX] + Hn[5] * dz[IY] + Hn[7] * dz[IA] + Hn[8] * dz[IM];
    rhs[IM] += Hn[3] * dz[IX] + Hn[6] * dz[IY] + Hn[8] * dz[IA] + Hn[9] * dz[IM];
    // the midpoint terms: Lb_k' omega_k + La_{k+1}' omega_{k+1}
    rhs[IX] += 0.5 * (om[0] + omn[0]); rhs[IY] += 0.5 * (om[1] + omn[1]); rhs[IA] += 0.5 * (om[2] + omn[2]); rhs[IM] += 0.5 * (om[3] + omn[3]);
    rhs[IVX] += e8 * (omn[0] - om[0]); rhs[IVY] += e8 * (omn[1] - om[1]); rhs[IW] += e8 * (omn[2] - om[2]);
    if (kn == K - 1) {
      const Scal s = lds_scal(sc, X_S);
      const double sig1 = sc[X_SIG1], sig2 = sc[X_SIG2], rs1 = sc[X_RS1], rs2 = sc[X_RS2];
      double QT[28], qd[7];
      const Terminal tm = TERM == 2 ? terminal_eval_any(d, z) : terminal_eval(d, z);
      ASC_UNROLL
      for (int i = 0; i < 28; i++) QT[i] = 0.0;
      if constexpr (TERM == 2) terminal_hessian_any(QT, tm, s.nu1, s.nu2, sig1, sig2);
      else terminal_hessian(QT, tm, s.nu3, s.nu1, s.nu2, sig1, sig2);
      symv(QT, dz, qd);
      ASC_UNROLL
      for (int i = 0; i < 7; i++) rhs[i] += qd[i];
      const double w1 = s.nu1 + sig1 * sc[X_CG1] + rs1, w2 = s.nu2 + sig2 * sc[X_CG2] + rs2;
      if constexpr (TERM == 2) {
        double g4[4];
        terminal_grad_any(tm, w1, w2, g4);
        rhs[IX] += g4[0]; rhs[IY] += g4[1]; rhs[IVX] += g4[2]; rhs[IVY] += g4[3];
      } else {
        rhs[IX] += s.nu3 * tm.e3g[0] + w1 * tm.g1g[0] + tm.e3g[0] * dnu3;
        rhs[IY] += s.nu3 * tm.e3g[1] + w1 * tm.g1g[1] + tm.e3g[1] * dnu3;
        rhs[IVX] += s.nu3 * tm.e3g[2] + w2 * tm.g2g[0] + tm.e3g[2] * dnu3;
        rhs[IVY] += s.nu3 * tm.e3g[3] + w2 * tm.g2g[1] + tm.e3g[3] * dnu3;
      }
    }
    // coefficients: column j = x, y, xdot, ydot of Abar' is row j of Abar; column angle is e_a + h e_w; w = -Abar' rhs = Ja' Jb^-T rhs
    double X[4][7], Ab[4][7];
    hs_rows(J, X, Ab);
    double xs[7], o8[8];
    hs_solve_jbt(J, rhs, xs);
    hs_apply_j8t(J, xs, o8);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      ASC_UNROLL
      for (int j = 0; j < 4; j++) stage[(6 * i + j) * LDH + col] = Ab[j][i];
      stage[(6 * i + 4) * LDH + col] = i == IA ? 1.0 : i == IW ? h : 0.0;
      stage[(6 * i + 5) * LDH + col] = -o8[i];
      stage[(A_RHS + i) * LDH + col] = rhs[i];
      stage[(A_C + i) * LDH + col] = ccn[i];
    }
    const double jc[22] = {J.eb[0], J.eb[1], J.eb[2], J.eb[3], J.es[0], J.es[1], J.es[2], J.es[3], J.b21[0], J.b21[1], J.b21[2], J.b21[3],
                           J.caq[0], J.caq[1], J.cav[0], J.cav[1], J.cwv[0], J.cwv[1], J.cmq[0], J.cmq[1], J.cmv[0], J.cmv[1]};
    ASC_UNROLL
    for (int i = 0; i < 22; i++) stage[(A_J + i) * LDH + col] = jc[i];
  }
  wsync();
  if (on && firstl) {
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { carry[C_GA + i] = q.ga[i]; carry[C_HA + i] = Hath[i]; }
    carry[C_WA] = q.wa[0]; carry[C_WA + 1] = q.wa[1];
    ASC_UNROLL
    for (int r = 0; r < 4; r++) carry[C_OM + r] = om[r];
  }
  return ccl;
}
// ... and after the sweep: Jb_k' dlam_k = psi_k - rhs_k (psi: rows 0-6 of the next column, or the carry at the end of a chunk); returns c . dlambda
ASC_PASS double hs_post_adjoint(ldbl *stage, const ldbl *carry, gdbl *stp, int K, int Kp, int kn, bool on, bool lastl, int col, double h, bool live) {
  double ccl = 0.0;
  if (on) {
    HsJ J;
    double jc[22];
    ASC_UNROLL
    for (int i = 0; i < 22; i++) jc[i] = stage[(A_J + i) * LDH + col];
    ASC_UNROLL
    for (int i = 0; i < 4; i++) { J.eb[i] = jc[i]; J.es[i] = jc[4 + i]; J.b21[i] = jc[8 + i]; }
    ASC_UNROLL
    for (int i = 0; i < 2; i++) { J.caq[i] = jc[12 + i]; J.cav[i] = jc[14 + i]; J.cwv[i] = jc[16 + i]; J.cmq[i] = jc[18 + i]; J.cmv[i] = jc[20 + i]; }
    J.hh = 0.5 * h;
    double ph[7], dl[7];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      const double psi = kn + 1 < K ? (lastl ? carry[C_PSI + i] : stage[i * LDH + col + 1]) : 0.0;
      ph[i] = psi - stage[(A_RHS + i) * LDH + col];
    }
    hs_solve_jbt(J, ph, dl);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) ccl += stage[(A_C + i) * LDH + col] * dl[i];
    if (live) {
      ASC_UNROLL
      for (int i = 0; i < 7; i++) stp[(O_L + i) * Kp + kn] = dl[i];
    }
  }
  wsync();
  return ccl;
}

// ==============================================================================================================
// h_solve: the whole interior-point loop of one grid level (scheme 2)
// ==============================================================================================================
template <int TERM, int WIDE>
__global__ __launch_bounds__(WAVE) void h_solve(const ascent_params *params, long batch, PGeo g, double *ws, int max_iter, double tol) {
  using L = Lay<0>;
  constexpr int NIT = L::NIT, R_ST = L::R_ST, R_KA = L::R_KA, R_K0 = L::R_K0, NROWS = L::NROWS;
  __shared__ double stage[H_ROWS * LDH];
  __shared__ double lds_t[NPW][11][8];                   // transposes: rows 0-6 the column lanes, 7-10 the spare lanes
  __shared__ double lds_d[NPW][3][15];                  // (row 2 stays zero)
  __shared__ double lsc[NPW][NSCAL];
  __shared__ double lds_c[NPW][C_N];
  __shared__ double lds_der[NPW][DER_N];                // (the evaluations copy it into registers of their own)
  constexpr int CHN = WIDE ? HCW : HCH;                 // nodes per chunk
  const int lane = threadIdx.x, grp = lane >> 4, role = lane & 15;
  const int nl = WIDE ? lane : role;                    // this lane's node within a chunk ...
  const bool nlane = nl < CHN;                          // ... if it has one
  const int cbase = WIDE ? 0 : grp * HCH;               // first column of this lane's NLP in the LDS stage
  const int col = nlane ? cbase + nl : 0;               // this lane's column in the node-parallel phases
  const int gi = WIDE ? 0 : grp;
  const long p = WIDE ? (long)blockIdx.x : (long)blockIdx.x * NPW + grp;
  const bool live = p < batch;
  const long pc = live ? p : batch - 1;                 // dead groups shadow the last NLP and never store
  const int K = g.K, Kp = g.Kp, nch = g.nch;
  double *w = ws + (size_t)pc * g.nlp_doubles();
  double *gsc = w + (size_t)NROWS * Kp;
  double *sc = lsc[gi];
  double *carry = lds_c[gi];
  const Der d = TERM == 2 ? derive_t(params[pc], 2) : derive(params[pc]);
  if (role == 0) der_store(lds_der[gi], d);
  const ldbl *dp = (const ldbl *)lds_der[gi];
  ldbl *lstage = (ldbl *)stage, *lcarry = (ldbl *)carry;
  const ldbl *lsc_ = (const ldbl *)sc;
  for (int r = role; r < NSCAL; r += 16) sc[r] = gsc[r];
  if (role < 15) lds_d[grp][2][role] = 0.0;
  wsync();
  if (!live && role == 0) sc[X_STATE] = ST_DONE;
  wsync();
  const double hT = (1.0 / K) * d.T;
  // lane roles of the factorisation sweep
  constexpr int RU = 7, RL = 8, RS = 11;                // the control's column; the first right-hand side; the first row of Lb Jb^-1 [Ja Ju]
  const bool col8 = role < 8, spare = role >= RS && role < RS + 4;

  PROF_DECL
  for (int round = 0; round < 64 * (max_iter + 2); round++) {
    PROF(9);
    // ============================ A: trial point, merit function and KKT error ======================================
    int state = (int)sc[X_STATE];
    if (__all(state == ST_DONE)) break;
    if (state == ST_TRIAL) {
      const bool first = sc[X_FIRST] != 0.0;
      const double alpha = first ? 0.0 : sc[X_ALPHA], adu = first ? 0.0 : sc[X_ADU], mu = sc[X_MU];
      const Scal s = lds_scal(sc, X_S), ds = lds_scal(sc, X_D);
      const Scal stt = trial_scal(d, s, ds, alpha, adu, mu, first);
      const int cur = (int)sc[X_CUR];
      const double *ic = w + (size_t)(cur * NIT) * Kp, *stp = w + (size_t)R_ST * Kp;
      double *in = w + (size_t)((1 - cur) * NIT) * Kp;
      Part P;
      if (sc[X_TEVAL] != 0.0) {             // evaluated by the adjoint phase of the previous round
        P.rd = sc[X_P + 0]; P.cinf = sc[X_P + 1]; P.pmin = sc[X_P + 2]; P.pmax = sc[X_P + 3]; P.l1 = sc[X_P + 4];
        P.zsum = sc[X_P + 5]; P.rth = sc[X_P + 6]; P.c1 = sc[X_P + 7]; P.sl = sc[X_P + 8]; P.mv = 0.0;
      } else {
        HsTrial t;
        t.alpha = alpha; t.adu = adu; t.mu = mu; t.dt = hT * stt.th; t.hT = hT; t.first = first;
        P.clear();
        for (int c = nch - 1; c >= 0; c--) {
          const int k = c * CHN + nl;
          P = hs_trial_chunk<TERM>(dp, lsc_, K, Kp, k, nlane && k < K, nl == CHN - 1, nl == 0, (const gdbl *)ic, (const gdbl *)stp, (gdbl *)in, t, live, lcarry, P);
        }
        P.template reduceW<0, WIDE>();
      }
      double rd = P.rd, cinf = P.cinf, pmin = P.pmin, pmax = P.pmax, l1 = P.l1, zsum = P.zsum;
      const double rth = 1.0 + P.rth, c1 = P.c1, sl = P.sl;
      // ---- decisions (all 16 lanes of the NLP alike; lane 0 writes) -------------------------------------------------
      double nu_pen = sc[X_NUP], iters = sc[X_ITERS], mu2 = mu;
      int nstate = ST_FACTOR;
      bool accepted = true;
      if (!first) {
        const double phi0 = sc[X_PHI0], Dm = sc[X_DM];
        const double phit = stt.th - mu * sl + nu_pen * c1;
        if (!(isfinite(phit) && phit <= phi0 + 1e-8 * alpha * Dm + 2.220446049250313e-15 * fabs(phi0))) {
          accepted = false;
          const int ls = (int)sc[X_LS] + 1;
          wsync();
          if (role == 0) {
            sc[X_LS] = ls; sc[X_TEVAL] = 0.0;
            if (ls >= 40) { sc[X_STATUS] = ASCENT_LINESEARCH_FAILED; sc[X_STATE] = ST_DONE; }
            else sc[X_ALPHA] = 0.5 * alpha;
          }
        } else {
          iters += 1.0;
        }
      }
      if (accepted) {
        ErrParts e;
        e.rd = fmax(rd, fabs(rth - stt.zlt + stt.zut));
        e.rd = fmax(e.rd, fmax(fabs(-stt.nu1 - stt.zs1), fabs(-stt.nu2 - stt.zs2)));
        e.cinf = cinf;
        const double pr[4] = {(stt.th - d.tlb) * stt.zlt, (d.tub - stt.th) * stt.zut, stt.s1 * stt.zs1, stt.s2 * stt.zs2};
        ASC_UNROLL
        for (int q = 0; q < 4; q++) { pmin = fmin(pmin, pr[q]); pmax = fmax(pmax, pr[q]); }
        e.pmin = pmin; e.pmax = pmax;
        l1 += fabs(stt.nu3) + fabs(stt.nu1) + fabs(stt.nu2);
        zsum += stt.zlt + stt.zut + stt.zs1 + stt.zs2;
        e.sd = fmax(100.0, (l1 + zsum) / (double)(13 * K + 7)) * 0.01;
        int status = -1;
        const bool probe = sc[X_PROBE] != 0.0;
        if (probe) { }
        else if (e.err(0.0) <= tol) { status = ASCENT_CONVERGED; nstate = ST_DONE; }
        else if ((int)iters >= max_iter) { status = ASCENT_MAX_ITER; nstate = ST_DONE; }
        else {
          while (mu2 > tol * 0.1 && e.err(mu2) <= 10.0 * mu2) {
            mu2 = fmax(tol * 0.1, fmin(0.2 * mu2, mu2 * sqrt(mu2)));
            nu_pen = 1.0;
          }
        }
#ifdef PERSIST_TRACE      // diagnostic build: the iteration history of one NLP
        if (role == 0 && p == PERSIST_TRACE && K > 100)
          printf("[hs] K=%d iter %2d mu %.1e E0 %.2e (dual %.1e primal %.1e compl %.1e..%.1e s_d %.2g) alpha %.3g adu %.3g ls %d dw %.1e nu_pen %.2g c1 %.2e\n", K, (int)iters, mu,
                 e.err(0.0), e.rd, e.cinf, e.pmin, e.pmax, e.sd, alpha, adu, (int)sc[X_LS], sc[X_DWL], nu_pen, c1);
#endif
        wsync();
        if (role == 0) {
          put_scal(sc, X_S, stt);
          sc[X_CUR] = 1 - cur; sc[X_FIRST] = 0.0; sc[X_ITERS] = iters; sc[X_LS] = 0.0; sc[X_C1] = c1; sc[X_SL] = sl; sc[X_RTH] = rth;
          sc[X_MU] = mu2; sc[X_NUP] = nu_pen; sc[X_DW] = probe ? sc[X_PDW] : 0.0; sc[X_STATE] = nstate; sc[X_TEVAL] = 0.0;
          if (status >= 0) sc[X_STATUS] = status;
        }
      }
    }
    wsync();
    PROF(0);
    // ============================ B: node blocks into LDS + backward factorisation ==================================
    state = (int)sc[X_STATE];
    if (__any(state == ST_FACTOR)) {
      const bool act = state == ST_FACTOR;
      const Scal s = lds_scal(sc, X_S);
      const double mu = sc[X_MU], dw = sc[X_DW];
      const double *it = w + (size_t)((int)sc[X_CUR] * NIT) * Kp;
      const double h = hT * s.th, bu = h * d.alpha, e8 = 0.125 * h;
      // what this lane gathers for component i of its vector at the start of a step: its column of the node's Hessian (the (x, y, angle,
      // mass) block; bound terms included), or the stationarity residual (lane 8), or the theta column (lane 9)
      int grow[7];
      double gsgn[7];
      {
        constexpr int hmap[8] = {0, 1, -1, -1, 2, -1, 3, -1};
        constexpr int hrow[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};
        ASC_UNROLL
        for (int i = 0; i < 7; i++) {
          int row = H_H; double sgn = 0.0;
          if (role < 7) {
            ASC_UNROLL
            for (int c = 0; c < 7; c++)
              if (c == role && hmap[i] >= 0 && hmap[c] >= 0) { row = H_H + hrow[hmap[i]][hmap[c]]; sgn = 1.0; }
          } else if (role == RL) { row = H_RZ + i; sgn = 1.0; }
          else if (role == RL + 1) { row = H_GT + i; sgn = 1.0; }
          grow[i] = row * LDH;
          gsgn[i] = sgn;
        }
      }
      // four words of sixteen step quantities each, one row per lane, broadcast within the NLP
      const int rowA = (role < 8 ? H_GA + role : H_GM + role - 8) * LDH;
      const int rowB = (role < 8 ? H_GB + role : role < 12 ? H_EB + role - 8 : H_ES + role - 12) * LDH;
      const int rowC = (role < 10 ? H_W + role : role < 13 ? H_R0 + role - 10 : H_R0) * LDH;
      const int rowD = (role < 7 ? H_C + role : role < 14 ? H_JT + role - 7 : H_C) * LDH;
      const int rowK = (role < 11 ? role : 11) * LDH;       // gains out: rows 0-6 kappa, 7 the pivot, 8-10 k0 (11: dummy; consumed rows of the column)
      const int drow = role == RL ? 0 : role == RL + 1 ? 1 : 2;
      // rows of Lb (what a spare lane starts a step with) and of La (what it adds after the pull-back)
      double lbC[7], laC[8];
      ASC_UNROLL
      for (int i = 0; i < 7; i++) {
        const int r = role - RS;
        const double half = (r == 0 && i == IX) || (r == 1 && i == IY) || (r == 2 && i == IA) || (r == 3 && i == IM) ? 0.5 : 0.0;
        const double eh = (r == 0 && i == IVX) || (r == 1 && i == IVY) || (r == 2 && i == IW) ? e8 : 0.0;
        lbC[i] = spare ? half - eh : 0.0;
        laC[i] = spare ? half + eh : 0.0;
      }
      laC[7] = 0.0;
      double a[7];
      ASC_UNROLL
      for (int i = 0; i < 7; i++) a[i] = 0.0;
      double U = 0.0, V = 0.0, Yc = 0.0, Yt = 0.0, k10 = 0.0, k11 = 0.0, k12 = 0.0, k20 = 0.0, k22 = 0.0, hthth = 0.0;
      int bad = 0;
      double zK[7];
      ASC_UNROLL
      for (int i = 0; i < 7; i++) zK[i] = it[(O_Z + i) * Kp + K - 1];
      const Terminal tm = TERM == 2 ? terminal_eval_any(d, zK) : terminal_eval(d, zK);
      const double is1 = rcp(s.s1), is2 = rcp(s.s2);
      const double sig1 = s.zs1 * is1 + dw, sig2 = s.zs2 * is2 + dw;
      const double rs1 = -mu * is1 - s.nu1, rs2 = -mu * is2 - s.nu2;
      {
        const double cg1 = tm.g1 - s.s1, cg2 = tm.g2 - s.s2;
        double Qt[28];
        ASC_UNROLL
        for (int i = 0; i < 28; i++) Qt[i] = 0.0;
        if constexpr (TERM == 2) terminal_hessian_any(Qt, tm, s.nu1, s.nu2, sig1, sig2);
        else terminal_hessian(Qt, tm, s.nu3, s.nu1, s.nu2, sig1, sig2);
        const double w1 = s.nu1 + sig1 * cg1 + rs1, w2 = s.nu2 + sig2 * cg2 + rs2;
        double r0[4] = {s.nu3 * tm.e3g[0] + w1 * tm.g1g[0], s.nu3 * tm.e3g[1] + w1 * tm.g1g[1],
                        s.nu3 * tm.e3g[2] + w2 * tm.g2g[0], s.nu3 * tm.e3g[3] + w2 * tm.g2g[1]};
        if constexpr (TERM == 2) terminal_grad_any(tm, w1, w2, r0);
        ASC_UNROLL
        for (int i = 0; i < 7; i++) {
          double v = 0.0;
          ASC_UNROLL
          for (int c = 0; c < 7; c++) v = role == c ? Qt[sid(i, c)] : v;
          if (i < 4) { v = role == RL ? r0[i] : v; v = role == RL + 2 ? tm.e3g[i] : v; }
          a[i] = v;
        }
      }
      const bool probe_rows = sc[X_PROBE] == 2.0;
      for (int c = nch - 1; c >= 0 && !probe_rows; c--) {
        // ---- node-parallel: the blocks of the steps of this chunk (hs_eval_factor) -------------------------------------------
        {
          const int k = c * CHN + nl;
          hthth += hs_eval_factor(dp, (const gdbl *)it, lstage, lcarry, K, Kp, k, nlane && k < K && act, nl == CHN - 1, nl == 0, col, h, hT, mu, dw);
        }
        wsync();
        PROF(1);
        // ---- serial: the steps of the chunk, backwards; 16 lanes per NLP -----------------------------------------------------
        if (act) {
          for (int jj = CHN - 1; jj >= 0; jj--) {
            const int k = c * CHN + jj;
            if (k >= K) continue;
            const int cj = cbase + jj;
            double gq[7];
            ASC_UNROLL
            for (int i = 0; i < 7; i++) gq[i] = stage[grow[i] + cj];
            const double gA = stage[rowA + cj], gB = stage[rowB + cj], gC = stage[rowC + cj], gD = stage[rowD + cj];
            const double Ga[8] = {bcast16<0>(gA), bcast16<1>(gA), bcast16<2>(gA), bcast16<3>(gA), bcast16<4>(gA), bcast16<5>(gA), bcast16<6>(gA), bcast16<7>(gA)};
            const double Gm[8] = {bcast16<8>(gA), bcast16<9>(gA), bcast16<10>(gA), bcast16<11>(gA), bcast16<12>(gA), bcast16<13>(gA), bcast16<14>(gA), bcast16<15>(gA)};
            const double Gb[8] = {bcast16<0>(gB), bcast16<1>(gB), bcast16<2>(gB), bcast16<3>(gB), bcast16<4>(gB), bcast16<5>(gB), bcast16<6>(gB), bcast16<7>(gB)};
            const double eb[4] = {bcast16<8>(gB), bcast16<9>(gB), bcast16<10>(gB), bcast16<11>(gB)};
            const double es[4] = {bcast16<12>(gB), bcast16<13>(gB), bcast16<14>(gB), bcast16<15>(gB)};
            const double W[10] = {bcast16<0>(gC), bcast16<1>(gC), bcast16<2>(gC), bcast16<3>(gC), bcast16<4>(gC), bcast16<5>(gC), bcast16<6>(gC), bcast16<7>(gC),
                                  bcast16<8>(gC), bcast16<9>(gC)};
            const double R0 = bcast16<10>(gC), ru = bcast16<11>(gC), huth = bcast16<12>(gC);
            const double cc[7] = {bcast16<0>(gD), bcast16<1>(gD), bcast16<2>(gD), bcast16<3>(gD), bcast16<4>(gD), bcast16<5>(gD), bcast16<6>(gD)};
            const double jt[7] = {bcast16<7>(gD), bcast16<8>(gD), bcast16<9>(gD), bcast16<10>(gD), bcast16<11>(gD), bcast16<12>(gD), bcast16<13>(gD)};
            HsJ J;
            hs_coeffs(Ga, Gm, Gb, eb, es, h, bu, J);
            // (i) the node's own terms; a spare lane starts from its row of Lb
            ASC_UNROLL
            for (int i = 0; i < 7; i++) a[i] = spare ? lbC[i] : a[i] + gsgn[i] * gq[i];
            if (dw != 0.0) {
              ASC_UNROLL
              for (int i = 0; i < 7; i++) a[i] += role == i ? dw : 0.0;
            }
            // (ii) N = Jb^-T (.) Jb^-1: every vector once, the columns once more after the transpose
            double b[7];
            hs_solve_jbt(J, a, b);
            if (role < 7) {
              ASC_UNROLL
              for (int i = 0; i < 7; i++) lds_t[grp][role][i] = b[i];
            }
            wsync();
            if (role < 7) {
              double t[7];
              ASC_UNROLL
              for (int l2 = 0; l2 < 7; l2++) t[l2] = lds_t[grp][l2][role];
              hs_solve_jbt(J, t, b);
            }
            // (iii) shift by the defect of the right-hand side: n' = n - N rc, rc = c | J_theta | 0; the constants of the border
            {
              double d0 = 0.0, d1 = 0.0;
              ASC_UNROLL
              for (int i = 0; i < 7; i++) { d0 += b[i] * cc[i]; d1 += b[i] * jt[i]; }
              if (role < 15) { lds_d[grp][0][role] = d0; lds_d[grp][1][role] = d1; }
            }
            wsync();
            double xo[4], xr[4], xt[4];       // this lane's four-vector of the rank-4 term; -Lb Jb^-1 c and -Lb Jb^-1 J_theta
            {
              double uu = 0.0, vv = 0.0;
              ASC_UNROLL
              for (int i = 0; i < 7; i++) {
                const double prc = lds_d[grp][drow][i];
                const double pj = b[i] - prc, sj = b[i] + pj;
                uu += jt[i] * sj;
                vv += cc[i] * sj;
                b[i] = pj;
              }
              U += uu; V += vv;
              ASC_UNROLL
              for (int r = 0; r < 4; r++) {
                xr[r] = -lds_d[grp][0][RS + r]; xt[r] = -lds_d[grp][1][RS + r];
                xo[r] = role == RL ? xr[r] : role == RL + 1 ? xt[r] : 0.0;
              }
            }
            // (iv) pull back through -[Ja Ju]: every vector once (a spare lane then holds its row of Lam = [La 0] - Lb Jb^-1 [Ja Ju]),
            //      the columns once more after the transpose: lanes 0-7 then hold the columns of the 8x8 form in (dz_{k-1}, du_k)
            double m8[8];
            hs_apply_j8t(J, b, m8);
            ASC_UNROLL
            for (int i = 0; i < 8; i++) m8[i] += laC[i];
            if (role < 7 || spare) {
              ASC_UNROLL
              for (int i = 0; i < 8; i++) lds_t[grp][role < 7 ? role : role - 4][i] = m8[i];
            }
            wsync();
            double a8[8];
            {
              double t[7], o8[8];
              ASC_UNROLL
              for (int l2 = 0; l2 < 7; l2++) t[l2] = lds_t[grp][l2][col8 ? role : 0];
              hs_apply_j8t(J, t, o8);
              ASC_UNROLL
              for (int i = 0; i < 8; i++) a8[i] = col8 ? o8[i] : m8[i];
              if (col8) {
                ASC_UNROLL
                for (int r = 0; r < 4; r++) xo[r] = lds_t[grp][7 + r][role];
              }
            }
            wsync();
            // (v) the midpoint's curvature: + Lam' W Lam on the columns, + Lam' W xi0 on the right-hand sides, xi0'W xi0 into the border
            {
              const double t0 = W[0] * xo[0] + W[1] * xo[1] + W[2] * xo[2] + W[3] * xo[3];
              const double t1 = W[1] * xo[0] + W[4] * xo[1] + W[5] * xo[2] + W[6] * xo[3];
              const double t2 = W[2] * xo[0] + W[5] * xo[1] + W[7] * xo[2] + W[8] * xo[3];
              const double t3 = W[3] * xo[0] + W[6] * xo[1] + W[8] * xo[2] + W[9] * xo[3];
              ASC_UNROLL
              for (int i = 0; i < 8; i++) {
                const double l0 = bcast16<RS>(m8[i]), l1 = bcast16<RS + 1>(m8[i]), l2 = bcast16<RS + 2>(m8[i]), l3 = bcast16<RS + 3>(m8[i]);
                a8[i] += (l0 * t0 + l1 * t1) + (l2 * t2 + l3 * t3);
              }
              Yc += (t0 * xr[0] + t1 * xr[1]) + (t2 * xr[2] + t3 * xr[3]);
              Yt += (t0 * xt[0] + t1 * xt[1]) + (t2 * xt[2] + t3 * xt[3]);
            }
            // (vi) the control: its own curvature and gradients, the pivot, the gains
            a8[7] += role == RU ? R0 : role == RL ? ru : role == RL + 1 ? huth : 0.0;
            const double D = bcast16<RU>(a8[7]);
            const double coef = a8[7] * rcp(D);
            ASC_UNROLL
            for (int i = 0; i < 7; i++) a[i] = a8[i] - bcast16<RU>(a8[i]) * coef;
            stage[rowK + cj] = role == RU ? D : coef;
          }
        }
        wsync();
        PROF(2);
        // ---- flush the feedback gains of the chunk (node-parallel) ---------------------------------------------------------
        {
          const int k = c * CHN + nl;
          if (nlane && k < K && act) {
            if (live) {
              ASC_UNROLL
              for (int i = 0; i < 7; i++) w[(size_t)(R_KA + i) * Kp + k] = stage[i * LDH + col];
              ASC_UNROLL
              for (int i = 0; i < 3; i++) w[(size_t)(R_K0 + i) * Kp + k] = stage[(8 + i) * LDH + col];
            }
            const double k00 = stage[8 * LDH + col], k01 = stage[9 * LDH + col], k02 = stage[10 * LDH + col], D = stage[7 * LDH + col];
            const double Dk1 = D * k01, Dk2 = D * k02;
            k10 += Dk1 * k00; k11 += Dk1 * k01; k12 += Dk1 * k02; k20 += Dk2 * k00; k22 += Dk2 * k02;
            if (!(D > 0.0)) bad = 1;
          }
        }
        wsync();
        PROF(3);
      }
      // ---- border: the 2x2 system in (theta, nu3); inertia ---------------------------------------------------------------------
      const double V0 = bcast16<RL>(V), U0 = bcast16<RL>(U), V1 = bcast16<RL + 1>(V), U1 = bcast16<RL + 1>(U), V2 = bcast16<RL + 2>(V), U2 = bcast16<RL + 2>(U);
      const double Yc1 = bcast16<RL + 1>(Yc), Yt1 = bcast16<RL + 1>(Yt);
      (void)V0;
      k10 = gsumW<WIDE>(k10); k11 = gsumW<WIDE>(k11); k12 = gsumW<WIDE>(k12); k20 = gsumW<WIDE>(k20); k22 = gsumW<WIDE>(k22);
      hthth = gsumW<WIDE>(hthth);
      bad = (int)gmaxW<WIDE>((double)bad);
      if (probe_rows) {
        if (act && role == 0) sc[X_STATE] = ST_DONE;
      } else if (act) {
        // bilinear constants r_alpha . x_beta of the inner system (alpha, beta = residual 0, theta 1, nu3 2)
        const double v10 = -0.5 * (V1 + U0) + Yc1 - k10, v11 = -U1 + Yt1 - k11, v20 = -0.5 * V2 - k20, v21 = -0.5 * U2 - k12, v22 = -k22;
        int ok = !bad;
        double dth = 0.0, dnu3 = 0.0;
        if (ok) {
          const double itl = rcp(s.th - d.tlb), itu = rcp(d.tub - s.th);
          const double rthp = sc[X_RTH] + mu * (itu - itl);
          const double sth = s.zlt * itl + s.zut * itu + dw + hthth;
          const double a11 = sth + v11, a12 = v21, a22 = TERM == 2 ? -1.0 : v22;      // (TERM 2: no r.v = 0 row; a unit pivot closes nu3)
          const double b1 = -rthp - v10, b2 = -tm.e3 - v20;
          const double det = a11 * a22 - a12 * a12;
          if (det < 0.0) {
            const double idet = 1.0 / det;
            dth = (b1 * a22 - a12 * b2) * idet;
            dnu3 = (a11 * b2 - a12 * b1) * idet;
          } else {
            ok = 0;
          }
        }
        if (role == 0) {
          if (ok) {
            sc[X_DTH] = dth; sc[X_DNU3] = dnu3; sc[X_SIG1] = sig1; sc[X_SIG2] = sig2; sc[X_RS1] = rs1; sc[X_RS2] = rs2;
            sc[X_CG1] = tm.g1 - s.s1; sc[X_CG2] = tm.g2 - s.s2;       // (the same bits in every phase: see ascent_persist.hip)
            sc[X_DWL] = dw; sc[X_STATE] = ST_FACTORED;
          } else {
            const double ndw = next_delta_w(dw, sc[X_DWL]);
            if (sc[X_PROBE] != 0.0) sc[X_STATE] = ST_DONE;           // a probe reports the refusal
            else if (ndw > 1e10) { sc[X_STATUS] = ASCENT_REGULARISATION_FAILED; sc[X_STATE] = ST_DONE; }
            else sc[X_DW] = ndw;
          }
        }
      }
    }
    wsync();
    // ============================ F + A: forward and adjoint substitution ==========================================
    state = (int)sc[X_STATE];
    if (__any(state == ST_FACTORED)) {
      const bool act = state == ST_FACTORED;
      const Scal s = lds_scal(sc, X_S);
      const double mu = sc[X_MU], dw = sc[X_DWL], dth = sc[X_DTH], dnu3 = sc[X_DNU3];
      const double sig1 = sc[X_SIG1], sig2 = sc[X_SIG2], rs1 = sc[X_RS1], rs2 = sc[X_RS2];
      const double *it = w + (size_t)((int)sc[X_CUR] * NIT) * Kp;
      double *stp = w + (size_t)R_ST * Kp;
      const double h = hT * s.th, bu = h * d.alpha, e8 = 0.125 * h;
      const double tau = fmax(0.99, 1.0 - mu);
      // ---- forward -----------------------------------------------------------------------------------------------
      // dz_k = M_k dz_{k-1} + v_k with node-local M_k = Abar_k - bu (Jb^-1 e_w) kappa_k', Abar = -Jb^-1 Ja, v_k = bu (Jb^-1 e_w) du0_k - Jb^-1 (c + J_theta dtheta):
      // one row of a 6x6 matrix-vector product per lane and step (lanes 0-5: x y xdot ydot angle angledot, lane 6: du), the mass component a
      // prefix sum over the nodes -- the serial step of p_solve
      {
        constexpr int NC = 6, FS = 7;
        const int fbase = (role < 7 ? FS * role : 0) * LDH;
        const int fout = (F_OUT + (role < 6 ? role : role == 6 ? 7 : 8)) * LDH;     // out rows 0-5 dz, 6 dz_m (from the scan), 7 du, 8 dummy
        double yown = 0.0, carry_m = 0.0;
        double rmax = 0.0, gsum = 0.0, adu = 1.0;
        double dzK[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < nch; c++) {
          const int kn = c * CHN + nl;
          const bool on = nlane && kn < K && act;
          carry_m = hs_eval_forward<WIDE>(dp, (const gdbl *)it, (const gdbl *)(w + (size_t)R_KA * Kp), lstage, K, Kp, kn, on, nl, col, h, hT, dth, dnu3, carry_m);
          wsync();
          PROF(4);
          if (act) {
            const int jn = min(CHN, K - c * CHN);
            for (int jj = 0; jj < jn; jj++) {
              const int cj = cbase + jj;
              const double *sj = stage + fbase + cj;
              const double m0 = sj[0], m1 = sj[LDH], m2 = sj[2 * LDH], m3 = sj[3 * LDH], m4 = sj[4 * LDH], m5 = sj[5 * LDH], vv = sj[NC * LDH];
              const double b0 = bcast16<0>(yown), b1 = bcast16<1>(yown), b2 = bcast16<2>(yown), b3 = bcast16<3>(yown),
                           b4 = bcast16<4>(yown), b5 = bcast16<5>(yown);
              const double e0 = (vv + m0 * b0) + m2 * b2, e1 = m1 * b1 + m3 * b3, e2 = m4 * b4 + m5 * b5;
              yown = (e0 + e1) + e2;
              stage[fout + cj] = yown;
            }
          }
          wsync();
          PROF(5);
          // ---- node-parallel: store the primal step, bound-multiplier steps, fraction to the boundary ----------------------
          if (on) {
            double dzn[8], zb[6];
            ASC_UNROLL
            for (int i = 0; i < 8; i++) dzn[i] = stage[(F_OUT + i) * LDH + col];
            const double a_ = it[(O_Z + IA) * Kp + kn], m_ = it[(O_Z + IM) * Kp + kn], u_ = it[O_U * Kp + kn];
            ASC_UNROLL
            for (int b = 0; b < 6; b++) zb[b] = it[(O_ZB + b) * Kp + kn];
            const double id[6] = {rcp(a_), rcp(d.aub - a_), rcp(m_), rcp(1.0 - m_), rcp(u_ + 1.0), rcp(1.0 - u_)};
            const double dza = dzn[IA], dzm = dzn[IM], du = dzn[7];
            if (kn == K - 1) cpy<7>(dzK, dzn);
            ASC_FTBR(rmax, id[0], dza); ASC_FTBR(rmax, id[1], -dza);
            ASC_FTBR(rmax, id[2], dzm); ASC_FTBR(rmax, id[3], -dzm);
            ASC_FTBR(rmax, id[4], du); ASC_FTBR(rmax, id[5], -du);
            gsum += dza * (id[1] - id[0]) + dzm * (id[3] - id[2]) + du * (id[5] - id[4]);
            const double dx3[3] = {dza, dzm, du};
            double dzb[6];
            ASC_UNROLL
            for (int b = 0; b < 3; b++) {
              const double zl = zb[2 * b], zu = zb[2 * b + 1];
              dzb[2 * b] = id[2 * b] * (mu - zl * dx3[b]) - zl;
              dzb[2 * b + 1] = id[2 * b + 1] * (mu + zu * dx3[b]) - zu;
              ASC_FTB(adu, zl, dzb[2 * b]);
              ASC_FTB(adu, zu, dzb[2 * b + 1]);
            }
            if (live) {
              ASC_UNROLL
              for (int i = 0; i < 7; i++) stp[(O_Z + i) * Kp + kn] = dzn[i];
              stp[O_U * Kp + kn] = du;
              ASC_UNROLL
              for (int b = 0; b < 6; b++) stp[(O_ZB + b) * Kp + kn] = dzb[b];
            }
          }
          wsync();
          PROF(6);
        }
        rmax = gmaxW<WIDE>(rmax); gsum = gsumW<WIDE>(gsum); adu = gminW<WIDE>(adu);
        ASC_UNROLL
        for (int i = 0; i < 7; i++) dzK[i] = gsumW<WIDE>(dzK[i]);          // only the lane of the last node holds non-zeros
        // ---- the scalars of the step and the step lengths ------------------------------------------------------------------------
        double zK[7];
        ASC_UNROLL
        for (int i = 0; i < 7; i++) zK[i] = it[(O_Z + i) * Kp + K - 1];
        const Terminal tmK = TERM == 2 ? terminal_eval_any(d, zK) : terminal_eval(d, zK);
        Scal ds;
        ds.th = dth; ds.nu3 = dnu3;
        ds.s1 = sc[X_CG1] + tmK.g1g[0] * dzK[IX] + tmK.g1g[1] * dzK[IY];
        ds.s2 = sc[X_CG2] + tmK.g2g[0] * dzK[IVX] + tmK.g2g[1] * dzK[IVY];
        if constexpr (TERM == 2) {
          ds.s1 += tmK.g1v[0] * dzK[IVX] + tmK.g1v[1] * dzK[IVY];
          ds.s2 += tmK.g2p[0] * dzK[IX] + tmK.g2p[1] * dzK[IY];
        }
        ds.nu1 = sig1 * ds.s1 + rs1;
        ds.nu2 = sig2 * ds.s2 + rs2;
        ds.zs1 = mu / s.s1 - s.zs1 - s.zs1 / s.s1 * ds.s1;
        ds.zs2 = mu / s.s2 - s.zs2 - s.zs2 / s.s2 * ds.s2;
        const double dl_ = s.th - d.tlb, dU = d.tub - s.th;
        ds.zlt = mu / dl_ - s.zlt - s.zlt / dl_ * ds.th;
        ds.zut = mu / dU - s.zut + s.zut / dU * ds.th;
        double apr = 1.0;
        if (rmax * apr > tau) apr = tau / rmax;
        ASC_FTB(apr, dl_, ds.th); ASC_FTB(apr, dU, -ds.th);
        ASC_FTB(apr, s.s1, ds.s1); ASC_FTB(apr, s.s2, ds.s2);
        ASC_FTB(adu, s.zlt, ds.zlt); ASC_FTB(adu, s.zut, ds.zut);
        ASC_FTB(adu, s.zs1, ds.zs1); ASC_FTB(adu, s.zs2, ds.zs2);
        HsTrial tc;
        tc.alpha = apr; tc.adu = adu; tc.mu = mu; tc.hT = hT; tc.first = false;
        tc.dt = hT * trial_scal(d, s, ds, apr, adu, mu, false).th;
        wsync();
        if (role == 0 && act) put_scal(sc, X_D, ds);      // (the trial point's scalars are formed from the record: iterate X_S, step X_D)
        wsync();
        double *in = w + (size_t)((1 - (int)sc[X_CUR]) * NIT) * Kp;
        Part P;
        P.clear();
        // ---- adjoint (backwards over the chunks) ---------------------------------------------------------------------------
        // With psi_k = -Ja_{k+1}' dlam_{k+1} the stationarity row of node k reads Jb_k' dlam_k = psi_k - rhs_k (rhs_k: everything the forward
        // sweep has fixed), and psi_{k-1} = Abar_k' (psi_k - rhs_k): affine in psi with node-local coefficients.  Columns angledot and mass of
        // Abar' are unit vectors and column angle is e_a + h e_w: the serial step of p_solve (five coefficients, the lane's own value, w).
        const int abase = (role < 7 ? 6 * role : 0) * LDH;
        const int aout = (role < 7 ? role : 8) * LDH;         // psi_{k-1} over the consumed coefficient rows 0-6 of the column (8: dummy)
        const double aself = (role == IW || role == IM) ? 1.0 : 0.0;
        double lown = 0.0;
        double cl = 0.0, ccl = 0.0;
        for (int c = nch - 1; c >= 0; c--) {
          const int kn = c * CHN + nl;
          const bool on = nlane && kn < K && act;
          ccl += hs_eval_adjoint<TERM>(dp, lsc_, (const gdbl *)it, (const gdbl *)stp, lstage, lcarry, K, Kp, kn, on, nl == CHN - 1, nl == 0, col, h, hT, mu, dw, dth, dnu3);
          if (role < 7) carry[C_PSI + role] = lown;            // psi of the chunk's last node: what the sweep of the chunk above has left
          wsync();
          PROF(7);
          if (act) {
            const int jj0 = min(CHN, K - c * CHN) - 1;
            for (int jj = jj0; jj >= 0; jj--) {
              const int cj = cbase + jj;
              const double *sj = stage + abase + cj;
              const double n0 = sj[0], n1 = sj[LDH], n2 = sj[2 * LDH], n3 = sj[3 * LDH], n4 = sj[4 * LDH], wv = sj[5 * LDH];
              const double b0 = bcast16<0>(lown), b1 = bcast16<1>(lown), b2 = bcast16<2>(lown), b3 = bcast16<3>(lown), b4 = bcast16<4>(lown);
              const double e0 = (wv + aself * lown) + n0 * b0, e1 = n1 * b1 + n2 * b2, e2 = n3 * b3 + n4 * b4;
              lown = (e0 + e1) + e2;
              stage[aout + cj] = lown;
            }
          }
          wsync();
          PROF(8);
          ccl += hs_post_adjoint(lstage, lcarry, (gdbl *)stp, K, Kp, kn, on, nl == CHN - 1, col, h, live);
          // ---- the step of the chunk is complete: its trial point at the first step length ---------------------------------------------
          P = hs_trial_chunk<TERM>(dp, lsc_, K, Kp, kn, on, nl == CHN - 1, nl == 0, (const gdbl *)it, (const gdbl *)stp, (gdbl *)in, tc, live, lcarry, P);
          PROF(0);
        }
        P.template reduceW<0, WIDE>();
        // ---- scalars of the step, merit bookkeeping -------------------------------------------------------------------------
        ccl = gsumW<WIDE>(ccl);
        if (act) {
          cl += ccl;
          const Terminal &tm = tmK;
          double gd = mu * gsum;
          gd += ds.th * (1.0 - mu / dl_ + mu / dU) - mu * ds.s1 / s.s1 - mu * ds.s2 / s.s2;
          cl += tm.e3 * (s.nu3 + ds.nu3) + sc[X_CG1] * (s.nu1 + ds.nu1) + sc[X_CG2] * (s.nu2 + ds.nu2);
          const double c1 = sc[X_C1], slog = sc[X_SL];
          double nu_pen = sc[X_NUP];
          const double curv = -gd + cl;
          if (c1 > 0.0) {
            const double need = (gd + 0.5 * fmax(curv, 0.0)) / (0.9 * c1);
            if (nu_pen < need) nu_pen = need + 1.0;
          }
          wsync();
          if (role == 0) {
            sc[X_NUP] = nu_pen;
            sc[X_DM] = gd - nu_pen * c1;
            sc[X_PHI0] = s.th - mu * slog + nu_pen * c1;
            sc[X_ALPHA] = apr; sc[X_ADU] = adu; sc[X_LS] = 0.0;
            sc[X_P + 0] = P.rd; sc[X_P + 1] = P.cinf; sc[X_P + 2] = P.pmin; sc[X_P + 3] = P.pmax; sc[X_P + 4] = P.l1;
            sc[X_P + 5] = P.zsum; sc[X_P + 6] = P.rth; sc[X_P + 7] = P.c1; sc[X_P + 8] = P.sl; sc[X_P + 9] = 0.0;
            sc[X_TEVAL] = 1.0;
            sc[X_STATE] = ST_TRIAL;
          }
        }
      }
    }
    wsync();
    if (sc[X_PROBE] != 0.0) break;
  }
  wsync();
  PROF_END;
#ifdef PERSIST_PROFILE
  if (blockIdx.x == 0 && threadIdx.x == 0) { printf("[hs trial] loads %llu points %llu defect %llu dual %llu rest %llu tail %llu\n", g_hsprof[0], g_hsprof[1], g_hsprof[2], g_hsprof[3], g_hsprof[4], g_hsprof[5]); for (int i = 0; i < 8; i++) g_hsprof[i] = 0; }
#endif
  if (live)
    for (int r = role; r < NSCAL; r += 16) gsc[r] = sc[r];
}

}  // namespace

namespace ascent {

// (declared in ascent_persist.hpp; g comes from persist's geo_of with the chunk sizes above)
void hs_launch_solve(long batch, hipStream_t stream, const ascent_params *dp, int K, int Kp, int nch, int term, int wide, double *w, int max_iter, double tol) {
  PGeo g;
  g.K = K; g.Kp = Kp; g.nch = nch; g.form = 0; g.mp = 0; g.term = term == 2 ? 2 : 0; g.wide = wide ? 1 : 0;
  if (wide) {
    const dim3 gw((unsigned)batch), bw(WAVE);
    if (g.term == 2) hipLaunchKernelGGL((h_solve<2, 1>), gw, bw, 0, stream, dp, batch, g, w, max_iter, tol);
    else hipLaunchKernelGGL((h_solve<0, 1>), gw, bw, 0, stream, dp, batch, g, w, max_iter, tol);
    return;
  }
  const dim3 grid((unsigned)((batch + NPW - 1) / NPW)), block(WAVE);
  if (g.term == 2) hipLaunchKernelGGL((h_solve<2, 0>), grid, block, 0, stream, dp, batch, g, w, max_iter, tol);
  else hipLaunchKernelGGL((h_solve<0, 0>), grid, block, 0, stream, dp, batch, g, w, max_iter, tol);
}
int hs_chunk_nodes(int wide) { return wide ? HCW : HCH; }

}  // namespace ascent
