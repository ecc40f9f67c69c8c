// Split pipeline of the batched ascent NLP solver for small and medium batches (gfx950).
//
// The fused kernel (ascent_solver.hip) gives every NLP one lane and sweeps the 199 collocation steps
// serially for everything; at batch 4096 that occupies 64 of the chip's 1024 SIMDs.  Only the three
// recurrences of a Newton step are inherently serial in time: the Riccati factorisation, the forward
// substitution and the adjoint substitution.  Everything else -- the trial point x + alpha*dx, the
// dynamics defects and their Jacobian/Hessian blocks at every collocation node (Launch_Optimiser.py:
// 114-136), the merit function and the KKT error -- has no recurrence.  This file therefore runs one
// interior-point iteration as
//     q_trial_eval   grid (tiles x step-chunks): trial point, node evaluation, merit + error partials
//     q_decide_factor one wavefront per tile: line-search / convergence / barrier decisions per lane,
//                     then the backward factorisation of the bordered block-tridiagonal KKT system
//     q_forward      one wavefront per tile: primal step, adjoint right-hand side, bound-multiplier
//                     steps, fraction-to-boundary
//     q_adjoint      one wavefront per tile: multiplier step, step-size/merit bookkeeping
// driven by a host loop that reads three counters per iteration.  Lanes carry a small state machine
// (trial / factor / factored / done), so a rejected line-search trial or a wrong-inertia
// factorisation re-runs only the lanes concerned.  The arithmetic per lane is the same as in the
// fused kernel; lanes of one tile still never talk to each other.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "ascent.h"
#include "ascent_device.hpp"
#include "ascent_tile.hpp"
#include "ascent_pipeline.hpp"

using namespace ascent;

namespace {

// ---- rows of one step record -----------------------------------------------------------------
constexpr int Q_IT = 0;                 // two iterate buffers of 21 rows: z[7] u lambda[7] zb[6]
constexpr int Q_ST = 42;                // step: dz[7] du dlambda[7] dzb[6]
constexpr int O_Z = 0, O_U = 7, O_L = 8, O_ZB = 15;
constexpr int Q_G = 63, Q_E = 71, Q_H = 75, Q_F = 85, Q_C = 92, Q_RZ = 99, Q_GT = 106, Q_SC = 113;
constexpr int Q_KA = 120, Q_K0 = 127, Q_R = 130;
constexpr int Q_STAGE = 137;
// Q_SC rows: R0 ru0 bza bzm bu (2 spare)
constexpr int CHUNK = 8;                // collocation steps per wavefront of q_trial_eval
constexpr int NPART = 15;               // rd cinf pmin pmax l1 zsum rth c1 sl + (c1, sl) at alpha/2, /4, /8
constexpr int NLAD = 3;                 // extra step sizes whose merit value q_trial_eval computes on the side

// ---- per-lane scalar rows (after the step records of the tile) -----------------------------------
enum {
  X_STATE, X_ITERS, X_STATUS, X_FBUF, X_LS, X_FIRST,
  X_S,                       // 10: th zlt zut s1 s2 zs1 zs2 nu3 nu1 nu2 (accepted iterate)
  X_T = X_S + 10,            // 10: trial scalars
  X_D = X_T + 10,            // 10: step scalars
  X_MU = X_D + 10, X_NUP, X_DW, X_DWL, X_ALPHA, X_ADU, X_PHI0, X_DM, X_C1, X_SL,
  X_DTH, X_DNU3, X_SIG1, X_SIG2, X_RS1, X_RS2, X_CUR, X_RTH, X_ROUNDS,
  X_LAD,                     // 6: merit pieces (c1, sl) of the trial ladder alpha/2, alpha/4, alpha/8
  NSCAL = X_LAD + 6
};
enum { ST_TRIAL = 0, ST_FACTOR = 1, ST_FACTORED = 2, ST_DONE = 3 };

using QTile = TileT<Q_STAGE>;

struct Geo {      // geometry of the workspace (+ formulation: 0 current script, 1 v1 script)
  int K, nch, form;
  int probe;      // parity probe (pipeline_probe): one Newton step at the caller's iterate with the caller's mu and delta_w --
                  // the iterate is taken as it is, no convergence test, no barrier update, no refactorisation
  __host__ __device__ size_t tile_doubles() const {
    return ((size_t)K * Q_STAGE + NSCAL + (size_t)nch * NPART) * WAVE;
  }
};
ASC_DEV gdbl *scal_base(const QTile &t_, const Geo &g) { return t_.base + (size_t)g.K * Q_STAGE * WAVE; }
ASC_DEV gdbl *part_base(const QTile &t_, const Geo &g, int chunk) {
  return t_.base + ((size_t)g.K * Q_STAGE + NSCAL + (size_t)chunk * NPART) * WAVE;
}
#define SC(r) ROW(sc, r)

ASC_DEV Scal load_scal(const QTile &t_, const gdbl *sc, int r0) {
  Scal s;
  s.th = SC(r0 + S_TH); s.zlt = SC(r0 + S_ZLT); s.zut = SC(r0 + S_ZUT); s.s1 = SC(r0 + S_S1);
  s.s2 = SC(r0 + S_S2); s.zs1 = SC(r0 + S_ZS1); s.zs2 = SC(r0 + S_ZS2); s.nu3 = SC(r0 + S_NU3);
  s.nu1 = SC(r0 + S_NU1); s.nu2 = SC(r0 + S_NU2);
  return s;
}
ASC_DEV void store_scal(const QTile &t_, gdbl *sc, int r0, const Scal &s) {
  SC(r0 + S_TH) = s.th; SC(r0 + S_ZLT) = s.zlt; SC(r0 + S_ZUT) = s.zut; SC(r0 + S_S1) = s.s1;
  SC(r0 + S_S2) = s.s2; SC(r0 + S_ZS1) = s.zs1; SC(r0 + S_ZS2) = s.zs2; SC(r0 + S_NU3) = s.nu3;
  SC(r0 + S_NU1) = s.nu1; SC(r0 + S_NU2) = s.nu2;
}

// trial scalars  s + alpha*ds  (primal / equality multipliers) and clipped bound multipliers
ASC_DEV Scal trial_scal(const Der &d, const Scal &s, const Scal &ds, double alpha, double adu, double mu,
                        bool first) {
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

// ==============================================================================================
// q_init: initial point into iterate buffer 0, zero step, per-lane state.  One wavefront = 64 NLPs x CHUNK steps
// (grid tiles x chunks); the wavefront of the last chunk also sets the scalars.
// ==============================================================================================
__global__ __launch_bounds__(WAVE) void q_init(const ascent_params *params, long batch, Geo g, double *ws,
                                               const double *guess, int warm, double mu_init,
                                               const double *probe_mu, const double *probe_dw) {
  const long p = (long)blockIdx.x * WAVE + threadIdx.x;
  if (p >= batch) return;
  const QTile t_((gdbl *)ws + (size_t)blockIdx.x * g.tile_doubles());
  const Der d = derive(params[p]);
  const int K = g.K, chunk = blockIdx.y;
  const int k_lo = chunk * CHUNK, k_hi = min(K, k_lo + CHUNK) - 1;
  // a guess whose theta is not positive means "no guess for this problem" (nested iteration: the coarse solve failed)
  const int asked_warm = warm;
  if (warm && !(guess[(21L * K + S_TH) * batch + p] > 0.0)) warm = 0;
  // cold start: straight-line states toward a tangential insertion point, u = 0
  const double tf0 = 0.9, dr = 0.166, aend = 0.5, vp = sqrt(d.vp2), dt0 = (1.0 / K) * d.T * tf0;
  const double sdr = sin(dr), cdr = cos(dr);
  const double xf = -d.rhof * sdr, yf = d.rhof * cdr - d.rho0;
  double zK[7];
  for (int k = k_lo; k <= k_hi; k++) {
    gdbl *sp = t_.st(k);
    double z[7], l[7], zb[6], u;
    if (warm) {
      ASC_UNROLL
      for (int i = 0; i < 7; i++) {
        z[i] = guess[(7L * k + i) * batch + p];
        l[i] = guess[(8L * K + 7L * k + i) * batch + p];
      }
      u = guess[(7L * K + k) * batch + p];
      ASC_UNROLL
      for (int b = 0; b < 6; b++) zb[b] = guess[(15L * K + 6L * k + b) * batch + p];
    } else {
      const double fr = (double)(k + 1) / K;
      z[IX] = fr * xf; z[IY] = fr * yf; z[IVX] = -fr * vp * cdr; z[IVY] = -fr * vp * sdr; z[IA] = fr * aend;
      z[IW] = aend / (K * dt0); z[IM] = d.mrate * dt0 * (k + 1);
      u = 0.0;
      if (g.form == 1) {       // v1: the angle is the control: angle = (ub/2)(u+1), no angular rate
        z[IW] = 0.0;
        u = z[IA] / (0.5 * d.aub) - 1.0;
      }
    }
    if (!g.probe) {
      z[IA] = push_in(z[IA], 0.0, d.aub);
      z[IM] = push_in(z[IM], 0.0, 1.0);
      u = push_in(u, -1.0, 1.0);
    }
    ASC_UNROLL
    for (int b = 0; b < 6; b++) zb[b] = warm == 2 ? (g.probe ? zb[b] : fmax(zb[b], 1e-12)) : 1.0;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) l[i] = warm == 2 ? l[i] : 0.0;
    stn<7>(t_, sp, Q_IT + O_Z, z);
    ROW(sp, Q_IT + O_U) = u;
    stn<7>(t_, sp, Q_IT + O_L, l);
    stn<6>(t_, sp, Q_IT + O_ZB, zb);
    ASC_UNROLL
    for (int r = 0; r < 21; r++) ROW(sp, Q_ST + r) = 0.0;
    if (k == K - 1) cpy<7>(zK, z);
  }
  if (k_hi != K - 1) return;
  // ---- scalars (the wavefront that holds the last step) ----------------------------------------------
  gdbl *sc = scal_base(t_, g);
  Scal s;
  if (warm) {
    const double *gs = guess + (21L * K) * batch + p;
    s.th = gs[S_TH * batch]; s.zlt = gs[S_ZLT * batch]; s.zut = gs[S_ZUT * batch]; s.s1 = gs[S_S1 * batch];
    s.s2 = gs[S_S2 * batch]; s.zs1 = gs[S_ZS1 * batch]; s.zs2 = gs[S_ZS2 * batch]; s.nu3 = gs[S_NU3 * batch];
    s.nu1 = gs[S_NU1 * batch]; s.nu2 = gs[S_NU2 * batch];
  } else {
    s.th = tf0;
  }
  if (!g.probe) s.th = push_in(s.th, d.tlb, d.tub);
  const Terminal tm = terminal_eval(d, zK);
  if (g.probe) {
    // the caller's iterate as it is
  } else if (warm != 2) {
    s.s1 = fmax(tm.g1, 1e-2); s.s2 = fmax(tm.g2, 1e-2);
    s.zlt = s.zut = s.zs1 = s.zs2 = 1.0;
    s.nu3 = s.nu1 = s.nu2 = 0.0;
  } else {
    s.s1 = fmax(s.s1, 1e-10); s.s2 = fmax(s.s2, 1e-10);
    s.zlt = fmax(s.zlt, 1e-12); s.zut = fmax(s.zut, 1e-12);
    s.zs1 = fmax(s.zs1, 1e-12); s.zs2 = fmax(s.zs2, 1e-12);
  }
  for (int r = 0; r < NSCAL; r++) SC(r) = 0.0;
  store_scal(t_, sc, X_S, s);
  SC(X_STATE) = ST_TRIAL; SC(X_FIRST) = 1.0; SC(X_STATUS) = ASCENT_MAX_ITER;
  SC(X_MU) = (asked_warm && !warm) ? 0.1 : mu_init; SC(X_NUP) = 1.0;
  if (g.probe) { SC(X_MU) = probe_mu[p]; SC(X_DW) = probe_dw[p]; }
}

// per-lane iterate buffer: lanes advance asynchronously, so which of the two iterate buffers holds a
// lane's current iterate is per-lane state (X_CUR); its rows are reached through a per-lane offset
#define ROWO(p, r, off) (p)[(r) * WAVE + (off)]
template <int N>
ASC_DEV void ldo(const gdbl *p, int r0, unsigned off, double *v) {
  ASC_UNROLL
  for (int i = 0; i < N; i++) v[i] = ROWO(p, r0 + i, off);
}
template <int N>
ASC_DEV void sto(gdbl *p, int r0, unsigned off, const double *v) {
  ASC_UNROLL
  for (int i = 0; i < N; i++) ROWO(p, r0 + i, off) = v[i];
}
ASC_DEV unsigned buf_off(const QTile &t_, int buf) { return t_.lane + (unsigned)buf * (21 * WAVE); }

// ==============================================================================================
// q_trial_eval: one wavefront = 64 NLPs x CHUNK consecutive steps
// ==============================================================================================
// SCHEME 0: backward Euler (the reference's NODES=2).  SCHEME 1: trapezoid with the control held over the
// step, z_k - z_{k-1} - dt/2 [f(z_k,u_k) + f(z_{k-1},u_k)]: node k then enters steps k and k+1 with weight
// 1/2 each, so every occurrence of lambda_k in the node's dual residual / Hessian weights becomes
// lambda_k + lambda_{k+1} with step constant cs = dt/2, and the stored step function is the mean of the two f's.
template <int SCHEME, int FORM>
__global__ __launch_bounds__(WAVE) void q_trial_eval(const ascent_params *params, long batch, Geo g,
                                                     double *ws, int *counters) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 3) counters[threadIdx.x] = 0;   // the round's counters
  const long p = (long)blockIdx.x * WAVE + threadIdx.x;
  if (p >= batch) return;
  const QTile t_((gdbl *)ws + (size_t)blockIdx.x * g.tile_doubles());
  gdbl *sc = scal_base(t_, g);
  if ((int)SC(X_STATE) != ST_TRIAL) return;
  const Der d = derive(params[p]);
  const int K = g.K, chunk = blockIdx.y;
  const int k_lo = chunk * CHUNK, k_hi = min(K, k_lo + CHUNK) - 1;
  const bool first = SC(X_FIRST) != 0.0;
  const double alpha = SC(X_ALPHA), adu = SC(X_ADU), mu = SC(X_MU);
  const Scal s = load_scal(t_, sc, X_S), ds = load_scal(t_, sc, X_D);
  const Scal st = trial_scal(d, s, ds, alpha, adu, mu, first);
  const double hT = (1.0 / K) * d.T, dt = hT * st.th, be = dt * d.alpha;
  const double cs = SCHEME == 1 ? 0.5 * dt : dt, hTc = SCHEME == 1 ? 0.5 * hT : hT;
  const double ha = 0.5 * d.aub;            // FORM 1: angle = ha * (u + 1)
  const int cur = (int)SC(X_CUR);
  const unsigned oc = buf_off(t_, cur), on = buf_off(t_, 1 - cur);   // current / trial iterate rows
  const double mlo = mu * 1e-10, mhi = mu * 1e10;

  // Backtracking ladder: besides the full evaluation at alpha, the l1-merit pieces at alpha/2, alpha/4
  // and alpha/8 are computed on the side (this kernel is HBM-bound, the arithmetic is free), so that a
  // rejected trial can jump straight to the first acceptable halving instead of one halving per round.
  const bool ladder = !first;
  double al[NLAD], dtl[NLAD], lc1[NLAD], lsl[NLAD];
  ASC_UNROLL
  for (int j = 0; j < NLAD; j++) {
    al[j] = alpha * (0.5 / (double)(1 << j));
    dtl[j] = hT * (s.th + al[j] * ds.th);
    lc1[j] = 0.0; lsl[j] = 0.0;
  }
  double z[7], ln[7], zc[7], zd[7];   // zc, zd: current state and its step at the step being processed
  {   // trial state of the chunk's last step, trial multipliers of the step after it
    const gdbl *sp = t_.st(k_hi);
    ldo<7>(sp, Q_IT + O_Z, oc, zc);
    ldn<7>(t_, sp, Q_ST + O_Z, zd);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) z[i] = zc[i] + alpha * zd[i];
    if (k_hi + 1 < K) {
      const gdbl *sn = t_.st(k_hi + 1);
      double dl[7];
      ldo<7>(sn, Q_IT + O_L, oc, ln);
      ldn<7>(t_, sn, Q_ST + O_L, dl);
      ASC_UNROLL
      for (int i = 0; i < 7; i++) ln[i] += alpha * dl[i];
    } else {
      ASC_UNROLL
      for (int i = 0; i < 7; i++) ln[i] = 0.0;
    }
  }
  double rd = 0.0, cinf = 0.0, pmin = 1e300, pmax = -1e300, l1 = 0.0, zsum = 0.0, rth = 0.0, c1 = 0.0, sl = 0.0;
  for (int k = k_hi; k >= k_lo; k--) {
    gdbl *sp = t_.st(k);
    double zp[7], l[7], zb[6], tmp7[7], tmp6[6], zpc[7];
    ldo<7>(sp, Q_IT + O_L, oc, l);
    ldn<7>(t_, sp, Q_ST + O_L, tmp7);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) l[i] += alpha * tmp7[i];
    const double uc = ROWO(sp, Q_IT + O_U, oc), ud = ROW(sp, Q_ST + O_U);
    const double u = uc + alpha * ud;
    ldo<6>(sp, Q_IT + O_ZB, oc, zb);
    ldn<6>(t_, sp, Q_ST + O_ZB, tmp6);
    const double a = z[IA], m = z[IM];
    const double dist[6] = {a, d.aub - a, m, 1.0 - m, u + 1.0, 1.0 - u};
    double id[6];
    ASC_UNROLL
    for (int b = 0; b < 6; b++) {
      id[b] = rcp(dist[b]);
      if (!first) zb[b] = fmin(fmax(zb[b] + adu * tmp6[b], mlo * id[b]), mhi * id[b]);
    }
    if (k > 0) {
      const gdbl *spp = t_.st(k - 1);
      ldo<7>(spp, Q_IT + O_Z, oc, zpc);
      ldn<7>(t_, spp, Q_ST + O_Z, tmp7);
      ASC_UNROLL
      for (int i = 0; i < 7; i++) zp[i] = zpc[i] + alpha * tmp7[i];
    } else {
      ASC_UNROLL
      for (int i = 0; i < 7; i++) { zp[i] = 0.0; zpc[i] = 0.0; tmp7[i] = 0.0; }
    }
    if (ladder) {   // merit pieces at the shorter steps (values only: no derivatives, nothing stored)
      ASC_UNROLL
      for (int j = 0; j < NLAD; j++) {
        double zj[7], Fj[7], axj, ayj;
        ASC_UNROLL
        for (int i = 0; i < 7; i++) zj[i] = zc[i] + al[j] * zd[i];
        const double uj = uc + al[j] * ud;
        accel<0>(d, zj[IX], zj[IY], zj[IA], zj[IM], 0.0, 0.0, axj, ayj, nullptr, nullptr);
        rhs_f<FORM>(d, zj, uj, axj, ayj, Fj);
        double zpj[7];
        ASC_UNROLL
        for (int i = 0; i < 7; i++) zpj[i] = zpc[i] + al[j] * tmp7[i];
        if (SCHEME == 1) {
          double Fb[7];
          accel<0>(d, zpj[IX], zpj[IY], zpj[IA], zpj[IM], 0.0, 0.0, axj, ayj, nullptr, nullptr);
          rhs_f<FORM>(d, zpj, uj, axj, ayj, Fb);
          ASC_UNROLL
          for (int i = 0; i < 7; i++) Fj[i] = 0.5 * (Fj[i] + Fb[i]);
        }
        ASC_UNROLL
        for (int i = 0; i < 7; i++)
          lc1[j] += (FORM == 1 && i == IA) ? fabs(zj[IA] - ha * (uj + 1.0)) : fabs(zj[i] - zpj[i] - dtl[j] * Fj[i]);
        const double pa = zj[IA] * (d.aub - zj[IA]), pm = zj[IM] * (1.0 - zj[IM]), pu = (uj + 1.0) * (1.0 - uj);
        lsl[j] += (pa > 0.0 && pm > 0.0 && pu > 0.0) ? log(pa * pm * pu) : NAN;
        if (k == K - 1) {
          const Terminal tj = terminal_eval(d, zj);
          const double thj = s.th + al[j] * ds.th, s1j = s.s1 + al[j] * ds.s1, s2j = s.s2 + al[j] * ds.s2;
          lc1[j] += fabs(tj.e3) + fabs(tj.g1 - s1j) + fabs(tj.g2 - s2j);
          const double ps = ((thj - d.tlb) * (d.tub - thj)) * (s1j * s2j);
          lsl[j] += ps > 0.0 ? log(ps) : NAN;
        }
      }
    }
    // the trial iterate of this step
    sto<7>(sp, Q_IT + O_Z, on, z);
    ROWO(sp, Q_IT + O_U, on) = u;
    sto<7>(sp, Q_IT + O_L, on, l);
    sto<6>(sp, Q_IT + O_ZB, on, zb);
    // node evaluation (Launch_Optimiser.py:114-136) with Jacobian and Lagrangian-Hessian blocks
    double G[8], E[4], H[10], F[7], fl[7], lt[7], ax, ay;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) lt[i] = SCHEME == 1 ? l[i] + ln[i] : l[i];
    accel<2>(d, z[IX], z[IY], z[IA], z[IM], -cs * lt[IVX], -cs * lt[IVY], ax, ay, G, H);
    rhs_f<FORM>(d, z, u, ax, ay, F);
    if (SCHEME == 1) {       // second evaluation point of the step: f(z_{k-1}, u_k)
      double Fb[7], axp, ayp;
      accel<0>(d, zp[IX], zp[IY], zp[IA], zp[IM], 0.0, 0.0, axp, ayp, nullptr, nullptr);
      rhs_f<FORM>(d, zp, u, axp, ayp, Fb);
      ASC_UNROLL
      for (int i = 0; i < 7; i++) F[i] = 0.5 * (F[i] + Fb[i]);
    }
    implicit_block(G, cs, E);
    fzt_lambda<FORM>(G, lt, fl);
    double rz[7], gt[7], cc[7];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      rz[i] = (FORM == 1 && i == IA) ? l[i] - cs * fl[i] : l[i] - cs * fl[i] - ln[i];
      gt[i] = -hTc * fl[i];
      cc[i] = (FORM == 1 && i == IA) ? z[IA] - ha * (u + 1.0) : z[i] - zp[i] - dt * F[i];
      c1 += fabs(cc[i]);
      cinf = fmax(cinf, fabs(cc[i]));
      rth -= hT * F[i] * l[i];
      l1 += fabs(l[i]);
    }
    // Hessian block with the bound-barrier curvature of angle and mass folded in
    H[7] += zb[0] * id[0] + zb[1] * id[1];
    H[9] += zb[2] * id[2] + zb[3] * id[3];
    stn<8>(t_, sp, Q_G, G);
    stn<4>(t_, sp, Q_E, E);
    stn<10>(t_, sp, Q_H, H);
    stn<7>(t_, sp, Q_F, F);
    stn<7>(t_, sp, Q_C, cc);
    stn<7>(t_, sp, Q_RZ, rz);
    stn<7>(t_, sp, Q_GT, gt);
    const double ru0 = FORM == 1 ? -ha * l[IA] : -be * l[IW];      // d(lambda'c)/du
    const double scr[5] = {zb[4] * id[4] + zb[5] * id[5], ru0, id[1] - id[0], id[3] - id[2], id[5] - id[4]};
    stn<5>(t_, sp, Q_SC, scr);
    // KKT error pieces (dual residual with the actual bound multipliers) and merit pieces
    double r[7];
    cpy<7>(r, rz);
    r[IA] += zb[1] - zb[0];
    r[IM] += zb[3] - zb[2];
    if (k == K - 1) {
      const Terminal t = terminal_eval(d, z);
      r[IX] += st.nu3 * t.e3g[0] + st.nu1 * t.g1g[0];
      r[IY] += st.nu3 * t.e3g[1] + st.nu1 * t.g1g[1];
      r[IVX] += st.nu3 * t.e3g[2] + st.nu2 * t.g2g[0];
      r[IVY] += st.nu3 * t.e3g[3] + st.nu2 * t.g2g[1];
      const double e1 = fabs(t.e3), e2 = fabs(t.g1 - st.s1), e3 = fabs(t.g2 - st.s2);
      cinf = fmax(cinf, fmax(e1, fmax(e2, e3)));
      c1 += e1 + e2 + e3;
      const double ps = ((st.th - d.tlb) * (d.tub - st.th)) * (st.s1 * st.s2);
      sl += ps > 0.0 ? log(ps) : NAN;
      store_scal(t_, sc, X_T, st);
    }
    ASC_UNROLL
    for (int i = 0; i < 7; i++) rd = fmax(rd, fabs(r[i]));
    rd = fmax(rd, fabs(ru0 - zb[4] + zb[5]));
    ASC_UNROLL
    for (int b = 0; b < 6; b++) {
      const double pr = dist[b] * zb[b];
      pmin = fmin(pmin, pr);
      pmax = fmax(pmax, pr);
      zsum += zb[b];
    }
    const double pa = dist[0] * dist[1], pm = dist[2] * dist[3], pu = dist[4] * dist[5];
    sl += (pa > 0.0 && pm > 0.0 && pu > 0.0) ? log(pa * pm * pu) : NAN;
    cpy<7>(ln, l);
    cpy<7>(z, zp);
    cpy<7>(zc, zpc);
    cpy<7>(zd, tmp7);
  }
  gdbl *pp = part_base(t_, g, chunk);
  ROW(pp, 0) = rd; ROW(pp, 1) = cinf; ROW(pp, 2) = pmin; ROW(pp, 3) = pmax; ROW(pp, 4) = l1;
  ROW(pp, 5) = zsum; ROW(pp, 6) = rth; ROW(pp, 7) = c1; ROW(pp, 8) = sl;
  ASC_UNROLL
  for (int j = 0; j < NLAD; j++) { ROW(pp, 9 + 2 * j) = lc1[j]; ROW(pp, 10 + 2 * j) = lsl[j]; }
}

#ifdef DF_STAMPS   // diagnostic build only (scripts/df_stamps.py)
__device__ unsigned long long g_df_stamps[6];
#endif
// ==============================================================================================
// q_decide_factor: per-lane decisions, then the backward factorisation
// ==============================================================================================
struct InQM {   // matrix part of a step record (double-buffered one step ahead)
  double G[8], E[4], H[10], R0;
};
struct InQV {   // vector part (loaded at the top of the step, consumed after the congruence)
  double rz[7], gt[7], cc[7], F[7], ru0, bza, bzm, bu;
};
ASC_DEV void loadQM(const QTile &t_, int k, InQM &in) {
  const gdbl *sp = t_.st(k);
  ldn<8>(t_, sp, Q_G, in.G);
  ldn<4>(t_, sp, Q_E, in.E);
  ldn<10>(t_, sp, Q_H, in.H);
  in.R0 = ROW(sp, Q_SC);
}
ASC_DEV void loadQV(const QTile &t_, int k, InQV &in) {
  const gdbl *sp = t_.st(k);
  ldn<7>(t_, sp, Q_RZ, in.rz);
  ldn<7>(t_, sp, Q_GT, in.gt);
  ldn<7>(t_, sp, Q_C, in.cc);
  ldn<7>(t_, sp, Q_F, in.F);
  in.ru0 = ROW(sp, Q_SC + 1); in.bza = ROW(sp, Q_SC + 2); in.bzm = ROW(sp, Q_SC + 3); in.bu = ROW(sp, Q_SC + 4);
}

// The per-NLP decisions between two Newton steps: Armijo test of the trial point (walking the halving ladder on
// rejection), acceptance, convergence test, barrier update.  Shared by q_decide_factor (one lane per NLP, the lane
// writes) and q_factor_wide (16 lanes per NLP compute it redundantly, lane 0 writes): everything later code needs
// comes back in registers, because the other lanes of a group must not depend on seeing lane 0's stores.
struct Decided {
  int state, cur;          // state to continue with; which iterate buffer is current
  double mu, dw, rth;      // barrier parameter, primal regularisation, d/dtheta of the Lagrangian
  Scal s;                  // scalars of the current iterate
};
#define WSC(r, v) do { if (writer) SC(r) = (v); } while (0)
ASC_DEV Decided decide_lane(const QTile &t_, gdbl *sc, const Geo &g, const Der &d, int max_iter, double tol,
                            int *counters, bool writer) {
  Decided out;
  const int state = (int)SC(X_STATE);
  out.state = state; out.cur = (int)SC(X_CUR); out.mu = SC(X_MU); out.dw = SC(X_DW); out.rth = SC(X_RTH);
  if (state == ST_DONE) return out;
  WSC(X_ROUNDS, SC(X_ROUNDS) + 1.0);
  if (state == ST_FACTORED) {
    if (writer) atomicAdd(&counters[1], 1);
    return out;
  }
  if (state == ST_FACTOR) {      // refactorisation with a larger delta_w
    out.s = load_scal(t_, sc, X_S);
    return out;
  }
  const int K = g.K;
  double mu = out.mu;
  // ---- reduce the partials of the trial point -------------------------------------------------
  double rd = 0.0, cinf = 0.0, pmin = 1e300, pmax = -1e300, l1 = 0.0, zsum = 0.0, rth = 1.0, c1 = 0.0, sl = 0.0;
  for (int c = g.nch - 1; c >= 0; c--) {
    const gdbl *pp = part_base(t_, g, c);
    rd = fmax(rd, ROW(pp, 0)); cinf = fmax(cinf, ROW(pp, 1));
    pmin = fmin(pmin, ROW(pp, 2)); pmax = fmax(pmax, ROW(pp, 3));
    l1 += ROW(pp, 4); zsum += ROW(pp, 5); rth += ROW(pp, 6); c1 += ROW(pp, 7); sl += ROW(pp, 8);
  }
  const bool first = SC(X_FIRST) != 0.0;
  const Scal st = load_scal(t_, sc, X_T);
  double nu_pen = SC(X_NUP), iters = SC(X_ITERS);
  if (!first) {   // Armijo test on the l1 merit function
    const double alpha = SC(X_ALPHA), phi0 = SC(X_PHI0), Dm = SC(X_DM);
    const double phit = st.th - mu * sl + nu_pen * c1;
    if (!(isfinite(phit) && phit <= phi0 + 1e-8 * alpha * Dm + 2.220446049250313e-15 * fabs(phi0))) {
      // rejected: walk the halving sequence alpha/2, alpha/4, ... through the ladder values computed on the
      // side; the first acceptable one (or, failing that, alpha/16) is what gets evaluated in the next round
      int ls = (int)SC(X_LS) + 1;
      double an = 0.5 * alpha;
      const double th0 = SC(X_S + S_TH), dth0 = SC(X_D + S_TH);
      ASC_UNROLL
      for (int j = 0; j < NLAD; j++) {
        double c1j = 0.0, slj = 0.0;
        for (int c = g.nch - 1; c >= 0; c--) {
          const gdbl *pp = part_base(t_, g, c);
          c1j += ROW(pp, 9 + 2 * j); slj += ROW(pp, 10 + 2 * j);
        }
        const double thj = th0 + an * dth0;
        const double phij = thj - mu * slj + nu_pen * c1j;
        if (isfinite(phij) && phij <= phi0 + 1e-8 * an * Dm + 2.220446049250313e-15 * fabs(phi0)) break;
        if (ls >= 40) break;
        an *= 0.5; ls++;
      }
      WSC(X_LS, ls);
      if (ls >= 40) {
        WSC(X_STATUS, ASCENT_LINESEARCH_FAILED); WSC(X_STATE, ST_DONE);
        out.state = ST_DONE;
      } else {
        WSC(X_ALPHA, an);                // stays in ST_TRIAL: evaluated in full in the next round
        if (writer) atomicAdd(&counters[0], 1);
      }
      return out;
    }
    iters += 1.0;
    WSC(X_ITERS, iters);
  }
  // ---- accepted: the trial point is the iterate ---------------------------------------------------
  if (writer) store_scal(t_, sc, X_S, st);
  out.s = st; out.cur = 1 - out.cur; out.rth = rth; out.dw = 0.0;
  WSC(X_CUR, (double)out.cur);
  WSC(X_FIRST, 0.0); WSC(X_C1, c1); WSC(X_SL, sl); WSC(X_LS, 0.0); WSC(X_RTH, rth);
  ErrParts e;
  e.rd = fmax(rd, fabs(rth - st.zlt + st.zut));
  e.rd = fmax(e.rd, fmax(fabs(-st.nu1 - st.zs1), fabs(-st.nu2 - st.zs2)));
  e.cinf = cinf;
  const double pr[4] = {(st.th - d.tlb) * st.zlt, (d.tub - st.th) * st.zut, st.s1 * st.zs1, st.s2 * st.zs2};
  ASC_UNROLL
  for (int j = 0; j < 4; j++) { pmin = fmin(pmin, pr[j]); pmax = fmax(pmax, pr[j]); }
  e.pmin = pmin; e.pmax = pmax;
  l1 += fabs(st.nu3) + fabs(st.nu1) + fabs(st.nu2);
  zsum += st.zlt + st.zut + st.zs1 + st.zs2;
  e.sd = fmax(100.0, (l1 + zsum) / (double)(13 * K + 7)) * 0.01;
  if (g.probe) {         // parity probe: factorise here and now with the caller's mu and delta_w
    out.dw = SC(X_DW);
    out.state = ST_FACTOR;
    return out;
  }
  if (e.err(0.0) <= tol) {
    WSC(X_STATUS, ASCENT_CONVERGED); WSC(X_STATE, ST_DONE);
    out.state = ST_DONE;
    return out;
  }
  if ((int)iters >= max_iter) {
    WSC(X_STATUS, ASCENT_MAX_ITER); WSC(X_STATE, ST_DONE);
    out.state = ST_DONE;
    return out;
  }
  while (mu > tol * 0.1 && e.err(mu) <= 10.0 * mu) {
    mu = fmax(tol * 0.1, fmin(0.2 * mu, mu * sqrt(mu)));
    nu_pen = 1.0;
  }
  WSC(X_MU, mu); WSC(X_NUP, nu_pen); WSC(X_DW, 0.0);
  out.mu = mu;
  out.state = ST_FACTOR;
  return out;
}

template <int SCHEME, int FORM>
__global__ __launch_bounds__(WAVE) void q_decide_factor(const ascent_params *params, long batch, Geo g,
                                                        double *ws, int max_iter, double tol, int *counters) {
  const long p = (long)blockIdx.x * WAVE + threadIdx.x;
  if (p >= batch) return;
  const QTile t_((gdbl *)ws + (size_t)blockIdx.x * g.tile_doubles());
  gdbl *sc = scal_base(t_, g);
  const Der d = derive(params[p]);
  const int K = g.K;
  const Decided dc = decide_lane(t_, sc, g, d, max_iter, tol, counters, true);
  if (dc.state != ST_FACTOR) return;
  const double mu = dc.mu;
  // ---- backward factorisation at the current iterate, primal regularisation dw ---------------------
  const unsigned oc = buf_off(t_, dc.cur);
  const Scal s = dc.s;
  const double dw = dc.dw;
  const double hT = (1.0 / K) * d.T, dt = hT * s.th, be = dt * d.alpha, ith = rcp(s.th);
  const double cs = SCHEME == 1 ? 0.5 * dt : dt;
  constexpr int IB = FORM == 1 ? IA : IW;          // the defect row the control enters ...
  const double bu = FORM == 1 ? 0.5 * d.aub : be;  // ... and its coefficient
  double P[28], p0[7], p1[7], p2[7];
  ASC_UNROLL
  for (int i = 0; i < 28; i++) P[i] = 0.0;
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { p0[i] = p1[i] = p2[i] = 0.0; }
  double S10 = 0.0, S11 = 0.0, S12 = 0.0, S20 = 0.0, S22 = 0.0;
  double zK[7];
  ldo<7>(t_.st(K - 1), Q_IT + O_Z, oc, zK);
  const Terminal tm = terminal_eval(d, zK);
  const double is1 = rcp(s.s1), is2 = rcp(s.s2);
  const double sig1 = s.zs1 * is1 + dw, sig2 = s.zs2 * is2 + dw;
  const double rs1 = -mu * is1 - s.nu1, rs2 = -mu * is2 - s.nu2;
  const double cg1 = tm.g1 - s.s1, cg2 = tm.g2 - s.s2;
  int bad = 0;
#ifdef DF_STAMPS
  long long st_[5] = {0, 0, 0, 0, 0};
#define STAMP(i_) do { __builtin_amdgcn_sched_barrier(0); long long t1__ = clock64(); __builtin_amdgcn_sched_barrier(0); st_[i_] += t1__ - t0__; t0__ = t1__; } while (0)
#else
#define STAMP(i_) do { } while (0)
#endif
  auto body = [&](InQM &cm, int k) __attribute__((always_inline)) {
#ifdef DF_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    long long t0__ = clock64();
    __builtin_amdgcn_sched_barrier(0);
#endif
    gdbl *sp = t_.st(k);
    InQV cv;
    loadQV(t_, k, cv);                    // in flight while the congruence runs
    const double *G = cm.G, *E = cm.E, *H = cm.H;
    if (FORM == 1 && k < K - 1) {     // step k+1 does not see angle_k: drop its row and column
      ASC_UNROLL
      for (int i = 0; i < 7; i++) P[sid(IA, i)] = 0.0;
      p0[IA] = 0.0; p1[IA] = 0.0; p2[IA] = 0.0;
    }
    if (SCHEME == 1 && k < K - 1) {   // pull the value function of step k+1 back through Abar = I + cs*F_z(z_k)
      congruence_abar(P, G, cs);
      double t_[7];
      fzt_lambda(G, p0, t_);
      ASC_UNROLL
      for (int i = 0; i < 7; i++) p0[i] += cs * t_[i];
      fzt_lambda(G, p1, t_);
      ASC_UNROLL
      for (int i = 0; i < 7; i++) p1[i] += cs * t_[i];
      fzt_lambda(G, p2, t_);
      ASC_UNROLL
      for (int i = 0; i < 7; i++) p2[i] += cs * t_[i];
    }
    P[sid(IX, IX)] += H[0]; P[sid(IX, IY)] += H[1]; P[sid(IX, IA)] += H[2]; P[sid(IX, IM)] += H[3];
    P[sid(IY, IY)] += H[4]; P[sid(IY, IA)] += H[5]; P[sid(IY, IM)] += H[6];
    P[sid(IA, IA)] += H[7]; P[sid(IA, IM)] += H[8]; P[sid(IM, IM)] += H[9];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) P[sid(i, i)] += dw;
    if (k == K - 1) terminal_hessian(P, tm, s.nu3, s.nu1, s.nu2, sig1, sig2);
    congruence<FORM>(P, G, E, cs);
    const double D = cm.R0 + dw + bu * bu * P[sid(IB, IB)];
    if (!(D > 0.0)) bad = 1;
    const double iD = rcp(D);
    double mw[7], kap[7];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { mw[i] = bu * P[sid(i, IB)]; kap[i] = mw[i] * iD; }
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      ASC_UNROLL
      for (int j = i; j < 7; j++) P[sid(i, j)] -= mw[i] * kap[j];
    }
    stn<7>(t_, sp, Q_KA, kap);
    STAMP(0);        // matrix part: N assembly, congruence, pivot, P update (waits for the matrix loads)
    // right-hand sides 0: residual, 1: -B_theta, 2: -B_nu3
    double *rz = cv.rz;
    const double *cc = cv.cc;
    rz[IA] += mu * cv.bza;
    rz[IM] += mu * cv.bzm;
    const double ru = cv.ru0 + mu * cv.bu;
    const double gu = FORM == 1 ? 0.0 : cv.ru0 * ith;
    if (k == K - 1) {
      const double w1 = s.nu1 + sig1 * cg1 + rs1, w2 = s.nu2 + sig2 * cg2 + rs2;
      rz[IX] += s.nu3 * tm.e3g[0] + w1 * tm.g1g[0];
      rz[IY] += s.nu3 * tm.e3g[1] + w1 * tm.g1g[1];
      rz[IVX] += s.nu3 * tm.e3g[2] + w2 * tm.g2g[0];
      rz[IVY] += s.nu3 * tm.e3g[3] + w2 * tm.g2g[1];
    }
    double n[7], nt[7], q0[7], q1[7], rc1[7], Prc[7], k00, k01, k02;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) n[i] = -rz[i] + p0[i];
    solveAT<FORM>(G, E, cs, n, nt);
    k00 = (bu * nt[IB] - ru) * iD;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { q0[i] = nt[i] - mw[i] * k00; n[i] = -cc[i]; }
    symv(P, n, Prc);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) p0[i] = q0[i] - Prc[i];
    STAMP(1);        // right-hand side 0 (waits for the vector loads)
    ASC_UNROLL
    for (int i = 0; i < 7; i++) n[i] = -cv.gt[i] + p1[i];
    solveAT<FORM>(G, E, cs, n, nt);
    k01 = (bu * nt[IB] - gu) * iD;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { q1[i] = nt[i] - mw[i] * k01; rc1[i] = hT * cv.F[i]; }
    symv(P, rc1, Prc);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) p1[i] = q1[i] - Prc[i];
    cpy<7>(n, p2);
    if (k == K - 1) { n[IX] -= tm.e3g[0]; n[IY] -= tm.e3g[1]; n[IVX] -= tm.e3g[2]; n[IVY] -= tm.e3g[3]; }
    solveAT<FORM>(G, E, cs, n, nt);
    k02 = bu * nt[IB] * iD;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) p2[i] = nt[i] - mw[i] * k02;
    ROW(sp, Q_K0) = k00; ROW(sp, Q_K0 + 1) = k01; ROW(sp, Q_K0 + 2) = k02;
    STAMP(2);        // right-hand sides 1 and 2
    double a10 = D * k01 * k00, a11 = D * k01 * k01, a12 = D * k01 * k02, a20 = D * k02 * k00;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      a10 += 0.5 * (rc1[i] * (q0[i] + p0[i]) - cc[i] * (q1[i] + p1[i]));
      a11 += rc1[i] * (q1[i] + p1[i]);
      a12 += rc1[i] * p2[i];
      a20 -= cc[i] * p2[i];
    }
    S10 += a10; S11 += a11; S12 += a12; S20 += a20; S22 += D * k02 * k02;
    STAMP(3);        // Schur sums
  };
#ifdef DF_STAMPS
  const long long tsw0_ = clock64();
#endif
#define LD_(k_, buf_) loadQM(t_, k_, buf_)
  ASC_SWEEP_BACKWARD(InQM, LD_, body)
#undef LD_
#ifdef DF_STAMPS
  if (threadIdx.x == 0 && SCHEME == 0) {
    st_[4] = clock64() - tsw0_;
    for (int i = 0; i < 5; i++) atomicAdd(&g_df_stamps[i], (unsigned long long)st_[i]);
    atomicAdd(&g_df_stamps[5], 1ull);
  }
#endif
  int ok = !bad;
  double dth = 0.0, dnu3 = 0.0;
  if (ok) {
    const double itl = rcp(s.th - d.tlb), itu = rcp(d.tub - s.th);
    const double rthp = dc.rth + mu * (itu - itl);   // d/dtheta of the barrier Lagrangian
    const double sth = s.zlt * itl + s.zut * itu + dw;
    const double a11 = sth - S11, a12 = -S12, a22 = -S22;
    const double b1 = -rthp + S10, b2 = -tm.e3 + S20;
    const double det = a11 * a22 - a12 * a12;
    if (det < 0.0) {
      const double idet = 1.0 / det;
      dth = (b1 * a22 - a12 * b2) * idet;
      dnu3 = (a11 * b2 - a12 * b1) * idet;
    } else {
      ok = 0;
    }
  }
  if (ok) {
    SC(X_DTH) = dth; SC(X_DNU3) = dnu3; SC(X_SIG1) = sig1; SC(X_SIG2) = sig2; SC(X_RS1) = rs1; SC(X_RS2) = rs2;
    SC(X_DWL) = dw;
    SC(X_STATE) = ST_FACTORED;
    atomicAdd(&counters[1], 1);
  } else {   // wrong inertia: raise the primal regularisation; factorised again in the next round
    const double dwl = SC(X_DWL);
    const double ndw = next_delta_w(dw, dwl);
    if (ndw > 1e10) {
      SC(X_STATUS) = ASCENT_REGULARISATION_FAILED; SC(X_STATE) = ST_DONE;
    } else {
      SC(X_DW) = ndw; SC(X_STATE) = ST_FACTOR;
      atomicAdd(&counters[0], 1);
      atomicAdd(&counters[2], 1);
    }
  }
}

// ==============================================================================================
// q_factor_wide: the backward factorisation with 16 lanes per NLP
// ==============================================================================================
// q_decide_factor gives every NLP one lane: 64 wavefronts for 4096 NLPs, each issuing ~1100 FP64 instructions
// per step, on a chip with 1024 SIMDs.  Here an NLP owns a row of 16 lanes (the DPP row of gfx950): lanes 0-6
// hold one column each of the 7x7 value-function matrix P, lanes 7-9 the three right-hand-side vectors, so the
// same instruction stream -- gather the node's terms, v <- A^-T v, pivot update v -= mw * coef -- advances all
// ten vectors at once.  N <- A^-T N A^-1 is two column-wise A^-T solves with a 7x7 transpose through LDS in
// between; the pivot column is handed round with DPP row broadcasts; P*rc comes from column dot products (P is
// symmetric) moved to the right-hand-side lanes through LDS.  One wavefront = 4 NLPs, one workgroup = 4
// wavefronts = 16 NLPs (a quarter tile: every 512-byte workspace row is read as four 128-byte pieces).
// ~4x fewer instructions per step on the critical path, 16x more wavefronts.
constexpr int WIDE_NLP_PER_BLOCK = 16, WIDE_THREADS = 256;

template <int SRC>
ASC_DEV double bcast16(double v) {   // lane SRC of this lane's row of 16, to the whole row (v_mov_b64_dpp row_newbcast)
  const long x = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(long, v), 0x150 + SRC, 0xf, 0xf, false);
  return __builtin_bit_cast(double, x);
}

// One step record as a wavefront of q_factor_wide loads it: 7 gathers private to the lane's vector, and two in
// which the 16 lanes of an NLP fetch 16 different rows that all of them need (G, E, R0, ru0, bu, bza | c, F,
// bzm) and then hand round with row broadcasts -- 9 loads per step instead of 40 (the path from L1 to the
// registers, shared by the four wavefronts of a CU, is what the replicated loads saturated).
struct InW {
  double gq[7], gA, gB;
};

template <int SCHEME, int FORM>
__global__ __launch_bounds__(WIDE_THREADS) void q_factor_wide(const ascent_params *params, long batch, Geo g,
                                                              double *ws, int max_iter, double tol, int *counters) {
  __shared__ double lds_t[WIDE_THREADS / 16][7][7];     // [group][column][row]
  __shared__ double lds_d[WIDE_THREADS / 16][2][8];     // [group][rhs][row]
  const int grp = threadIdx.x >> 4, role = threadIdx.x & 15;
  const long p = (long)blockIdx.x * WIDE_NLP_PER_BLOCK + grp;
  const unsigned L = (unsigned)(p & (WAVE - 1));
  const QTile t_((gdbl *)ws + (size_t)(blockIdx.x >> 2) * g.tile_doubles(), L);
  gdbl *sc = scal_base(t_, g);
  if (p >= batch) return;
  const Der d = derive(params[p]);
  const int K = g.K;
  // the decisions of the round, computed by all 16 lanes of the NLP, written by lane 0 (no separate launch)
  const Decided dc = decide_lane(t_, sc, g, d, max_iter, tol, counters, role == 0);
  if (dc.state != ST_FACTOR) return;
  const double mu = dc.mu;
  const unsigned oc = L + (unsigned)dc.cur * (21 * WAVE);
  const Scal s = dc.s;
  const double dw = dc.dw;
  const double hT = (1.0 / K) * d.T, dt = hT * s.th, be = dt * d.alpha, ith = rcp(s.th);
  const double cs = SCHEME == 1 ? 0.5 * dt : dt;
  constexpr int IB = FORM == 1 ? IA : IW;
  const double bu = FORM == 1 ? 0.5 * d.aub : be;
  const bool col = role < 7, rhs = role >= 7 && role < 10;
  // what this lane gathers from a step record for row i of its vector, and with which sign
  unsigned goff[7];
  double gsc[7];
  {
    constexpr int hmap[7] = {0, 1, -1, -1, 2, -1, 3};   // position of a state among (x, y, angle, mass)
    constexpr int hrow[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      int row = Q_H; double sgn = 0.0;
      if (col) {
        ASC_UNROLL
        for (int c = 0; c < 7; c++)
          if (c == role && hmap[i] >= 0 && hmap[c] >= 0) { row = Q_H + hrow[hmap[i]][hmap[c]]; sgn = 1.0; }
      } else if (role == 7) { row = Q_RZ + i; sgn = -1.0; }
      else if (role == 8) { row = Q_GT + i; sgn = -1.0; }
      goff[i] = (unsigned)row * WAVE + L;
      gsc[i] = sgn;
    }
  }
  const double bsc = role == 7 ? -mu : 0.0;           // barrier terms of the residual right-hand side
  const int rowA = role < 8 ? Q_G + role : role < 12 ? Q_E + role - 8 : role == 12 ? Q_SC : role == 13 ? Q_SC + 1
                   : role == 14 ? Q_SC + 4 : Q_SC + 2;
  const int rowB = role < 7 ? Q_C + role : role < 14 ? Q_F + role - 7 : role == 14 ? Q_SC + 3 : Q_SC;
  const unsigned offA = (unsigned)rowA * WAVE + L, offB = (unsigned)rowB * WAVE + L;
  const unsigned offK = (unsigned)(role < 7 ? Q_KA + role : role < 10 ? Q_K0 + role - 7 : Q_R) * WAVE + L;
  auto loadW = [&](int k, InW &in) __attribute__((always_inline)) {
    const gdbl *sp = t_.st(k);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) in.gq[i] = sp[goff[i]];
    in.gA = sp[offA];
    in.gB = sp[offB];
  };
  double a[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) a[i] = 0.0;
  double U = 0.0, V = 0.0, k10 = 0.0, k11 = 0.0, k12 = 0.0, k20 = 0.0, k22 = 0.0;
  int bad = 0;
  // terminal node: Hessian of the terminal Lagrangian + slack-eliminated barrier terms (column lanes), the
  // terminal parts of the right-hand sides (lanes 7 and 9)
  double zK[7];
  ldo<7>(t_.st(K - 1), Q_IT + O_Z, oc, zK);
  const Terminal tm = terminal_eval(d, zK);
  const double is1 = rcp(s.s1), is2 = rcp(s.s2);
  const double sig1 = s.zs1 * is1 + dw, sig2 = s.zs2 * is2 + dw;
  const double rs1 = -mu * is1 - s.nu1, rs2 = -mu * is2 - s.nu2;
  {
    const double cg1 = tm.g1 - s.s1, cg2 = tm.g2 - s.s2;
    double Qt[28];
    ASC_UNROLL
    for (int i = 0; i < 28; i++) Qt[i] = 0.0;
    terminal_hessian(Qt, tm, s.nu3, s.nu1, s.nu2, sig1, sig2);
    const double w1 = s.nu1 + sig1 * cg1 + rs1, w2 = s.nu2 + sig2 * cg2 + rs2;
    const double r0[4] = {s.nu3 * tm.e3g[0] + w1 * tm.g1g[0], s.nu3 * tm.e3g[1] + w1 * tm.g1g[1],
                          s.nu3 * tm.e3g[2] + w2 * tm.g2g[0], s.nu3 * tm.e3g[3] + w2 * tm.g2g[1]};
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      double v = 0.0;
      ASC_UNROLL
      for (int c = 0; c < 7; c++) v = role == c ? Qt[sid(i, c)] : v;
      if (i < 4) { v = role == 7 ? -r0[i] : v; v = role == 9 ? -tm.e3g[i] : v; }
      a[i] = v;
    }
  }
  auto body = [&](InW &in, int k) __attribute__((always_inline)) {
    gdbl *sp = t_.st(k);
    if (FORM == 1 && k < K - 1) {     // step k+1 does not see angle_k: drop its row and column
      a[IA] = 0.0;
      if (role == IA) {
        ASC_UNROLL
        for (int i = 0; i < 7; i++) a[i] = 0.0;
      }
    }
    const double G[8] = {bcast16<0>(in.gA), bcast16<1>(in.gA), bcast16<2>(in.gA), bcast16<3>(in.gA),
                         bcast16<4>(in.gA), bcast16<5>(in.gA), bcast16<6>(in.gA), bcast16<7>(in.gA)};
    const double E[4] = {bcast16<8>(in.gA), bcast16<9>(in.gA), bcast16<10>(in.gA), bcast16<11>(in.gA)};
    const double R0 = bcast16<12>(in.gA), ru0 = bcast16<13>(in.gA), bur = bcast16<14>(in.gA);
    const double bza = bcast16<15>(in.gA), bzm = bcast16<14>(in.gB);
    const double cc[7] = {bcast16<0>(in.gB), bcast16<1>(in.gB), bcast16<2>(in.gB), bcast16<3>(in.gB),
                          bcast16<4>(in.gB), bcast16<5>(in.gB), bcast16<6>(in.gB)};
    const double rc1[7] = {hT * bcast16<7>(in.gB), hT * bcast16<8>(in.gB), hT * bcast16<9>(in.gB),
                           hT * bcast16<10>(in.gB), hT * bcast16<11>(in.gB), hT * bcast16<12>(in.gB),
                           hT * bcast16<13>(in.gB)};
    if (SCHEME == 1 && k < K - 1) {   // pull the value function of step k+1 back through Abar = I + cs*F_z(z_k):
      double t[7];                    // Abar' on every column and right-hand side, transpose, Abar' on the columns again
      fzt_lambda(G, a, t);
      ASC_UNROLL
      for (int i = 0; i < 7; i++) a[i] += cs * t[i];
      if (col) {
        ASC_UNROLL
        for (int i = 0; i < 7; i++) lds_t[grp][role][i] = a[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (col) {
        double r[7];
        ASC_UNROLL
        for (int l = 0; l < 7; l++) r[l] = lds_t[grp][l][role];
        fzt_lambda(G, r, t);
        ASC_UNROLL
        for (int i = 0; i < 7; i++) a[i] = r[i] + cs * t[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // lds_t is written again below
      __builtin_amdgcn_wave_barrier();
    }
    ASC_UNROLL
    for (int i = 0; i < 7; i++) a[i] += gsc[i] * in.gq[i];
    a[IA] += bsc * bza;
    a[IM] += bsc * bzm;
    if (dw != 0.0) {
      ASC_UNROLL
      for (int i = 0; i < 7; i++) a[i] += role == i ? dw : 0.0;
    }
    double b[7];
    solveAT<FORM>(G, E, cs, a, b);                       // columns of T = A^-T N ; nt = A^-T n on lanes 7-9
    if (col) {
      ASC_UNROLL
      for (int i = 0; i < 7; i++) lds_t[grp][role][i] = b[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (col) {
      double t[7];
      ASC_UNROLL
      for (int l = 0; l < 7; l++) t[l] = lds_t[grp][l][role];   // row `role` of T = column of T'
      solveAT<FORM>(G, E, cs, t, b);                     // M = A^-T T'
    }
    double mw[7];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) mw[i] = bu * bcast16<IB>(b[i]);
    const double D = R0 + dw + bu * mw[IB];
    if (!(D > 0.0)) bad = 1;
    const double iD = rcp(D);
    const double ru = ru0 + mu * bur, gu = FORM == 1 ? 0.0 : ru0 * ith;
    const double rsel = role == 7 ? ru : role == 8 ? gu : 0.0;
    const double coef = (bu * b[IB] - rsel) * iD;        // kap_c on column lanes, k0_j on lanes 7-9
    ASC_UNROLL
    for (int i = 0; i < 7; i++) a[i] = b[i] - mw[i] * coef;   // P column / q_j
    sp[offK] = coef;      // lanes 10-15 all write into the first row of Q_R, which q_local rewrites before anything reads it
    // P rc_0, P rc_1 (rc_0 = -c, rc_1 = hT F, rc_2 = 0): by symmetry element c is the dot product with column c
    if (col) {
      double d0 = 0.0, d1 = 0.0;
      ASC_UNROLL
      for (int i = 0; i < 7; i++) { d0 -= a[i] * cc[i]; d1 += a[i] * rc1[i]; }
      lds_d[grp][0][role] = d0;
      lds_d[grp][1][role] = d1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (rhs) {
      double prc[7];
      ASC_UNROLL
      for (int i = 0; i < 7; i++) prc[i] = role == 9 ? 0.0 : lds_d[grp][role == 8 ? 1 : 0][i];
      double u = 0.0, v = 0.0;
      ASC_UNROLL
      for (int i = 0; i < 7; i++) {
        const double pj = a[i] - prc[i], sj = a[i] + pj;
        u += rc1[i] * sj; v += cc[i] * sj;
        a[i] = pj;
      }
      U += u; V += v;
    }
    const double k00 = bcast16<7>(coef), k01 = bcast16<8>(coef), k02 = bcast16<9>(coef);
    const double Dk1 = D * k01, Dk2 = D * k02;
    k10 += Dk1 * k00; k11 += Dk1 * k01; k12 += Dk1 * k02; k20 += Dk2 * k00; k22 += Dk2 * k02;
  };
#define LD_(k_, buf_) loadW(k_, buf_)
  ASC_SWEEP_BACKWARD4U(InW, LD_, body)
#undef LD_
  const double U0 = bcast16<7>(U), U1 = bcast16<8>(U), V1 = bcast16<8>(V), U2 = bcast16<9>(U), V2 = bcast16<9>(V);
  if (role != 0) return;
  const double S10 = k10 + 0.5 * (U0 - V1), S11 = k11 + U1, S12 = k12 + 0.5 * U2, S20 = k20 - 0.5 * V2, S22 = k22;
  int ok = !bad;
  double dth = 0.0, dnu3 = 0.0;
  if (ok) {
    const double itl = rcp(s.th - d.tlb), itu = rcp(d.tub - s.th);
    const double rthp = dc.rth + mu * (itu - itl);
    const double sth = s.zlt * itl + s.zut * itu + dw;
    const double a11 = sth - S11, a12 = -S12, a22 = -S22;
    const double b1 = -rthp + S10, b2 = -tm.e3 + S20;
    const double det = a11 * a22 - a12 * a12;
    if (det < 0.0) {
      const double idet = 1.0 / det;
      dth = (b1 * a22 - a12 * b2) * idet;
      dnu3 = (a11 * b2 - a12 * b1) * idet;
    } else {
      ok = 0;
    }
  }
  if (ok) {
    SC(X_DTH) = dth; SC(X_DNU3) = dnu3; SC(X_SIG1) = sig1; SC(X_SIG2) = sig2; SC(X_RS1) = rs1; SC(X_RS2) = rs2;
    SC(X_DWL) = dw;
    SC(X_STATE) = ST_FACTORED;
    atomicAdd(&counters[1], 1);
  } else {
    const double dwl = SC(X_DWL);
    const double ndw = next_delta_w(dw, dwl);
    if (ndw > 1e10) {
      SC(X_STATUS) = ASCENT_REGULARISATION_FAILED; SC(X_STATE) = ST_DONE;
    } else {
      SC(X_DW) = ndw; SC(X_STATE) = ST_FACTOR;
      atomicAdd(&counters[0], 1);
      atomicAdd(&counters[2], 1);
    }
  }
}

// ==============================================================================================
// q_forward: the forward recurrence only (primal step dz, du)
// ==============================================================================================
struct InQF {
  double G[8], E[4], cc[7], F[7], ka[7], k0[3];
};
ASC_DEV void loadQF(const QTile &t_, int k, InQF &in) {
  const gdbl *sp = t_.st(k);
  ldn<8>(t_, sp, Q_G, in.G);
  ldn<4>(t_, sp, Q_E, in.E);
  ldn<7>(t_, sp, Q_C, in.cc);
  ldn<7>(t_, sp, Q_F, in.F);
  ldn<7>(t_, sp, Q_KA, in.ka);
  ldn<3>(t_, sp, Q_K0, in.k0);
}

template <int SCHEME, int FORM>
__global__ __launch_bounds__(WAVE) void q_forward(const ascent_params *params, long batch, Geo g, double *ws) {
  const long p = (long)blockIdx.x * WAVE + threadIdx.x;
  if (p >= batch) return;
  const QTile t_((gdbl *)ws + (size_t)blockIdx.x * g.tile_doubles());
  gdbl *sc = scal_base(t_, g);
  if ((int)SC(X_STATE) != ST_FACTORED) return;
  const Der d = derive(params[p]);
  const int K = g.K;
  const double th = SC(X_S + S_TH), dth = SC(X_DTH), dnu3 = SC(X_DNU3);
  const double hT = (1.0 / K) * d.T, dt = hT * th, be = dt * d.alpha;
  const double cs = SCHEME == 1 ? 0.5 * dt : dt;
  double dzp[7], Gp[8];          // previous step's dz; for the trapezoid also its Jacobian block (Abar_k = I + cs*F_z(z_{k-1}))
  ASC_UNROLL
  for (int i = 0; i < 7; i++) dzp[i] = 0.0;
  ASC_UNROLL
  for (int i = 0; i < 8; i++) Gp[i] = 0.0;
  auto body = [&](InQF &cur_, int k) __attribute__((always_inline)) {
    gdbl *sp = t_.st(k);
    double xi[7], dz[7];
    double du = cur_.k0[0] + cur_.k0[1] * dth + cur_.k0[2] * dnu3;
    if (SCHEME == 1) {
      double t2_[7];
      fz_mul(Gp, dzp, t2_);
      ASC_UNROLL
      for (int i = 0; i < 7; i++) dzp[i] += cs * t2_[i];
    }
    if (FORM == 1) dzp[IA] = 0.0;     // the angle row has no coupling to the previous angle
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      xi[i] = dzp[i] - cur_.cc[i] + hT * cur_.F[i] * dth;
      du -= cur_.ka[i] * xi[i];
    }
    if (FORM == 1) xi[IA] += 0.5 * d.aub * du; else xi[IW] += be * du;
    solveA<FORM>(cur_.G, cur_.E, cs, xi, dz);
    stn<7>(t_, sp, Q_ST + O_Z, dz);
    ROW(sp, Q_ST + O_U) = du;
    cpy<7>(dzp, dz);
    if (SCHEME == 1) cpy<8>(Gp, cur_.G);
  };
#define LD_(k_, buf_) loadQF(t_, k_, buf_)
  ASC_SWEEP_FORWARD4(InQF, LD_, body)
#undef LD_
}

// ==============================================================================================
// q_local: everything of the step that has no recurrence -- adjoint right-hand side, bound-multiplier
// steps, both fraction-to-boundary rules, barrier slope.  One wavefront = 64 NLPs x CHUNK steps.
// ==============================================================================================
__global__ __launch_bounds__(WAVE) void q_local(const ascent_params *params, long batch, Geo g, double *ws) {
  const long p = (long)blockIdx.x * WAVE + threadIdx.x;
  if (p >= batch) return;
  const QTile t_((gdbl *)ws + (size_t)blockIdx.x * g.tile_doubles());
  gdbl *sc = scal_base(t_, g);
  if ((int)SC(X_STATE) != ST_FACTORED) return;
  const Der d = derive(params[p]);
  const int K = g.K, chunk = blockIdx.y;
  const int k_lo = chunk * CHUNK, k_hi = min(K, k_lo + CHUNK) - 1;
  const unsigned oc = buf_off(t_, (int)SC(X_CUR));
  const Scal s = load_scal(t_, sc, X_S);
  const double mu = SC(X_MU), dw = SC(X_DWL), dth = SC(X_DTH), dnu3 = SC(X_DNU3);
  const double tau = fmax(0.99, 1.0 - mu);
  double rmax = 0.0, gsum = 0.0, adu = 1.0, ccl = 0.0;
  for (int k = k_lo; k <= k_hi; k++) {
    gdbl *sp = t_.st(k);
    double H[10], rz[7], gt[7], dz[7], zb[6];
    {   // defects . multipliers, half of the curvature estimate c'(lambda + dlambda)
      double cc[7], l[7];
      ldn<7>(t_, sp, Q_C, cc);
      ldo<7>(sp, Q_IT + O_L, oc, l);
      ASC_UNROLL
      for (int i = 0; i < 7; i++) ccl += cc[i] * l[i];
    }
    ldn<10>(t_, sp, Q_H, H);
    ldn<7>(t_, sp, Q_RZ, rz);
    ldn<7>(t_, sp, Q_GT, gt);
    ldn<7>(t_, sp, Q_ST + O_Z, dz);
    ldo<6>(sp, Q_IT + O_ZB, oc, zb);
    const double du = ROW(sp, Q_ST + O_U);
    const double bza = ROW(sp, Q_SC + 2), bzm = ROW(sp, Q_SC + 3);
    const double a = ROWO(sp, Q_IT + O_Z + IA, oc), m = ROWO(sp, Q_IT + O_Z + IM, oc), u = ROWO(sp, Q_IT + O_U, oc);
    // right-hand side of the adjoint recursion: -(rz + gt*dtheta) - Q dz   (dlambda_{k+1} is added by q_adjoint)
    double r[7];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) r[i] = -rz[i] - gt[i] * dth - dw * dz[i];
    r[IA] -= mu * bza;
    r[IM] -= mu * bzm;
    r[IX] -= H[0] * dz[IX] + H[1] * dz[IY] + H[2] * dz[IA] + H[3] * dz[IM];
    r[IY] -= H[1] * dz[IX] + H[4] * dz[IY] + H[5] * dz[IA] + H[6] * dz[IM];
    r[IA] -= H[2] * dz[IX] + H[5] * dz[IY] + H[7] * dz[IA] + H[8] * dz[IM];
    r[IM] -= H[3] * dz[IX] + H[6] * dz[IY] + H[8] * dz[IA] + H[9] * dz[IM];
    if (k == K - 1) {
      double zK[7], QT[28], qd[7];
      ldo<7>(sp, Q_IT + O_Z, oc, zK);
      const Terminal tm = terminal_eval(d, zK);
      const double sig1 = SC(X_SIG1), sig2 = SC(X_SIG2), rs1 = SC(X_RS1), rs2 = SC(X_RS2);
      ASC_UNROLL
      for (int i = 0; i < 28; i++) QT[i] = 0.0;
      terminal_hessian(QT, tm, s.nu3, s.nu1, s.nu2, sig1, sig2);
      symv(QT, dz, qd);
      ASC_UNROLL
      for (int i = 0; i < 7; i++) r[i] -= qd[i];
      const double w1 = s.nu1 + sig1 * (tm.g1 - s.s1) + rs1, w2 = s.nu2 + sig2 * (tm.g2 - s.s2) + rs2;
      r[IX] -= s.nu3 * tm.e3g[0] + w1 * tm.g1g[0] + tm.e3g[0] * dnu3;
      r[IY] -= s.nu3 * tm.e3g[1] + w1 * tm.g1g[1] + tm.e3g[1] * dnu3;
      r[IVX] -= s.nu3 * tm.e3g[2] + w2 * tm.g2g[0] + tm.e3g[2] * dnu3;
      r[IVY] -= s.nu3 * tm.e3g[3] + w2 * tm.g2g[1] + tm.e3g[3] * dnu3;
    }
    stn<7>(t_, sp, Q_R, r);
    const double id[6] = {rcp(a), rcp(d.aub - a), rcp(m), rcp(1.0 - m), rcp(u + 1.0), rcp(1.0 - u)};
    ASC_FTBR(rmax, id[0], dz[IA]); ASC_FTBR(rmax, id[1], -dz[IA]);
    ASC_FTBR(rmax, id[2], dz[IM]); ASC_FTBR(rmax, id[3], -dz[IM]);
    ASC_FTBR(rmax, id[4], du); ASC_FTBR(rmax, id[5], -du);
    gsum += dz[IA] * (id[1] - id[0]) + dz[IM] * (id[3] - id[2]) + du * (id[5] - id[4]);
    const double dx[3] = {dz[IA], dz[IM], du};
    double dzb[6];
    ASC_UNROLL
    for (int b = 0; b < 3; b++) {
      const double zl = zb[2 * b], zu = zb[2 * b + 1];
      dzb[2 * b] = id[2 * b] * (mu - zl * dx[b]) - zl;
      dzb[2 * b + 1] = id[2 * b + 1] * (mu + zu * dx[b]) - zu;
      ASC_FTB(adu, zl, dzb[2 * b]);
      ASC_FTB(adu, zu, dzb[2 * b + 1]);
    }
    stn<6>(t_, sp, Q_ST + O_ZB, dzb);
  }
  gdbl *pp = part_base(t_, g, chunk);       // reuse partial rows 0..2 (the trial partials are consumed by now)
  ROW(pp, 0) = rmax; ROW(pp, 1) = gsum; ROW(pp, 2) = adu; ROW(pp, 3) = ccl;
}

// ==============================================================================================
// q_adjoint: multiplier step, then the step-size / merit bookkeeping of the iteration
// ==============================================================================================
struct InQA {
  double G[8], E[4], r[7], cc[7];
};
ASC_DEV void loadQA(const QTile &t_, int k, InQA &in) {
  const gdbl *sp = t_.st(k);
  ldn<8>(t_, sp, Q_G, in.G);
  ldn<4>(t_, sp, Q_E, in.E);
  ldn<7>(t_, sp, Q_R, in.r);
  ldn<7>(t_, sp, Q_C, in.cc);
}

template <int SCHEME, int FORM>
__global__ __launch_bounds__(WAVE) void q_adjoint(const ascent_params *params, long batch, Geo g, double *ws) {
  const long p = (long)blockIdx.x * WAVE + threadIdx.x;
  if (p >= batch) return;
  const QTile t_((gdbl *)ws + (size_t)blockIdx.x * g.tile_doubles());
  gdbl *sc = scal_base(t_, g);
  if ((int)SC(X_STATE) != ST_FACTORED) return;
  const Der d = derive(params[p]);
  const int K = g.K;
  const unsigned oc = buf_off(t_, (int)SC(X_CUR));
  const Scal s = load_scal(t_, sc, X_S);
  const double mu = SC(X_MU), dth = SC(X_DTH), dnu3 = SC(X_DNU3);
  const double sig1 = SC(X_SIG1), sig2 = SC(X_SIG2), rs1 = SC(X_RS1), rs2 = SC(X_RS2);
  const double hT = (1.0 / K) * d.T, dt = hT * s.th;
  const double cs = SCHEME == 1 ? 0.5 * dt : dt;
  const double tau = fmax(0.99, 1.0 - mu);
  double dln[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) dln[i] = 0.0;
  double cl = 0.0;
  auto body = [&](InQA &cur_, int k) __attribute__((always_inline)) {
    gdbl *sp = t_.st(k);
    double r[7], dl[7];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) r[i] = (FORM == 1 && i == IA) ? cur_.r[i] : cur_.r[i] + dln[i];
    if (SCHEME == 1) {           // Abar_{k+1}' dlambda_{k+1} = dlambda_{k+1} + cs * F_z(z_k)' dlambda_{k+1}
      double t2_[7];
      fzt_lambda(cur_.G, dln, t2_);
      ASC_UNROLL
      for (int i = 0; i < 7; i++) r[i] += cs * t2_[i];
    }
    solveAT<FORM>(cur_.G, cur_.E, cs, r, dl);
    stn<7>(t_, sp, Q_ST + O_L, dl);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) cl += cur_.cc[i] * dl[i];
    cpy<7>(dln, dl);
  };
#define LD_(k_, buf_) loadQA(t_, k_, buf_)
  ASC_SWEEP_BACKWARD4(InQA, LD_, body)
#undef LD_
  // partials of q_local (chunks in forward order, as the fused kernel's forward pass accumulates)
  double rmax = 0.0, gsum = 0.0, adu = 1.0;
  for (int c = 0; c < g.nch; c++) {
    const gdbl *pp = part_base(t_, g, c);
    rmax = fmax(rmax, ROW(pp, 0)); gsum += ROW(pp, 1); adu = fmin(adu, ROW(pp, 2)); cl += ROW(pp, 3);
  }
  // scalars of the step
  double zK[7], dzK[7];
  ldo<7>(t_.st(K - 1), Q_IT + O_Z, oc, zK);
  ldn<7>(t_, t_.st(K - 1), Q_ST + O_Z, dzK);
  const Terminal tm = terminal_eval(d, zK);
  Scal ds;
  ds.th = dth; ds.nu3 = dnu3;
  ds.s1 = (tm.g1 - s.s1) + tm.g1g[0] * dzK[IX] + tm.g1g[1] * dzK[IY];
  ds.s2 = (tm.g2 - s.s2) + tm.g2g[0] * dzK[IVX] + tm.g2g[1] * dzK[IVY];
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
  double gd = mu * gsum;
  gd += ds.th * (1.0 - mu / dl_ + mu / dU) - mu * ds.s1 / s.s1 - mu * ds.s2 / s.s2;
  cl += tm.e3 * (s.nu3 + ds.nu3) + (tm.g1 - s.s1) * (s.nu1 + ds.nu1) + (tm.g2 - s.s2) * (s.nu2 + ds.nu2);
  // l1 merit: penalty update (Nocedal & Wright eq. 18.36), reference value, slope
  const double c1 = SC(X_C1), slog = SC(X_SL);
  double nu_pen = SC(X_NUP);
  const double curv = -gd + cl;
  if (c1 > 0.0) {
    const double need = (gd + 0.5 * fmax(curv, 0.0)) / (0.9 * c1);
    if (nu_pen < need) nu_pen = need + 1.0;
  }
  store_scal(t_, sc, X_D, ds);
  SC(X_NUP) = nu_pen;
  SC(X_DM) = gd - nu_pen * c1;
  SC(X_PHI0) = s.th - mu * slog + nu_pen * c1;
  SC(X_ALPHA) = apr; SC(X_ADU) = adu; SC(X_LS) = 0.0;
  SC(X_STATE) = ST_TRIAL;
}

// ==============================================================================================
// q_forward_wide / q_adjoint_wide: the two substitution sweeps with 16 lanes per NLP
// ==============================================================================================
// The substitutions are short vector recurrences whose cost is fetching ~30 rows per step: with one lane per NLP
// that is ~30 load instructions per step and a wavefront cannot keep more than 64 in flight.  Here the 16 lanes
// of an NLP fetch 16 different rows with one gather, hand them round with row broadcasts and all compute the
// (tiny) step redundantly; lane i stores element i.
struct InV {
  double gA, gB, gC;
};
// v[role] without dynamic register indexing (which the compiler lowers to scratch memory): a dot product with
// the lane's one-hot mask, kept in registers for the whole sweep
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

template <int SCHEME, int FORM>
__global__ __launch_bounds__(WIDE_THREADS) void q_forward_wide(const ascent_params *params, long batch, Geo g,
                                                               double *ws) {
  const int grp = threadIdx.x >> 4, role = threadIdx.x & 15;
  const long p = (long)blockIdx.x * WIDE_NLP_PER_BLOCK + grp;
  const unsigned L = (unsigned)(p & (WAVE - 1));
  const QTile t_((gdbl *)ws + (size_t)(blockIdx.x >> 2) * g.tile_doubles(), L);
  gdbl *sc = scal_base(t_, g);
  if (p >= batch || (int)SC(X_STATE) != ST_FACTORED) return;
  const Der d = derive(params[p]);
  const int K = g.K;
  const double th = SC(X_S + S_TH), dth = SC(X_DTH), dnu3 = SC(X_DNU3);
  const double hT = (1.0 / K) * d.T, dt = hT * th, be = dt * d.alpha, cs = SCHEME == 1 ? 0.5 * dt : dt;
  const int rowA = role < 8 ? Q_G + role : role < 12 ? Q_E + role - 8 : role < 15 ? Q_K0 + role - 12 : Q_G;
  const int rowB = role < 7 ? Q_C + role : role < 14 ? Q_F + role - 7 : Q_C;
  const int rowC = role < 7 ? Q_KA + role : Q_KA;
  const unsigned offA = (unsigned)rowA * WAVE + L, offB = (unsigned)rowB * WAVE + L, offC = (unsigned)rowC * WAVE + L;
  // dz[0..6], du are rows Q_ST + 0..7; lanes 8-15 store what lane 0 stores (an unconditional store keeps the
  // compiler's s_waitcnt bookkeeping exact)
  const unsigned offS = (unsigned)(Q_ST + (role < 8 ? role : 0)) * WAVE + L;
  const OneHot<8> hot(role);
  double dzp[7], Gp[8];          // previous step's dz; for the trapezoid also its Jacobian block (Abar_k = I + cs*F_z(z_{k-1}))
  ASC_UNROLL
  for (int i = 0; i < 7; i++) dzp[i] = 0.0;
  ASC_UNROLL
  for (int i = 0; i < 8; i++) Gp[i] = 0.0;
  auto loadV = [&](int k, InV &in) __attribute__((always_inline)) {
    const gdbl *sp = t_.st(k);
    in.gA = sp[offA]; in.gB = sp[offB]; in.gC = sp[offC];
  };
  auto body = [&](InV &in, int k) __attribute__((always_inline)) {
    gdbl *sp = t_.st(k);
    const double G[8] = {bcast16<0>(in.gA), bcast16<1>(in.gA), bcast16<2>(in.gA), bcast16<3>(in.gA),
                         bcast16<4>(in.gA), bcast16<5>(in.gA), bcast16<6>(in.gA), bcast16<7>(in.gA)};
    const double E[4] = {bcast16<8>(in.gA), bcast16<9>(in.gA), bcast16<10>(in.gA), bcast16<11>(in.gA)};
    const double k0[3] = {bcast16<12>(in.gA), bcast16<13>(in.gA), bcast16<14>(in.gA)};
    const double cc[7] = {bcast16<0>(in.gB), bcast16<1>(in.gB), bcast16<2>(in.gB), bcast16<3>(in.gB),
                          bcast16<4>(in.gB), bcast16<5>(in.gB), bcast16<6>(in.gB)};
    const double F[7] = {bcast16<7>(in.gB), bcast16<8>(in.gB), bcast16<9>(in.gB), bcast16<10>(in.gB),
                         bcast16<11>(in.gB), bcast16<12>(in.gB), bcast16<13>(in.gB)};
    const double ka[7] = {bcast16<0>(in.gC), bcast16<1>(in.gC), bcast16<2>(in.gC), bcast16<3>(in.gC),
                          bcast16<4>(in.gC), bcast16<5>(in.gC), bcast16<6>(in.gC)};
    __builtin_amdgcn_sched_barrier(0);   // all broadcasts first: a DPP result consumed at once stalls the wavefront
    double xi[7], dz[8];
    if (SCHEME == 1) {
      double t2_[7];
      fz_mul(Gp, dzp, t2_);
      ASC_UNROLL
      for (int i = 0; i < 7; i++) dzp[i] += cs * t2_[i];
    }
    if (FORM == 1) dzp[IA] = 0.0;     // the angle row has no coupling to the previous angle
    double du = k0[0] + k0[1] * dth + k0[2] * dnu3;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      xi[i] = dzp[i] - cc[i] + hT * F[i] * dth;
      du -= ka[i] * xi[i];
    }
    if (FORM == 1) xi[IA] += 0.5 * d.aub * du; else xi[IW] += be * du;
    solveA<FORM>(G, E, cs, xi, dz);
    dz[7] = du;
    sp[offS] = hot.pick(dz);
    cpy<7>(dzp, dz);
    if (SCHEME == 1) cpy<8>(Gp, G);
  };
#define LD_(k_, buf_) loadV(k_, buf_)
  ASC_SWEEP_FORWARD8U(InV, LD_, body)
#undef LD_
}

template <int SCHEME, int FORM>
__global__ __launch_bounds__(WIDE_THREADS) void q_adjoint_wide(const ascent_params *params, long batch, Geo g,
                                                               double *ws) {
  const int grp = threadIdx.x >> 4, role = threadIdx.x & 15;
  const long p = (long)blockIdx.x * WIDE_NLP_PER_BLOCK + grp;
  const unsigned L = (unsigned)(p & (WAVE - 1));
  const QTile t_((gdbl *)ws + (size_t)(blockIdx.x >> 2) * g.tile_doubles(), L);
  gdbl *sc = scal_base(t_, g);
  if (p >= batch || (int)SC(X_STATE) != ST_FACTORED) return;
  const Der d = derive(params[p]);
  const int K = g.K;
  const unsigned oc = L + (unsigned)((int)SC(X_CUR)) * (21 * WAVE);
  const Scal s = load_scal(t_, sc, X_S);
  const double mu = SC(X_MU), dth = SC(X_DTH), dnu3 = SC(X_DNU3);
  const double sig1 = SC(X_SIG1), sig2 = SC(X_SIG2), rs1 = SC(X_RS1), rs2 = SC(X_RS2);
  const double hT = (1.0 / K) * d.T, dt = hT * s.th, cs = SCHEME == 1 ? 0.5 * dt : dt;
  const double tau = fmax(0.99, 1.0 - mu);
  const int rowA = role < 8 ? Q_G + role : role < 12 ? Q_E + role - 8 : Q_G;
  const int rowB = role < 7 ? Q_R + role : role < 14 ? Q_C + role - 7 : Q_R;
  const unsigned offA = (unsigned)rowA * WAVE + L, offB = (unsigned)rowB * WAVE + L;
  const unsigned offS = (unsigned)(Q_ST + O_L + (role < 7 ? role : 0)) * WAVE + L;
  const OneHot<7> hot(role);
  double dln[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) dln[i] = 0.0;
  double cl = 0.0;
  auto loadV = [&](int k, InV &in) __attribute__((always_inline)) {
    const gdbl *sp = t_.st(k);
    in.gA = sp[offA]; in.gB = sp[offB];
  };
  auto body = [&](InV &in, int k) __attribute__((always_inline)) {
    gdbl *sp = t_.st(k);
    const double G[8] = {bcast16<0>(in.gA), bcast16<1>(in.gA), bcast16<2>(in.gA), bcast16<3>(in.gA),
                         bcast16<4>(in.gA), bcast16<5>(in.gA), bcast16<6>(in.gA), bcast16<7>(in.gA)};
    const double E[4] = {bcast16<8>(in.gA), bcast16<9>(in.gA), bcast16<10>(in.gA), bcast16<11>(in.gA)};
    const double rr[7] = {bcast16<0>(in.gB), bcast16<1>(in.gB), bcast16<2>(in.gB), bcast16<3>(in.gB),
                          bcast16<4>(in.gB), bcast16<5>(in.gB), bcast16<6>(in.gB)};
    const double cc[7] = {bcast16<7>(in.gB), bcast16<8>(in.gB), bcast16<9>(in.gB), bcast16<10>(in.gB),
                          bcast16<11>(in.gB), bcast16<12>(in.gB), bcast16<13>(in.gB)};
    __builtin_amdgcn_sched_barrier(0);   // all broadcasts first (see q_forward_wide)
    double r[7], dl[7];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) r[i] = (FORM == 1 && i == IA) ? rr[i] : rr[i] + dln[i];
    if (SCHEME == 1) {           // Abar_{k+1}' dlambda_{k+1} = dlambda_{k+1} + cs * F_z(z_k)' dlambda_{k+1}
      double t2_[7];
      fzt_lambda(G, dln, t2_);
      ASC_UNROLL
      for (int i = 0; i < 7; i++) r[i] += cs * t2_[i];
    }
    solveAT<FORM>(G, E, cs, r, dl);
    sp[offS] = hot.pick(dl);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) cl += cc[i] * dl[i];
    cpy<7>(dln, dl);
  };
#define LD_(k_, buf_) loadV(k_, buf_)
  ASC_SWEEP_BACKWARD8U(InV, LD_, body)
#undef LD_
  if (role != 0) return;
  double rmax = 0.0, gsum = 0.0, adu = 1.0;
  for (int c = 0; c < g.nch; c++) {
    const gdbl *pp = part_base(t_, g, c);
    rmax = fmax(rmax, ROW(pp, 0)); gsum += ROW(pp, 1); adu = fmin(adu, ROW(pp, 2)); cl += ROW(pp, 3);
  }
  double zK[7], dzK[7];
  ldo<7>(t_.st(K - 1), Q_IT + O_Z, oc, zK);
  ldn<7>(t_, t_.st(K - 1), Q_ST + O_Z, dzK);
  const Terminal tm = terminal_eval(d, zK);
  Scal ds;
  ds.th = dth; ds.nu3 = dnu3;
  ds.s1 = (tm.g1 - s.s1) + tm.g1g[0] * dzK[IX] + tm.g1g[1] * dzK[IY];
  ds.s2 = (tm.g2 - s.s2) + tm.g2g[0] * dzK[IVX] + tm.g2g[1] * dzK[IVY];
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
  double gd = mu * gsum;
  gd += ds.th * (1.0 - mu / dl_ + mu / dU) - mu * ds.s1 / s.s1 - mu * ds.s2 / s.s2;
  cl += tm.e3 * (s.nu3 + ds.nu3) + (tm.g1 - s.s1) * (s.nu1 + ds.nu1) + (tm.g2 - s.s2) * (s.nu2 + ds.nu2);
  const double c1 = SC(X_C1), slog = SC(X_SL);
  double nu_pen = SC(X_NUP);
  const double curv = -gd + cl;
  if (c1 > 0.0) {
    const double need = (gd + 0.5 * fmax(curv, 0.0)) / (0.9 * c1);
    if (nu_pen < need) nu_pen = need + 1.0;
  }
  store_scal(t_, sc, X_D, ds);
  SC(X_NUP) = nu_pen;
  SC(X_DM) = gd - nu_pen * c1;
  SC(X_PHI0) = s.th - mu * slog + nu_pen * c1;
  SC(X_ALPHA) = apr; SC(X_ADU) = adu; SC(X_LS) = 0.0;
  SC(X_STATE) = ST_TRIAL;
}

// ==============================================================================================
// q_finish: results from each lane's current iterate buffer
// ==============================================================================================
__global__ __launch_bounds__(WAVE) void q_finish(const ascent_params *params, long batch, Geo g, double *ws,
                                                 double *traj, double *tf_out, int *status_out,
                                                 int *iters_out, double *blob, int rounds_instead_of_iters) {
  const long p = (long)blockIdx.x * WAVE + threadIdx.x;
  if (p >= batch) return;
  const QTile t_((gdbl *)ws + (size_t)blockIdx.x * g.tile_doubles());
  gdbl *sc = scal_base(t_, g);
  const Der d = derive(params[p]);
  const int K = g.K, nt = K + 1, chunk = blockIdx.y;
  const int k_lo = chunk * CHUNK, k_hi = min(K, k_lo + CHUNK) - 1;     // steps of this wavefront (node = step + 1)
  const unsigned oc = buf_off(t_, (int)SC(X_CUR));
  if (chunk == 0) {      // scalars, and node 0 (the fixed initial state)
    const Scal s = load_scal(t_, sc, X_S);
    tf_out[p] = s.th;
    status_out[p] = (int)SC(X_STATUS);
    iters_out[p] = rounds_instead_of_iters ? (int)SC(X_ROUNDS) : (int)SC(X_ITERS);
    if (blob) {
      double *bs = blob + (21L * K) * batch + p;
      bs[S_TH * batch] = s.th; bs[S_ZLT * batch] = s.zlt; bs[S_ZUT * batch] = s.zut; bs[S_S1 * batch] = s.s1;
      bs[S_S2 * batch] = s.s2; bs[S_ZS1 * batch] = s.zs1; bs[S_ZS2 * batch] = s.zs2; bs[S_NU3 * batch] = s.nu3;
      bs[S_NU1 * batch] = s.nu1; bs[S_NU2 * batch] = s.nu2;
    }
    if (traj) {
      double ax, ay;
      accel<0>(d, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, ax, ay, nullptr, nullptr);
      const double v[10] = {0.0, 0.0, 0.0, 0.0, ax, ay, 0.0, 0.0, 0.0, 0.0};
      ASC_UNROLL
      for (int f = 0; f < 10; f++) traj[((long)f * nt) * batch + p] = v[f];
    }
  }
  for (int kk = k_lo; kk <= k_hi; kk++) {
    const gdbl *sp = t_.st(kk);
    double z[7], ax, ay;
    ldo<7>(sp, Q_IT + O_Z, oc, z);
    const double u = ROWO(sp, Q_IT + O_U, oc);
    if (blob) {
      ASC_UNROLL
      for (int i = 0; i < 7; i++) {
        blob[(7L * kk + i) * batch + p] = z[i];
        blob[(8L * K + 7L * kk + i) * batch + p] = ROWO(sp, Q_IT + O_L + i, oc);
      }
      blob[(7L * K + kk) * batch + p] = u;
      ASC_UNROLL
      for (int b = 0; b < 6; b++) blob[(15L * K + 6L * kk + b) * batch + p] = ROWO(sp, Q_IT + O_ZB + b, oc);
    }
    if (traj) {
      accel<0>(d, z[IX], z[IY], z[IA], z[IM], 0.0, 0.0, ax, ay, nullptr, nullptr);
      const double v[10] = {z[IX], z[IY], z[IVX], z[IVY], ax, ay, z[IA], z[IW], u, z[IM]};
      ASC_UNROLL
      for (int f = 0; f < 10; f++) traj[((long)f * nt + kk + 1) * batch + p] = v[f];
    }
  }
}


// ==============================================================================================
// q_probe_out: what one round of the pipeline left behind, in the external layouts (parity surfaces
// ascent_kkt_step / ascent_eval_nodes of include/ascent.h).  One wavefront = 64 NLPs x CHUNK steps.
// ==============================================================================================
__global__ __launch_bounds__(WAVE) void q_probe_out(const ascent_params *params, long batch, Geo g, double *ws,
                                                    double *step, int *inertia, double *defects, double *jac,
                                                    double *hess) {
  const long p = (long)blockIdx.x * WAVE + threadIdx.x;
  if (p >= batch) return;
  const QTile t_((gdbl *)ws + (size_t)blockIdx.x * g.tile_doubles());
  gdbl *sc = scal_base(t_, g);
  const Der d = derive(params[p]);
  const int K = g.K, chunk = blockIdx.y;
  const int k_lo = chunk * CHUNK, k_hi = min(K, k_lo + CHUNK) - 1;
  const unsigned oc = buf_off(t_, (int)SC(X_CUR));
  if (chunk == 0) {
    if (inertia) inertia[p] = (int)SC(X_STATE) == ST_FACTOR ? 1 : 0;     // a lane whose factorisation was refused is still waiting for one
    if (step)
      for (int r = 0; r < NSC; r++) step[(21L * K + r) * batch + p] = SC(X_D + r);
  }
  for (int k = k_lo; k <= k_hi; k++) {
    const gdbl *sp = t_.st(k);
    if (step) {
      ASC_UNROLL
      for (int i = 0; i < 7; i++) {
        step[(7L * k + i) * batch + p] = ROW(sp, Q_ST + O_Z + i);
        step[(8L * K + 7L * k + i) * batch + p] = ROW(sp, Q_ST + O_L + i);
      }
      step[(7L * K + k) * batch + p] = ROW(sp, Q_ST + O_U);
      ASC_UNROLL
      for (int b = 0; b < 6; b++) step[(15L * K + 6L * k + b) * batch + p] = ROW(sp, Q_ST + O_ZB + b);
    }
    if (defects) {
      ASC_UNROLL
      for (int i = 0; i < 7; i++) defects[(7L * k + i) * batch + p] = ROW(sp, Q_C + i);
      ASC_UNROLL
      for (int i = 0; i < 8; i++) jac[(8L * k + i) * batch + p] = ROW(sp, Q_G + i);
      // the stored Hessian block carries the bound-barrier curvature of angle and mass (see q_trial_eval): take it out again
      const double a = ROWO(sp, Q_IT + O_Z + IA, oc), m = ROWO(sp, Q_IT + O_Z + IM, oc);
      double zb[4];
      ldo<4>(sp, Q_IT + O_ZB, oc, zb);
      const double sa = zb[0] * rcp(a) + zb[1] * rcp(d.aub - a), sm = zb[2] * rcp(m) + zb[3] * rcp(1.0 - m);
      ASC_UNROLL
      for (int i = 0; i < 10; i++)
        hess[(10L * k + i) * batch + p] = ROW(sp, Q_H + i) - (i == 7 ? sa : i == 9 ? sm : 0.0);
    }
  }
}

}  // namespace

// ==============================================================================================
// host driver
// ==============================================================================================
#ifdef DF_STAMPS
extern "C" int ascent_debug_df_stamps(unsigned long long *out6, int reset) {
  unsigned long long z[6] = {0};
  if (hipMemcpyFromSymbol(out6, HIP_SYMBOL(g_df_stamps), sizeof z) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_df_stamps), z, sizeof z) != hipSuccess) return -1;
  return 0;
}
#endif

namespace ascent {

size_t pipeline_ws_bytes(int K, long batch) {
  Geo g{K, (K + CHUNK - 1) / CHUNK, 0, 0};
  const size_t tiles = (size_t)((batch + WAVE - 1) / WAVE);
  return tiles * g.tile_doubles() * sizeof(double) + 64;   // + counters
}

#define ASC_LAUNCH(KERNEL, GRID, ...)                                                                               \
  do {                                                                                                              \
    if (form == 1) hipLaunchKernelGGL((KERNEL<0, 1>), GRID, dim3(WAVE), 0, stream, __VA_ARGS__);                    \
    else if (scheme == 1) hipLaunchKernelGGL((KERNEL<1, 0>), GRID, dim3(WAVE), 0, stream, __VA_ARGS__);            \
    else hipLaunchKernelGGL((KERNEL<0, 0>), GRID, dim3(WAVE), 0, stream, __VA_ARGS__);                              \
  } while (0)
#define ASC_LAUNCH_WIDE(KERNEL, ...)                                                                                \
  do {                                                                                                              \
    if (form == 1) hipLaunchKernelGGL((KERNEL<0, 1>), wgrid, dim3(WIDE_THREADS), 0, stream, __VA_ARGS__);           \
    else if (scheme == 1) hipLaunchKernelGGL((KERNEL<1, 0>), wgrid, dim3(WIDE_THREADS), 0, stream, __VA_ARGS__);   \
    else hipLaunchKernelGGL((KERNEL<0, 0>), wgrid, dim3(WIDE_THREADS), 0, stream, __VA_ARGS__);                     \
  } while (0)
#define PCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { snprintf(err, errlen, "%s: %s", #call, hipGetErrorString(e_)); return ASCENT_E_HIP; } } while (0)

int pipeline_run(const ascent_params *dp, long batch, int K, int scheme, int form, double *ws, const double *dguess, int warm,
                 int max_iter, double tol, double mu0, double *dtraj, double *dtf, int *dstatus, int *diters,
                 double *dblob, hipStream_t stream, PipelineStats *stats, char *err, size_t errlen) {
  Geo g{K, (K + CHUNK - 1) / CHUNK, form, 0};
  const unsigned tiles = (unsigned)((batch + WAVE - 1) / WAVE);
  int *counters = (int *)((char *)ws + (size_t)tiles * g.tile_doubles() * sizeof(double));
  // pinned mirror of the device counters (a copy into pageable memory stalls the stream for ~20 us per burst)
  static int *host_cnt_dev[64] = {nullptr};      // per device: calls on one device are serialised by the caller's mutex
  int dev_ = 0;
  PCHK(hipGetDevice(&dev_));
  dev_ &= 63;
  if (!host_cnt_dev[dev_]) PCHK(hipHostMalloc((void **)&host_cnt_dev[dev_], 4 * sizeof(int)));
  int *host_cnt = host_cnt_dev[dev_];
  const bool debug = getenv("ASCENT_DEBUG") != nullptr;
  int launches = 0;
  hipLaunchKernelGGL(q_init, dim3(tiles, g.nch), dim3(WAVE), 0, stream, dp, batch, g, ws, dguess, warm, mu0,
                     (const double *)nullptr, (const double *)nullptr);
  PCHK(hipGetLastError());
  // Each round advances every lane by one stage of its own state machine: a lane in its normal flow
  // completes one interior-point iteration per round; a rejected line-search trial or a wrong-inertia
  // factorisation costs that lane (only) one more round.  Every kernel is guarded by the lanes' states, so a
  // round enqueued for lanes that turn out to have nothing to do is harmless: the host therefore enqueues
  // `burst` rounds back to back and reads the counters of the last one only then -- the device never waits
  // for the host inside a burst.  A lane needs at most max_iter+1 accepted trial points plus a bounded number
  // of rejected trials and refactorisations per iteration, so the loop terminates.
  // the 16-lanes-per-NLP factorisation pays while the chip has idle SIMDs (see q_factor_wide)
  bool wide = batch <= 4096;     // one wavefront per SIMD; beyond that the one-lane sweeps win (scripts/batch_sweep.py)
  if (const char *e = getenv("ASCENT_FACTOR")) wide = e[0] == 'w';
  int burst = 4;
  if (const char *e = getenv("ASCENT_ROUNDS_PER_SYNC")) { const int v = atoi(e); if (v >= 1 && v <= 64) burst = v; }
  for (long round = 0;;) {
    if (round > 100L * (max_iter + 2)) { snprintf(err, errlen, "pipeline did not terminate within %ld rounds (solver condition, not a HIP error)", round); return ASCENT_E_NOTERM; }
    for (int r = 0; r < burst; r++, round++) {
      ASC_LAUNCH(q_trial_eval, dim3(tiles, g.nch), dp, batch, g, ws, counters);
      if (!wide) ASC_LAUNCH(q_decide_factor, dim3(tiles), dp, batch, g, ws, max_iter, tol, counters);
      const dim3 wgrid((unsigned)((batch + WIDE_NLP_PER_BLOCK - 1) / WIDE_NLP_PER_BLOCK));
      if (wide) ASC_LAUNCH_WIDE(q_factor_wide, dp, batch, g, ws, max_iter, tol, counters);
      if (r == burst - 1) PCHK(hipMemcpyAsync(host_cnt, counters, 3 * sizeof(int), hipMemcpyDeviceToHost, stream));
      if (!wide) ASC_LAUNCH(q_forward, dim3(tiles), dp, batch, g, ws);
      else ASC_LAUNCH_WIDE(q_forward_wide, dp, batch, g, ws);
      hipLaunchKernelGGL(q_local, dim3(tiles, g.nch), dim3(WAVE), 0, stream, dp, batch, g, ws);
      if (!wide) ASC_LAUNCH(q_adjoint, dim3(tiles), dp, batch, g, ws);
      else ASC_LAUNCH_WIDE(q_adjoint_wide, dp, batch, g, ws);
      launches += 5;
      PCHK(hipGetLastError());      // a refused launch must not hide behind the rest of the burst
    }
    PCHK(hipStreamSynchronize(stream));
    const int n_pending = host_cnt[0], n_factored = host_cnt[1];   // retrial/refactor lanes; lanes with a step to take
    if (debug) fprintf(stderr, "[ascent pipeline] after round %ld: pending %d (refactor %d), stepping %d\n", round - 1, n_pending, host_cnt[2], n_factored);
    if (n_pending == 0 && n_factored == 0) break;
  }
  hipLaunchKernelGGL(q_finish, dim3(tiles, g.nch), dim3(WAVE), 0, stream, dp, batch, g, ws, dtraj, dtf, dstatus, diters,
                     dblob, getenv("ASCENT_DEBUG_ROUNDS") != nullptr);
  PCHK(hipGetLastError());
  if (stats) stats->launches = launches + 2;
  return ASCENT_OK;
}


// One round of the pipeline at a caller-supplied iterate (parity surface): q_init takes the iterate as it is, then exactly
// the kernels a solve launches per round -- q_trial_eval, the factorisation (one-lane or 16-lane), forward, q_local,
// adjoint -- with mu and delta_w per problem from the caller; q_probe_out hands back the Newton step and/or the node rows.
int pipeline_probe(const ascent_params *dp, long batch, int K, int scheme, int form, double *ws, const double *diterate,
                   const double *dmu, const double *ddw, bool wide, bool step_too, double *dstep, int *dinertia,
                   double *ddefects, double *djac, double *dhess, hipStream_t stream, char *err, size_t errlen) {
  Geo g{K, (K + CHUNK - 1) / CHUNK, form, 1};
  const unsigned tiles = (unsigned)((batch + WAVE - 1) / WAVE);
  int *counters = (int *)((char *)ws + (size_t)tiles * g.tile_doubles() * sizeof(double));
  hipLaunchKernelGGL(q_init, dim3(tiles, g.nch), dim3(WAVE), 0, stream, dp, batch, g, ws, diterate, 2, 0.1, dmu, ddw);
  PCHK(hipGetLastError());
  ASC_LAUNCH(q_trial_eval, dim3(tiles, g.nch), dp, batch, g, ws, counters);
  PCHK(hipGetLastError());
  if (step_too) {
    const dim3 wgrid((unsigned)((batch + WIDE_NLP_PER_BLOCK - 1) / WIDE_NLP_PER_BLOCK));
    if (wide) {
      ASC_LAUNCH_WIDE(q_factor_wide, dp, batch, g, ws, 1 << 30, -1.0, counters);
      ASC_LAUNCH_WIDE(q_forward_wide, dp, batch, g, ws);
    } else {
      ASC_LAUNCH(q_decide_factor, dim3(tiles), dp, batch, g, ws, 1 << 30, -1.0, counters);
      ASC_LAUNCH(q_forward, dim3(tiles), dp, batch, g, ws);
    }
    PCHK(hipGetLastError());
    hipLaunchKernelGGL(q_local, dim3(tiles, g.nch), dim3(WAVE), 0, stream, dp, batch, g, ws);
    if (wide) ASC_LAUNCH_WIDE(q_adjoint_wide, dp, batch, g, ws);
    else ASC_LAUNCH(q_adjoint, dim3(tiles), dp, batch, g, ws);
    PCHK(hipGetLastError());
  }
  hipLaunchKernelGGL(q_probe_out, dim3(tiles, g.nch), dim3(WAVE), 0, stream, dp, batch, g, ws, step_too ? dstep : nullptr,
                     step_too ? dinertia : nullptr, ddefects, djac, dhess);
  PCHK(hipGetLastError());
  return ASCENT_OK;
}

}  // namespace ascent
