// Dense-block solver path of the batched ascent NLP solver (gfx950): one WAVEFRONT per NLP.
//
// The hand-tuned paths (ascent_solver.hip, ascent_pipeline.hip) exploit the sparsity of the backward-Euler /
// trapezoid step Jacobian A = I - dt*df/dz (a closed-form inverse through one 2x2 block).  Hermite-Simpson
// collocation (ascent_opts.scheme = 2; the report of the reference cites Kelly's tutorial as its method source, PDF
// p3/p25) evaluates the dynamics of /root/reference/Launch_Optimiser.py:114-136 at the interior point
//     z_m = (z_{k-1}+z_k)/2 + dt/8 [f(z_{k-1},u_k) - f(z_k,u_k)]
// of every step, so both step Jacobians  d c_k/d z_{k-1},  d c_k/d z_k  and the three Hessian blocks of lambda_k'c_k
// (incl. the cross block between the two nodes) are dense 7x7.  This path therefore works on dense 8x8 blocks (7
// states + one padding slot), for all three schemes:
//
//   d_eval     lane = (NLP, step), consecutive lanes = consecutive steps: trial point, step defect, dense Jacobian
//              and Hessian blocks (chain rule by hand through the three evaluation points), merit / KKT-error pieces.
//   d_newton   one wavefront = one NLP, lane (i,j) = element (i,j) of every 8x8 block: line-search / convergence /
//              barrier decisions, then the bordered block-tridiagonal KKT solve as a stage-wise Riccati recursion
//              on dense blocks -- 8x8 products through LDS broadcasts, Gauss-Jordan inverse of the step Jacobian in
//              LDS, the two border unknowns (the free final time theta that multiplies every step, LO:114-123, and
//              the multiplier of the terminal r.v = 0, LO:173) carried as extra columns -- the forward substitution
//              with the multiplier step computed on the way (no adjoint sweep), the bound-multiplier steps and both
//              fraction-to-boundary rules.  A factorisation with the wrong inertia is repeated with a larger
//              primal regularisation inside the kernel: the wavefront owns its NLP, nobody else waits.
//
// MFMA: v_mfma_f64_4x4x4 would tile these 8x8 products, but the FP64 matrix rate of MI355X equals its FP64 vector
// rate and the products are latency-, not throughput-bound at one wavefront per NLP: plain FMA.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "ascent.h"
#include "ascent_device.hpp"
#include "ascent_tile.hpp"
#include "ascent_dense.hpp"
#include "ascent_blocktri.hpp"

using namespace ascent;

namespace {

// ---- per-NLP workspace (doubles), all node arrays [row][K] with the step index contiguous ----------------------
constexpr int NIT = 26;                       // rows of an iterate / step: z[7] u lambda[7] zb[6] | move penalty: lambda_u p n z_p z_n
constexpr int O_Z = 0, O_U = 7, O_L = 8, O_ZB = 15;
constexpr int O_LU = 21, O_PP = 22, O_PN = 23, O_ZP = 24, O_ZN = 25;    // (used with DGeo::dc only, see "move penalty" below)
constexpr int NV_GA = 0, NV_GB = 8, NV_P = 16, NNV = 27;   // node vectors: Ja'lambda[8], Jb'lambda[8], partials[11]
// partials written by d_eval: 0 c1 (sum |c|)  1 cinf  2 sum of logs  3 pmin  4 pmax  5 zsum  6 l1  7 J_theta'lambda
//                             8 H_u,theta  9 H_theta,theta  10 dcost (p + n)
//
// Move penalty (ascent_opts.move_penalty = 1; the reference's angledoubledot.DCOST = 1e-5, LO:99): the objective gains
// dcost * sum_k |u_k - u_{k-1}|, u_{-1} = 0, as an l1 term with a slack pair per step: u_k - u_{k-1} = p_k - n_k, p, n >= 0,
// cost dcost (p_k + n_k).  The control of step k then couples to the control of step k-1: u_k becomes the EIGHTH STATE of
// the stage (the padding slot of the 8x8 blocks: row 7 of the step defect is the movement equation, Ja[7][7] = -1,
// Jb[7][7] = 1, column 7 of Jb is the old control column, the control's bound terms and its cross derivative with theta move
// into the state blocks), and the stage's scalar control is delta_k = p_k - n_k, whose two bounded slacks reduce to one
// pivot: with Sigma_p = z_p/p, Sigma_n = z_n/n and the stationarity residuals r_p = dcost - mu/p - lambda_u,
// r_n = dcost - mu/n + lambda_u the pair behaves like a control with curvature R = 1/(1/Sigma_p + 1/Sigma_n), gradient
// g = R (r_p/Sigma_p - r_n/Sigma_n) and control column -e_7.  The Riccati recursion itself does not change.
constexpr int G_JA = 0, G_JB = 1, G_HAA = 2, G_HAB = 3, G_HBB = 4, G_V = 5, NGRID = 6;    // stage record: 8x8 grids
// rows of the vector grid G_V:  0 c   1 Ju   2 Jtheta   3 Ha,theta   4 Hb,theta
constexpr int F_EA = 0, F_LA = 1, F_EL2 = 2, F_GAIN = 3;    // forward record: three 8x8 grids + 16 gains
constexpr int FWD_DOUBLES = 3 * 64 + 16;
enum {  // scalar record
  X_STATE, X_ITERS, X_STATUS, X_CUR, X_FIRST, X_LS, X_MU, X_NUP, X_DWL, X_ALPHA, X_ADU, X_PHI0, X_DM, X_C1, X_SL,
  X_S,                       // 10 scalars of the iterate: th zlt zut s1 s2 zs1 zs2 nu3 nu1 nu2
  X_D = X_S + 10,            // 10 scalars of the step
  X_T = X_D + 10,            // 10 trial scalars
  X_REFAC = X_T + 10, X_RTH, X_DW, X_MV,
  NSCAL = 64
};
enum { ST_TRIAL = 0, ST_NEWTON = 1, ST_DONE = 3 };     // ST_NEWTON: waiting for the PCR solve of its Newton system

struct DGeo {
  int K, scheme, terminal, dc;       // dc: the l1 move penalty is on
  __host__ __device__ size_t nlp_doubles() const {
    return (size_t)K * (3 * NIT + NNV + NGRID * 64 + FWD_DOUBLES) + NSCAL;
  }
  __host__ __device__ size_t off_it(int buf) const { return (size_t)buf * NIT * K; }
  __host__ __device__ size_t off_st() const { return (size_t)2 * NIT * K; }
  __host__ __device__ size_t off_nv() const { return (size_t)3 * NIT * K; }
  __host__ __device__ size_t off_rec() const { return (size_t)(3 * NIT + NNV) * K; }
  __host__ __device__ size_t off_fwd() const { return off_rec() + (size_t)K * NGRID * 64; }
  __host__ __device__ size_t off_sc() const { return off_fwd() + (size_t)K * FWD_DOUBLES; }
};

ASC_DEV Scal load_scal(const double *sc, int r0) {
  Scal s;
  s.th = sc[r0 + S_TH]; s.zlt = sc[r0 + S_ZLT]; s.zut = sc[r0 + S_ZUT]; s.s1 = sc[r0 + S_S1]; s.s2 = sc[r0 + S_S2];
  s.zs1 = sc[r0 + S_ZS1]; s.zs2 = sc[r0 + S_ZS2]; s.nu3 = sc[r0 + S_NU3]; s.nu1 = sc[r0 + S_NU1]; s.nu2 = sc[r0 + S_NU2];
  return s;
}
ASC_DEV void store_scal(double *sc, int r0, const Scal &s) {
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

// ==============================================================================================================
// d_init: external blob / cold start -> iterate buffer 0, zero step, scalar record.  Lane = (NLP, step).
// ==============================================================================================================
__global__ __launch_bounds__(WAVE) void d_init(const ascent_params *params, long batch, DGeo g, double *ws,
                                               const double *guess, int warm, double mu_init, int probe,
                                               const double *probe_mu) {
  const long p = blockIdx.y;
  const int k = blockIdx.x * WAVE + threadIdx.x, K = g.K;
  if (k >= K) return;
  double *w = ws + (size_t)p * g.nlp_doubles();
  const Der d = derive_t(params[p], g.terminal);
  const int asked_warm = warm;
  if (warm && !(guess[(21L * K + S_TH) * batch + p] > 0.0)) warm = 0;   // "no guess for this problem" (nested iteration)
  const double tf0 = 0.9, dr = 0.166, aend = 0.5, vp = sqrt(d.vp2), dt0 = (1.0 / K) * d.T * tf0;
  const double sdr = sin(dr), cdr = cos(dr);
  const double xf = -d.rhof * sdr, yf = d.rhof * cdr - d.rho0;
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
  }
  if (!probe) {
    z[IA] = push_in(z[IA], 0.0, d.aub);
    z[IM] = push_in(z[IM], 0.0, 1.0);
    u = push_in(u, -1.0, 1.0);
  }
  ASC_UNROLL
  for (int b = 0; b < 6; b++) zb[b] = warm == 2 ? (probe ? zb[b] : fmax(zb[b], 1e-12)) : 1.0;
  ASC_UNROLL
  for (int i = 0; i < 7; i++) l[i] = warm == 2 ? l[i] : 0.0;
  double *it = w + g.off_it(0), *st = w + g.off_st();
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { it[(O_Z + i) * K + k] = z[i]; it[(O_L + i) * K + k] = l[i]; }
  it[O_U * K + k] = u;
  ASC_UNROLL
  for (int b = 0; b < 6; b++) it[(O_ZB + b) * K + k] = zb[b];
  for (int r = 0; r < NIT; r++) st[r * K + k] = 0.0;
  if (g.dc) {      // slacks of the movement equation around the guess's own movement; multipliers that zero their stationarity rows
    double up = (k > 0 && warm) ? guess[(7L * K + k - 1) * batch + p] : 0.0;
    if (k > 0 && !probe) up = push_in(up, -1.0, 1.0);
    const double dl = u - up, eps = warm ? 1e-4 : 1e-2, dc = params[p].dcost;
    it[O_LU * K + k] = 0.0;
    it[O_PP * K + k] = fmax(dl, 0.0) + eps; it[O_PN * K + k] = fmax(-dl, 0.0) + eps;
    it[O_ZP * K + k] = dc; it[O_ZN * K + k] = dc;
  }
  if (k != K - 1) return;
  double *sc = w + g.off_sc();
  Scal s;
  if (warm) {
    const double *gs = guess + (21L * K) * batch + p;
    s.th = gs[S_TH * batch]; s.zlt = gs[S_ZLT * batch]; s.zut = gs[S_ZUT * batch]; s.s1 = gs[S_S1 * batch];
    s.s2 = gs[S_S2 * batch]; s.zs1 = gs[S_ZS1 * batch]; s.zs2 = gs[S_ZS2 * batch]; s.nu3 = gs[S_NU3 * batch];
    s.nu1 = gs[S_NU1 * batch]; s.nu2 = gs[S_NU2 * batch];
  } else {
    s.th = tf0;
  }
  if (!probe) s.th = push_in(s.th, d.tlb, d.tub);
  const Terminal tm = terminal_of(d, z);
  if (probe) {
  } else if (warm != 2) {
    s.s1 = fmax(tm.g1, 1e-2); s.s2 = fmax(tm.g2, 1e-2);
    s.zlt = s.zut = s.zs1 = s.zs2 = 1.0;
    s.nu3 = s.nu1 = s.nu2 = 0.0;
  } else {
    s.s1 = fmax(s.s1, 1e-10); s.s2 = fmax(s.s2, 1e-10);
    s.zlt = fmax(s.zlt, 1e-12); s.zut = fmax(s.zut, 1e-12);
    s.zs1 = fmax(s.zs1, 1e-12); s.zs2 = fmax(s.zs2, 1e-12);
  }
  for (int r = 0; r < NSCAL; r++) sc[r] = 0.0;
  store_scal(sc, X_S, s);
  sc[X_STATE] = ST_TRIAL; sc[X_FIRST] = 1.0; sc[X_STATUS] = ASCENT_MAX_ITER;
  sc[X_MU] = probe ? probe_mu[p] : ((asked_warm && !warm) ? 0.1 : mu_init);
  sc[X_NUP] = 1.0;
}

// ==============================================================================================================
// d_eval: lane = (NLP, step).  Trial point, step defect, dense Jacobian / Hessian blocks, partials.
// ==============================================================================================================
// With step weights (wa, wm, wb) on f(z_{k-1}), f(z_m), f(z_k):  BE (0,0,1), trapezoid (1/2,0,1/2), HS (1/6,4/6,1/6):
//   c  = zb - za - dt (wa fa + wm fm + wb fb),          zm = (za+zb)/2 + (dt/8)(fa - fb)
//   Ma = dzm/dza = I/2 + e Fa,  Mb = dzm/dzb = I/2 - e Fb,  m_th = dzm/dtheta = (hT/8)(fa - fb),   e = dt/8, F. = df/dz
//   Ja = -I - sa Fa - sm Fm Ma        Jb = I - sb Fb - sm Fm Mb        Ju = -dt f_u        Jth = -hT(sum w f) - sm Fm m_th
// and, with l(z) = lambda'f(z,u), gm = Fm'lambda, H[q;w] = Hessian of w.(ax,ay) at q (linear in w):
//   Haa = -H[qa; sa lam + sm e gm] - sm Ma'Hm Ma     Hbb = -H[qb; sb lam - sm e gm] - sm Mb'Hm Mb     Hab = -sm Ma'Hm Mb
//   Ha,th = -hT(wa Fa'lam + wm Ma'gm) - sm[(hT/8) Fa'gm + Ma'Hm m_th]      Hb,th likewise with -(hT/8) Fb'gm
//   Hth,th = -2 hT wm gm'm_th - sm m_th'Hm m_th          Hu,th = -hT alpha lam_w
// (the control enters f additively and linearly, so every second derivative with respect to u and a state vanishes, and
//  zm does not depend on u: f_u cancels in fa - fb.)
struct QMap { int q; double c; };     // state index -> (index among (x,y,angle,mass), coefficient of M. on that row)

ASC_DEV void h4_full(const double *H10, double H4[4][4]) {   // packed upper triangle xx xy xa xm yy ya ym aa am mm
  H4[0][0] = H10[0]; H4[0][1] = H4[1][0] = H10[1]; H4[0][2] = H4[2][0] = H10[2]; H4[0][3] = H4[3][0] = H10[3];
  H4[1][1] = H10[4]; H4[1][2] = H4[2][1] = H10[5]; H4[1][3] = H4[3][1] = H10[6];
  H4[2][2] = H10[7]; H4[2][3] = H4[3][2] = H10[8]; H4[3][3] = H10[9];
}

__global__ __launch_bounds__(WAVE) void d_eval(const ascent_params *params, long batch, DGeo g, double *ws) {
  const long p = blockIdx.y;
  const int k = blockIdx.x * WAVE + threadIdx.x, K = g.K;
  double *w = ws + (size_t)p * g.nlp_doubles();
  double *sc = w + g.off_sc();
  if ((int)sc[X_STATE] != ST_TRIAL || k >= K) return;
  const Der d = derive_t(params[p], g.terminal);
  const bool first = sc[X_FIRST] != 0.0;
  const double alpha = first ? 0.0 : sc[X_ALPHA], adu = first ? 0.0 : sc[X_ADU], mu = sc[X_MU];
  const Scal s = load_scal(sc, X_S), ds = load_scal(sc, X_D);
  const Scal stt = trial_scal(d, s, ds, alpha, adu, mu, first);
  const int cur = (int)sc[X_CUR];
  const double *ic = w + g.off_it(cur), *st = w + g.off_st();
  double *in = w + g.off_it(1 - cur), *nv = w + g.off_nv(), *rec = w + g.off_rec() + (size_t)k * NGRID * 64;
  const double hT = (1.0 / K) * d.T, dt = hT * stt.th;
  const double wa = g.scheme == 0 ? 0.0 : g.scheme == 1 ? 0.5 : 1.0 / 6.0;
  const double wm = g.scheme == 2 ? 4.0 / 6.0 : 0.0;
  const double wb = g.scheme == 0 ? 1.0 : g.scheme == 1 ? 0.5 : 1.0 / 6.0;
  const double sa = dt * wa, sm = dt * wm, sb = dt * wb, e8 = 0.125 * dt, h8 = 0.125 * hT;
  // ---- trial point ----------------------------------------------------------------------------------------------
  double za[7], zb[7], lam[7], zbd[6], u;
  ASC_UNROLL
  for (int i = 0; i < 7; i++) {
    zb[i] = ic[(O_Z + i) * K + k] + alpha * st[(O_Z + i) * K + k];
    za[i] = k ? ic[(O_Z + i) * K + k - 1] + alpha * st[(O_Z + i) * K + k - 1] : 0.0;
    lam[i] = ic[(O_L + i) * K + k] + alpha * st[(O_L + i) * K + k];
  }
  u = ic[O_U * K + k] + alpha * st[O_U * K + k];
  const double dist[6] = {zb[IA], d.aub - zb[IA], zb[IM], 1.0 - zb[IM], u + 1.0, 1.0 - u};
  const double mlo = mu * 1e-10, mhi = mu * 1e10;
  ASC_UNROLL
  for (int b = 0; b < 6; b++) {
    double v = ic[(O_ZB + b) * K + k];
    if (!first) { const double id = rcp(dist[b]); v = fmin(fmax(v + adu * st[(O_ZB + b) * K + k], mlo * id), mhi * id); }
    zbd[b] = v;
    in[(O_ZB + b) * K + k] = v;
  }
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { in[(O_Z + i) * K + k] = zb[i]; in[(O_L + i) * K + k] = lam[i]; }
  in[O_U * K + k] = u;
  double lu = 0.0, pp = 1.0, pn = 1.0, zp = 0.0, zn = 0.0, cu = 0.0;
  if (g.dc) {
    const double up = k ? ic[O_U * K + k - 1] + alpha * st[O_U * K + k - 1] : 0.0;
    lu = ic[O_LU * K + k] + alpha * st[O_LU * K + k];
    pp = ic[O_PP * K + k] + alpha * st[O_PP * K + k];
    pn = ic[O_PN * K + k] + alpha * st[O_PN * K + k];
    zp = ic[O_ZP * K + k]; zn = ic[O_ZN * K + k];
    if (!first) {
      const double ip = rcp(pp), in_ = rcp(pn);
      zp = fmin(fmax(zp + adu * st[O_ZP * K + k], mlo * ip), mhi * ip);
      zn = fmin(fmax(zn + adu * st[O_ZN * K + k], mlo * in_), mhi * in_);
    }
    in[O_LU * K + k] = lu; in[O_PP * K + k] = pp; in[O_PN * K + k] = pn; in[O_ZP * K + k] = zp; in[O_ZN * K + k] = zn;
    cu = u - up - pp + pn;
  }
  if (k == K - 1) store_scal(sc, X_T, stt);
  // ---- the three evaluation points --------------------------------------------------------------------------------
  double Ga[8], Gb[8], Gm[8], Hw[10], fa[7], fb[7], fm[7], zm[7], ax, ay;
  ASC_UNROLL
  for (int i = 0; i < 8; i++) { Ga[i] = 0.0; Gm[i] = 0.0; }
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { fa[i] = 0.0; fm[i] = 0.0; zm[i] = 0.0; }
  accel<1>(d, zb[IX], zb[IY], zb[IA], zb[IM], 0.0, 0.0, ax, ay, Gb, nullptr);
  rhs_f(d, zb, u, ax, ay, fb);
  if (g.scheme != 0) {
    accel<1>(d, za[IX], za[IY], za[IA], za[IM], 0.0, 0.0, ax, ay, Ga, nullptr);
    rhs_f(d, za, u, ax, ay, fa);
  }
  double H4m[4][4];
  ASC_UNROLL
  for (int i = 0; i < 4; i++) {
    ASC_UNROLL
    for (int j = 0; j < 4; j++) H4m[i][j] = 0.0;
  }
  double gm[7], mth[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { gm[i] = 0.0; mth[i] = 0.0; }
  if (g.scheme == 2) {
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { mth[i] = h8 * (fa[i] - fb[i]); zm[i] = 0.5 * (za[i] + zb[i]) + e8 * (fa[i] - fb[i]); }
    accel<2>(d, zm[IX], zm[IY], zm[IA], zm[IM], lam[IVX], lam[IVY], ax, ay, Gm, Hw);
    rhs_f(d, zm, u, ax, ay, fm);
    h4_full(Hw, H4m);
    fzt_lambda(Gm, lam, gm);
  }
  // Hessians of the weighted accelerations at the end points
  double H4a[4][4], H4b[4][4];
  {
    double t1, t2, Gx[8];
    accel<2>(d, zb[IX], zb[IY], zb[IA], zb[IM], sb * lam[IVX] - sm * e8 * gm[IVX], sb * lam[IVY] - sm * e8 * gm[IVY], t1, t2, Gx, Hw);
    h4_full(Hw, H4b);
    if (g.scheme != 0) {
      accel<2>(d, za[IX], za[IY], za[IA], za[IM], sa * lam[IVX] + sm * e8 * gm[IVX], sa * lam[IVY] + sm * e8 * gm[IVY], t1, t2, Gx, Hw);
      h4_full(Hw, H4a);
    } else {
      ASC_UNROLL
      for (int i = 0; i < 4; i++) {
        ASC_UNROLL
        for (int j = 0; j < 4; j++) H4a[i][j] = 0.0;
      }
    }
  }
  // ---- defect, vectors -------------------------------------------------------------------------------------------
  const QMap qa[7] = {{0, 0.5}, {1, 0.5}, {0, e8}, {1, e8}, {2, 0.5}, {2, e8}, {3, 0.5}};      // rows of Ma
  const QMap qb[7] = {{0, 0.5}, {1, 0.5}, {0, -e8}, {1, -e8}, {2, 0.5}, {2, -e8}, {3, 0.5}};   // rows of Mb
  constexpr int q2s[4] = {IX, IY, IA, IM};
  double c[7], Jth[7], Hath[7], Hbth[7], ga[7], gb[7], t7[7], t7b[7];
  double Hmm[4];                       // Hm m_theta restricted to (x,y,angle,mass)
  {
    const double mq[4] = {mth[IX], mth[IY], mth[IA], mth[IM]};
    ASC_UNROLL
    for (int i = 0; i < 4; i++) Hmm[i] = H4m[i][0] * mq[0] + H4m[i][1] * mq[1] + H4m[i][2] * mq[2] + H4m[i][3] * mq[3];
  }
  fz_mul(Gm, mth, t7);                 // Fm m_theta
  double c1 = 0.0, cinf = 0.0;
  ASC_UNROLL
  for (int i = 0; i < 7; i++) {
    c[i] = zb[i] - za[i] - dt * (wa * fa[i] + wm * fm[i] + wb * fb[i]);
    Jth[i] = -hT * (wa * fa[i] + wm * fm[i] + wb * fb[i]) - sm * t7[i];
    c1 += fabs(c[i]);
    cinf = fmax(cinf, fabs(c[i]));
  }
  double Fal[7], Fbl[7], Fag[7], Fbg[7];
  fzt_lambda(Ga, lam, Fal); fzt_lambda(Gb, lam, Fbl); fzt_lambda(Ga, gm, Fag); fzt_lambda(Gb, gm, Fbg);
  double rth = 0.0, mHm = 0.0, gmm = 0.0;
  ASC_UNROLL
  for (int i = 0; i < 7; i++) {
    const double Mag = 0.5 * gm[i] + e8 * Fag[i], Mbg = 0.5 * gm[i] - e8 * Fbg[i];      // Ma'gm, Mb'gm
    ga[i] = -lam[i] - sa * Fal[i] - sm * Mag;
    gb[i] = lam[i] - sb * Fbl[i] - sm * Mbg;
    const double MaHm = qa[i].c * Hmm[qa[i].q], MbHm = qb[i].c * Hmm[qb[i].q];          // (Ma'Hm m_th)_i
    Hath[i] = -hT * (wa * Fal[i] + wm * Mag) - sm * (h8 * Fag[i] + MaHm);
    Hbth[i] = -hT * (wb * Fbl[i] + wm * Mbg) - sm * (-h8 * Fbg[i] + MbHm);
    rth += Jth[i] * lam[i];
    gmm += gm[i] * mth[i];
  }
  mHm = mth[IX] * Hmm[0] + mth[IY] * Hmm[1] + mth[IA] * Hmm[2] + mth[IM] * Hmm[3];
  (void)t7b;
  const double Hthth = -2.0 * hT * wm * gmm - sm * mHm;
  const double Huth = -hT * d.alpha * lam[IW];
  if (g.dc) { c1 += fabs(cu); cinf = fmax(cinf, fabs(cu)); }
  // ---- stage record: 8x8 grids, element (i,j) at [i*8+j] ------------------------------------------------------------
  // Ja = -I - sa Fa - sm Fm Ma ; Jb = I - sb Fb - sm Fm Mb : row by row (F has three unit entries and the 2x4 block G)
  auto frow = [&](const double *G, int r, double *row) __attribute__((always_inline)) {      // row r of F = df/dz
    ASC_UNROLL
    for (int j = 0; j < 7; j++) row[j] = 0.0;
    if (r == IX) row[IVX] = 1.0;
    if (r == IY) row[IVY] = 1.0;
    if (r == IVX) { row[IX] = G[0]; row[IY] = G[1]; row[IA] = G[2]; row[IM] = G[3]; }
    if (r == IVY) { row[IX] = G[4]; row[IY] = G[5]; row[IA] = G[6]; row[IM] = G[7]; }
    if (r == IA) row[IW] = 1.0;
  };
  ASC_UNROLL
  for (int r = 0; r < 8; r++) {
    double ja[8], jb[8];
    ASC_UNROLL
    for (int j = 0; j < 8; j++) { ja[j] = 0.0; jb[j] = 0.0; }
    if (r < 7) {
      double fra[7], frb[7], frm[7];
      frow(Ga, r, fra); frow(Gb, r, frb); frow(Gm, r, frm);
      // (Fm Ma)[r,:] = sum_q Fm[r,q] Ma[q,:],  Ma[q,:] = e_q/2 + e Fa[q,:]
      double fmma[7], fmmb[7];
      ASC_UNROLL
      for (int j = 0; j < 7; j++) { fmma[j] = 0.5 * frm[j]; fmmb[j] = 0.5 * frm[j]; }
      ASC_UNROLL
      for (int q = 0; q < 7; q++) {
        double fqa[7], fqb[7];
        frow(Ga, q, fqa); frow(Gb, q, fqb);
        ASC_UNROLL
        for (int j = 0; j < 7; j++) { fmma[j] += e8 * frm[q] * fqa[j]; fmmb[j] -= e8 * frm[q] * fqb[j]; }
      }
      ASC_UNROLL
      for (int j = 0; j < 7; j++) {
        ja[j] = (j == r ? -1.0 : 0.0) - sa * fra[j] - sm * fmma[j];
        jb[j] = (j == r ? 1.0 : 0.0) - sb * frb[j] - sm * fmmb[j];
      }
    } else {
      jb[7] = 1.0;          // the padding slot: x8_k = 0; with the move penalty: u_k - u_{k-1} - p_k + n_k
      if (g.dc) ja[7] = -1.0;
    }
    if (g.dc && r == IW) jb[7] = -dt * d.alpha;       // the control is a state: its column of the step Jacobian
    ASC_UNROLL
    for (int j = 0; j < 8; j++) { rec[G_JA * 64 + r * 8 + j] = ja[j]; rec[G_JB * 64 + r * 8 + j] = jb[j]; }
  }
  ASC_UNROLL
  for (int i = 0; i < 8; i++) {
    ASC_UNROLL
    for (int j = 0; j < 8; j++) {
      double haa = 0.0, hab = 0.0, hbb = 0.0;
      if (i < 7 && j < 7) {
        const double hm = H4m[qa[i].q][qa[j].q];
        haa = -sm * qa[i].c * qa[j].c * hm;
        hab = -sm * qa[i].c * qb[j].c * hm;
        hbb = -sm * qb[i].c * qb[j].c * hm;
        // the end-point Hessians live on (x, y, angle, mass) only
        int qi = -1, qj = -1;
        ASC_UNROLL
        for (int q = 0; q < 4; q++) { if (q2s[q] == i) qi = q; if (q2s[q] == j) qj = q; }
        if (qi >= 0 && qj >= 0) { haa -= H4a[qi][qj]; hbb -= H4b[qi][qj]; }
      }
      rec[G_HAA * 64 + i * 8 + j] = haa;
      rec[G_HAB * 64 + i * 8 + j] = hab;
      rec[G_HBB * 64 + i * 8 + j] = hbb;
    }
  }
  ASC_UNROLL
  for (int j = 0; j < 8; j++) {
    rec[G_V * 64 + 0 * 8 + j] = j < 7 ? c[j] : cu;
    rec[G_V * 64 + 1 * 8 + j] = g.dc ? (j == 7 ? -1.0 : 0.0) : (j == IW ? -dt * d.alpha : 0.0);
    rec[G_V * 64 + 2 * 8 + j] = j < 7 ? Jth[j] : 0.0;
    rec[G_V * 64 + 3 * 8 + j] = j < 7 ? Hath[j] : 0.0;
    rec[G_V * 64 + 4 * 8 + j] = j < 7 ? Hbth[j] : (g.dc ? Huth : 0.0);
    rec[G_V * 64 + 5 * 8 + j] = 0.0; rec[G_V * 64 + 6 * 8 + j] = 0.0; rec[G_V * 64 + 7 * 8 + j] = 0.0;
  }
  // ---- node vectors and partials ----------------------------------------------------------------------------------
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { nv[(NV_GA + i) * K + k] = ga[i]; nv[(NV_GB + i) * K + k] = gb[i]; }
  nv[(NV_GA + 7) * K + k] = -lu;                                  // Ja'lambda, Jb'lambda of the control slot
  nv[(NV_GB + 7) * K + k] = lu - dt * d.alpha * lam[IW];
  double pmin = 1e300, pmax = -1e300, zsum = 0.0, l1 = 0.0;
  ASC_UNROLL
  for (int b = 0; b < 6; b++) { const double pr = dist[b] * zbd[b]; pmin = fmin(pmin, pr); pmax = fmax(pmax, pr); zsum += zbd[b]; }
  ASC_UNROLL
  for (int i = 0; i < 7; i++) l1 += fabs(lam[i]);
  const double pa = dist[0] * dist[1], pm = dist[2] * dist[3], pu = dist[4] * dist[5];
  double sl = (pa > 0.0 && pm > 0.0 && pu > 0.0) ? log(pa * pm * pu) : NAN;
  if (g.dc) {
    pmin = fmin(pmin, fmin(pp * zp, pn * zn)); pmax = fmax(pmax, fmax(pp * zp, pn * zn));
    zsum += zp + zn; l1 += fabs(lu);
    sl += (pp > 0.0 && pn > 0.0) ? log(pp * pn) : NAN;
  }
  if (k == K - 1) {
    const Terminal t = terminal_of(d, zb);
    const double e1 = fabs(t.e3), e2 = fabs(t.g1 - stt.s1), e3 = fabs(t.g2 - stt.s2);
    cinf = fmax(cinf, fmax(e1, fmax(e2, e3)));
    c1 += e1 + e2 + e3;
    const double ps = ((stt.th - d.tlb) * (d.tub - stt.th)) * (stt.s1 * stt.s2);
    sl += ps > 0.0 ? log(ps) : NAN;
  }
  nv[(NV_P + 0) * K + k] = c1; nv[(NV_P + 1) * K + k] = cinf; nv[(NV_P + 2) * K + k] = sl; nv[(NV_P + 3) * K + k] = pmin;
  nv[(NV_P + 4) * K + k] = pmax; nv[(NV_P + 5) * K + k] = zsum; nv[(NV_P + 6) * K + k] = l1; nv[(NV_P + 7) * K + k] = rth;
  nv[(NV_P + 8) * K + k] = g.dc ? 0.0 : Huth; nv[(NV_P + 9) * K + k] = Hthth;
  nv[(NV_P + 10) * K + k] = g.dc ? params[p].dcost * (pp + pn) : 0.0;
}

// ==============================================================================================================
// wave-level dense 8x8 primitives: lane l holds element (i, j) = (l >> 3, l & 7)
// ==============================================================================================================
struct Lds8 {
  double A[64], B[64];
};
ASC_DEV void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// C = op(A) B, op = transpose if TA
template <bool TA>
ASC_DEV double mm(Lds8 &L, double a, double b) {
  const int l = threadIdx.x, i = l >> 3, j = l & 7;
  L.A[l] = a; L.B[l] = b;
  wsync();
  double acc = 0.0;
  ASC_UNROLL
  for (int q = 0; q < 8; q++) acc = fma(TA ? L.A[q * 8 + i] : L.A[i * 8 + q], L.B[q * 8 + j], acc);
  wsync();
  return acc;
}
ASC_DEV double tr8(double v) { const int l = threadIdx.x; return __shfl(v, ((l & 7) << 3) | (l >> 3)); }
ASC_DEV double rowsum(double v) {     // sum over j, result in every lane of the row
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
  return v;
}
ASC_DEV double colsum(double v) {     // sum over i, result in every lane of the column
  v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
  return v;
}
ASC_DEV double wmax(double v) {
  ASC_UNROLL
  for (int o = 1; o < 64; o <<= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
ASC_DEV double wmin(double v) {
  ASC_UNROLL
  for (int o = 1; o < 64; o <<= 1) v = fmin(v, __shfl_xor(v, o));
  return v;
}
ASC_DEV double wsum(double v) {
  ASC_UNROLL
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  return v;
}
// element (r, c) of a grid, to every lane
ASC_DEV double pick(double v, int r, int c) { return __shfl(v, (r << 3) | c); }
// row r of a grid as a "column-indexed" vector: lane (i,j) gets element (r, j)
ASC_DEV double rowvec(double v, int r) { return __shfl(v, (r << 3) | (threadIdx.x & 7)); }
// row r of a grid as a "row-indexed" vector: lane (i,j) gets element (r, i)
ASC_DEV double rowvec_t(double v, int r) { return __shfl(v, (r << 3) | (threadIdx.x >> 3)); }
// column c of a grid as a "row-indexed" vector: lane (i,j) gets element (i, c)
ASC_DEV double colvec(double v, int c) { return __shfl(v, (threadIdx.x & ~7) | c); }

// Inverse of a grid by Gauss-Jordan elimination without pivoting (the step Jacobian Jb = I - dt*(...) keeps unit-size
// pivots in this order); `bad` is raised by a vanishing or non-finite pivot.
ASC_DEV double ginv(Lds8 &L, double m, int &bad) {
  const int l = threadIdx.x, i = l >> 3, j = l & 7;
  double v = i == j ? 1.0 : 0.0;
  for (int q = 0; q < 8; q++) {
    L.A[l] = m; L.B[l] = v;
    wsync();
    const double piv = L.A[q * 8 + q], mik = L.A[i * 8 + q], mkj = L.A[q * 8 + j], vkj = L.B[q * 8 + j];
    wsync();
    if (!(fabs(piv) > 1e-300)) bad = 1;
    const double ip = 1.0 / piv;
    if (i == q) { m = mkj * ip; v = vkj * ip; }
    else { const double f = mik * ip; m = fma(-f, mkj, m); v = fma(-f, vkj, v); }
  }
  return v;
}

// ==============================================================================================================
// d_newton: one wavefront = one NLP
// ==============================================================================================================
struct Border { double Oth, Otn, Onn, oth, onu; };     // Omega (2x2 symmetric) and omega of the value function

// MODE 0: decisions + Riccati solve + step (the whole Newton iteration).  MODE 1: decisions only; an NLP that needs a Newton
// step is handed to the PCR kernels (pc_assemble -> ascent_blocktri's cyclic reduction -> pc_step) in state ST_NEWTON.
#ifdef DENSE_TRACE
ASC_DEV double s0th_dbg(const double *sc) { return sc[X_S + S_TH]; }
#endif
// PIPE = 1: the inputs of a recursion step are loaded one step ahead (they do not depend on the recursion; a step is
// otherwise a chain of dependent HBM round trips in front of every product).  That needs 291 registers = one wavefront per
// SIMD, which pays while the batch does not offer more than one anyway (<= 1024 NLPs: -18 %; 4096 NLPs: +11 %, so the host
// picks PIPE = 0 there: same code, loads at the top of the step, two wavefronts per SIMD).
template <int MODE, int PIPE = 0>
__global__ __launch_bounds__(WAVE, PIPE ? 1 : 2) void d_newton(const ascent_params *params, long batch, DGeo g, double *ws, int max_iter,
                                                 double tol, int probe, const double *probe_dw, int *counters) {
  __shared__ Lds8 L;
  const long p = blockIdx.x;
  const int l = threadIdx.x, i = l >> 3, j = l & 7, K = g.K;
  double *w = ws + (size_t)p * g.nlp_doubles();
  double *sc = w + g.off_sc();
  if ((int)sc[X_STATE] != ST_TRIAL) return;
  const Der d = derive_t(params[p], g.terminal);
  double *nv = w + g.off_nv();
  const bool first = sc[X_FIRST] != 0.0;
  double mu = sc[X_MU], nu_pen = sc[X_NUP];
  int cur = (int)sc[X_CUR];
  // ---- reduce the partials of the trial point (lane = step, strided) -----------------------------------------------
  double c1 = 0.0, cinf = 0.0, sl = 0.0, pmin = 1e300, pmax = -1e300, zsum = 0.0, l1 = 0.0, rth = 0.0, mv = 0.0;
  for (int k = l; k < K; k += WAVE) {
    c1 += nv[(NV_P + 0) * K + k]; cinf = fmax(cinf, nv[(NV_P + 1) * K + k]); sl += nv[(NV_P + 2) * K + k];
    pmin = fmin(pmin, nv[(NV_P + 3) * K + k]); pmax = fmax(pmax, nv[(NV_P + 4) * K + k]);
    zsum += nv[(NV_P + 5) * K + k]; l1 += nv[(NV_P + 6) * K + k]; rth += nv[(NV_P + 7) * K + k];
    mv += nv[(NV_P + 10) * K + k];
  }
  c1 = wsum(c1); cinf = wmax(cinf); sl = wsum(sl); pmin = wmin(pmin); pmax = wmax(pmax); zsum = wsum(zsum); l1 = wsum(l1);
  rth = 1.0 + wsum(rth);
  mv = g.dc ? wsum(mv) : 0.0;                     // the move penalty's part of the objective
  const double dcw = g.dc ? params[p].dcost : 0.0;
  const Scal stt = load_scal(sc, X_T);
  double iters = sc[X_ITERS];
  if (!first) {     // Armijo test on the l1 merit function
    const double alpha = sc[X_ALPHA], phi0 = sc[X_PHI0], Dm = sc[X_DM];
    const double phit = (stt.th + mv) - mu * sl + nu_pen * c1;
#ifdef DENSE_TRACE
    if (l == 0 && p == 0) printf("[dense] K=%d it %d ls %d alpha %.3g: phit-phi0 %.3e (th %.3e mv %.3e) alpha*Dm %.3e c1 %.3e nu %.3g mu %.1e\n", K, (int)iters, (int)sc[X_LS], alpha, phit - phi0, stt.th - s0th_dbg(sc), mv, alpha * Dm, c1, nu_pen, mu);
#endif
    if (!(isfinite(phit) && phit <= phi0 + 1e-8 * alpha * Dm + 2.220446049250313e-15 * fabs(phi0))) {
      const int ls = (int)sc[X_LS] + 1;
      if (l == 0) {
        sc[X_LS] = ls;
        if (ls >= 40) { sc[X_STATUS] = ASCENT_LINESEARCH_FAILED; sc[X_STATE] = ST_DONE; }
        else { sc[X_ALPHA] = 0.5 * alpha; atomicAdd(&counters[0], 1); }
      }
      return;
    }
    iters += 1.0;
  }
  // ---- accepted: the trial point is the iterate ---------------------------------------------------------------------
  const Scal s = stt;
  cur = 1 - cur;
  const double *it = w + g.off_it(cur);
  // dual residual: node k collects Jb_k'lambda_k + Ja_{k+1}'lambda_{k+1} and its bound multipliers
  double rd = 0.0;
  const Terminal tm = [&]() { double zK[7]; for (int q = 0; q < 7; q++) zK[q] = it[(O_Z + q) * K + K - 1]; return terminal_of(d, zK); }();
  for (int k = l; k < K; k += WAVE) {
    double r[7];
    ASC_UNROLL
    for (int q = 0; q < 7; q++) r[q] = nv[(NV_GB + q) * K + k] + (k + 1 < K ? nv[(NV_GA + q) * K + k + 1] : 0.0);
    r[IA] += it[(O_ZB + 1) * K + k] - it[(O_ZB + 0) * K + k];
    r[IM] += it[(O_ZB + 3) * K + k] - it[(O_ZB + 2) * K + k];
    if (k == K - 1 && d.term == 2) {
      double g4[4];
      terminal_grad_any(tm, s.nu1, s.nu2, g4);
      r[IX] += g4[0]; r[IY] += g4[1]; r[IVX] += g4[2]; r[IVY] += g4[3];
    } else if (k == K - 1) {
      r[IX] += s.nu3 * tm.e3g[0] + s.nu1 * tm.g1g[0];
      r[IY] += s.nu3 * tm.e3g[1] + s.nu1 * tm.g1g[1];
      r[IVX] += s.nu3 * tm.e3g[2] + s.nu2 * tm.g2g[0];
      r[IVY] += s.nu3 * tm.e3g[3] + s.nu2 * tm.g2g[1];
    }
    ASC_UNROLL
    for (int q = 0; q < 7; q++) rd = fmax(rd, fabs(r[q]));
    const double dt = (1.0 / K) * d.T * s.th;
    if (g.dc) {      // the control's row carries the multipliers of its two movement equations; rows of the slack pair
      const double r7 = nv[(NV_GB + 7) * K + k] + (k + 1 < K ? nv[(NV_GA + 7) * K + k + 1] : 0.0);
      rd = fmax(rd, fabs(r7 - it[(O_ZB + 4) * K + k] + it[(O_ZB + 5) * K + k]));
      rd = fmax(rd, fmax(fabs(dcw - it[O_LU * K + k] - it[O_ZP * K + k]), fabs(dcw + it[O_LU * K + k] - it[O_ZN * K + k])));
    } else {
      rd = fmax(rd, fabs(-dt * d.alpha * it[(O_L + IW) * K + k] - it[(O_ZB + 4) * K + k] + it[(O_ZB + 5) * K + k]));
    }
  }
  rd = wmax(rd);
  ErrParts e;
  e.rd = fmax(rd, fabs(rth - s.zlt + s.zut));
  e.rd = fmax(e.rd, fmax(fabs(-s.nu1 - s.zs1), fabs(-s.nu2 - s.zs2)));
  e.cinf = cinf;
  {
    const double pr[4] = {(s.th - d.tlb) * s.zlt, (d.tub - s.th) * s.zut, s.s1 * s.zs1, s.s2 * s.zs2};
    ASC_UNROLL
    for (int q = 0; q < 4; q++) { pmin = fmin(pmin, pr[q]); pmax = fmax(pmax, pr[q]); }
  }
  e.pmin = pmin; e.pmax = pmax;
  l1 += fabs(s.nu3) + fabs(s.nu1) + fabs(s.nu2);
  zsum += s.zlt + s.zut + s.zs1 + s.zs2;
  e.sd = fmax(100.0, (l1 + zsum) / (double)((g.dc ? 16 : 13) * K + 7)) * 0.01;
  if (l == 0) {
    store_scal(sc, X_S, s);
    sc[X_CUR] = cur; sc[X_FIRST] = 0.0; sc[X_ITERS] = iters; sc[X_LS] = 0.0; sc[X_C1] = c1; sc[X_SL] = sl;
  }
  double dw = 0.0;
#ifdef DENSE_TRACE
  if (l == 0 && p == 0) printf("[dense] K=%d accepted it %d: E0 %.3e (dual %.2e primal %.2e compl %.2e..%.2e) mu %.1e th %.10f mv %.3e\n", K, (int)iters, e.err(0.0), e.rd, e.cinf, e.pmin, e.pmax, mu, s.th, mv);
#endif
  if (!probe) {
    if (e.err(0.0) <= tol) {
      if (l == 0) { sc[X_STATUS] = ASCENT_CONVERGED; sc[X_STATE] = ST_DONE; }
      return;
    }
    if ((int)iters >= max_iter) {
      if (l == 0) { sc[X_STATUS] = ASCENT_MAX_ITER; sc[X_STATE] = ST_DONE; }
      return;
    }
    while (mu > tol * 0.1 && e.err(mu) <= 10.0 * mu) {
      mu = fmax(tol * 0.1, fmin(0.2 * mu, mu * sqrt(mu)));
      nu_pen = 1.0;
    }
  } else {
    dw = probe_dw[p];
  }
  if (MODE == 1) {
    if (l == 0) {
      sc[X_MU] = mu; sc[X_NUP] = nu_pen; sc[X_RTH] = rth; sc[X_DW] = dw; sc[X_MV] = mv; sc[X_STATE] = ST_NEWTON;
      atomicAdd(&counters[0], 1);
    }
    return;
  }
  // ---- Newton step: backward recursion (repeated with a larger delta_w while the inertia is wrong) ---------------------
  const double hT = (1.0 / K) * d.T, dt = hT * s.th;
  const double is1 = rcp(s.s1), is2 = rcp(s.s2);
  const double cg1 = tm.g1 - s.s1, cg2 = tm.g2 - s.s2;
  const double *rec0 = w + g.off_rec();
  double *fwd0 = w + g.off_fwd();
  double dw_last = sc[X_DWL];
  double dth = 0.0, dnu3 = 0.0, sig1 = 0.0, sig2 = 0.0, rs1 = 0.0, rs2 = 0.0;
  int refac = 0;
  for (;;) {
    sig1 = s.zs1 * is1 + dw; sig2 = s.zs2 * is2 + dw;
    rs1 = -mu * is1 - s.nu1; rs2 = -mu * is2 - s.nu2;
    // the terminal terms of the last node, per lane: its element of the terminal Hessian, its rows of the terminal gradients
    double term_add = 0.0, term_rt = 0.0, term_e3 = 0.0;
    {
      const double w1 = s.nu1 + sig1 * cg1 + rs1, w2 = s.nu2 + sig2 * cg2 + rs2;
      double QT[28];
      ASC_UNROLL
      for (int q = 0; q < 28; q++) QT[q] = 0.0;
      if (d.term == 2) terminal_hessian_any(QT, tm, s.nu1, s.nu2, sig1, sig2);
      else terminal_hessian(QT, tm, s.nu3, s.nu1, s.nu2, sig1, sig2);
      ASC_UNROLL
      for (int a = 0; a < 4; a++) {
        ASC_UNROLL
        for (int b = 0; b < 4; b++) term_add = (i == a && j == b) ? QT[sid(a, b)] : term_add;
      }
      double rt4[4] = {s.nu3 * tm.e3g[0] + w1 * tm.g1g[0], s.nu3 * tm.e3g[1] + w1 * tm.g1g[1],
                       s.nu3 * tm.e3g[2] + w2 * tm.g2g[0], s.nu3 * tm.e3g[3] + w2 * tm.g2g[1]};
      if (d.term == 2) terminal_grad_any(tm, w1, w2, rt4);
      ASC_UNROLL
      for (int a = 0; a < 4; a++) { term_rt = i == a ? rt4[a] : term_rt; term_e3 = i == a ? tm.e3g[a] : term_e3; }
    }
    double P = 0.0;                      // value-function Hessian (grid)
    double R = 0.0;                      // grid of vectors: column 0 = p, column 1 = Pi_theta, column 2 = Pi_nu3
    Border B{0.0, 0.0, 0.0, 0.0, tm.e3};
    int bad = 0;
    // One step's inputs are loaded while the step before is computed: they do not depend on the recursion, and a step is
    // otherwise a chain of dependent HBM round trips (records, node quantities) in front of every product.
    struct StepIn { double Ja, Jb, Haa, Hab, Hbb, V, a_, m_, u_, zb[6], gb, ga, lw, Huth, Hthth, pp, pn, zp, zn, lu; };
    auto load_step = [&](int k, StepIn &q) __attribute__((always_inline)) {
      const double *rec = rec0 + (size_t)k * NGRID * 64;
      q.Ja = rec[G_JA * 64 + l]; q.Jb = rec[G_JB * 64 + l]; q.Haa = rec[G_HAA * 64 + l]; q.Hab = rec[G_HAB * 64 + l];
      q.Hbb = rec[G_HBB * 64 + l]; q.V = rec[G_V * 64 + l];
      q.a_ = it[(O_Z + IA) * K + k]; q.m_ = it[(O_Z + IM) * K + k]; q.u_ = it[O_U * K + k];
      ASC_UNROLL
      for (int b = 0; b < 6; b++) q.zb[b] = it[(O_ZB + b) * K + k];
      q.gb = nv[(NV_GB + i) * K + k]; q.ga = k + 1 < K ? nv[(NV_GA + i) * K + k + 1] : 0.0;
      q.lw = it[(O_L + IW) * K + k];
      q.Huth = nv[(NV_P + 8) * K + k]; q.Hthth = nv[(NV_P + 9) * K + k];
      if (g.dc) { q.pp = it[O_PP * K + k]; q.pn = it[O_PN * K + k]; q.zp = it[O_ZP * K + k]; q.zn = it[O_ZN * K + k]; q.lu = it[O_LU * K + k]; }
      else { q.pp = q.pn = 1.0; q.zp = q.zn = q.lu = 0.0; }
    };
    StepIn cur_, nxt_;
    if (PIPE) { load_step(K - 1, cur_); nxt_ = cur_; }
    for (int k = K - 1; k >= 0; k--) {
      if (!PIPE) load_step(k, cur_);
      else if (k > 0) load_step(k - 1, nxt_);
      const double Ja = cur_.Ja, Jb = cur_.Jb, Haa = cur_.Haa, Hab = cur_.Hab;
      double Fxx = cur_.Hbb + P;
      const double V = cur_.V;
      // node quantities of this step
      const double a_ = cur_.a_, m_ = cur_.m_, u_ = cur_.u_;
      const double id0 = rcp(a_), id1 = rcp(d.aub - a_), id2 = rcp(m_), id3 = rcp(1.0 - m_), id4 = rcp(u_ + 1.0), id5 = rcp(1.0 - u_);
      const double siga = cur_.zb[0] * id0 + cur_.zb[1] * id1;
      const double sigm = cur_.zb[2] * id2 + cur_.zb[3] * id3;
      double sigu = cur_.zb[4] * id4 + cur_.zb[5] * id5;
      if (i == j) Fxx += (i == IA ? siga : i == IM ? sigm : (i == 7 && g.dc) ? sigu : 0.0) + dw;
      // residual of node k (barrier form), row-indexed: r_i in lane (i, *)
      double rx = (i < 7 || g.dc) ? cur_.gb + cur_.ga : 0.0;
      if (i == IA) rx += mu * (id1 - id0);
      if (i == IM) rx += mu * (id3 - id2);
      if (i == 7 && g.dc) rx += mu * (id5 - id4);
      double Fxb_nu = colvec(R, 2);        // F_x,nu3 = Pi_nu3 (+ terminal gradient of r.v), row-indexed
      if (k == K - 1) { Fxx += term_add; rx += term_rt; Fxb_nu += term_e3; }      // the last node's terminal terms (formed above)
      double ru = -dt * d.alpha * cur_.lw + mu * (id5 - id4);     // Ju'lambda + barrier gradient
      if (g.dc) {      // the stage's control is delta = p - n: curvature and gradient of the reduced slack pair
        const double pp = cur_.pp, pn = cur_.pn, lu = cur_.lu;
        const double ip = rcp(pp), in_ = rcp(pn);
        const double isp = rcp(cur_.zp * ip + dw), isn = rcp(cur_.zn * in_ + dw);
        sigu = rcp(isp + isn);
        ru = sigu * ((dcw - mu * ip - lu) * isp - (dcw - mu * in_ + lu) * isn);
      }
      const double Huth = cur_.Huth, Hthth = cur_.Hthth;
      // vectors of the record, row-indexed (component i in lane (i,*))
      const double cvec = rowvec_t(V, 0), Ju = rowvec_t(V, 1), Jth = rowvec_t(V, 2), Hath = rowvec_t(V, 3), Hbth = rowvec_t(V, 4);
      const double fx = rx + colvec(R, 0);                       // f_x = r^x + p
      const double Fxb_th = Hbth + colvec(R, 1);                 // F_x,theta = Hb,theta + Pi_theta
      // E = Jb^-1 [Ja | Ju Jth 0 c]
      const double Jbi = ginv(L, Jb, bad);
      const double Ea = mm<false>(L, Jbi, Ja);
      const double B2 = j == 0 ? Ju : j == 1 ? Jth : j == 3 ? cvec : 0.0;
      const double E2 = mm<false>(L, Jbi, B2);                   // columns: Eu, Eth, 0, e
      // T = Fxx E - Fxy  (Fx,xi = Hab', Fxu = 0, Fx,th, Fx,nu; constant column: Fxx e - fx)
      const double Ta = mm<false>(L, Fxx, Ea) - tr8(Hab);
      const double T2 = mm<false>(L, Fxx, E2) - (j == 1 ? Fxb_th : j == 2 ? Fxb_nu : j == 3 ? fx : 0.0);
      // Lambda = Jb^-T T
      const double La = mm<true>(L, Jbi, Ta);
      const double L2 = mm<true>(L, Jbi, T2);                    // columns: Lu, Lth, Lnu, Le
      // G = Fyy - Fyx E + J*' Lambda
      const double Gxx = Haa - mm<false>(L, Hab, Ea) + mm<true>(L, Ja, La);
      const double G2 = (j == 1 ? Hath : 0.0) - mm<false>(L, Hab, E2) + mm<true>(L, Ja, L2);   // columns: G_xi,u  G_xi,th  G_xi,nu  g_xi
      // scalar rows of G for y in (u, theta, nu):  Y' L2  and  Y' E2  with Y = [Ju | Jth | Fxb_th | Fxb_nu]
      const double Y = j == 0 ? Ju : j == 1 ? Jth : j == 2 ? Fxb_th : j == 3 ? Fxb_nu : 0.0;
      const double YL = mm<true>(L, Y, L2), YE = mm<true>(L, Y, E2);
      const double Guu = sigu + (g.dc ? 0.0 : dw) + pick(YL, 0, 0);
      const double Guth = Huth + pick(YL, 0, 1), Gunu = pick(YL, 0, 2), gu = ru + pick(YL, 0, 3);
      const double Gthth = Hthth + B.Oth - pick(YE, 2, 1) + pick(YL, 1, 1);
      const double Gthnu = B.Otn + pick(YL, 1, 2);
      const double Gnunu = B.Onn;
      const double gth = B.oth - pick(YE, 2, 3) + pick(YL, 1, 3);
      const double gnu = B.onu - pick(YE, 3, 3);
      if (!(Guu > 0.0)) bad = 1;
      const double iD = 1.0 / Guu;
      // gains of the control: u = -(ku_xi' xi + ku_th dth + ku_nu dnu + ku_0)
      const double Gxu_r = colvec(G2, 0);                        // G_xi,u row-indexed (component i)
      const double Gxu_c = tr8(Gxu_r);                           // column-indexed (component j)
      const double ku_th = Guth * iD, ku_nu = Gunu * iD, ku_0 = gu * iD;
      P = Gxx - Gxu_r * Gxu_c * iD;
      const double gxi = colvec(G2, 3), Gxth = colvec(G2, 1), Gxnu = colvec(G2, 2);     // (shuffles stay outside divergent code)
      R = j == 0 ? gxi - Gxu_r * ku_0 : j == 1 ? Gxth - Gxu_r * ku_th : j == 2 ? Gxnu - Gxu_r * ku_nu : 0.0;
      B.Oth = Gthth - Guth * ku_th; B.Otn = Gthnu - Guth * ku_nu; B.Onn = Gnunu - Gunu * ku_nu;
      B.oth = gth - Guth * ku_0; B.onu = gnu - Gunu * ku_0;
      // forward record
      double *fw = fwd0 + (size_t)k * FWD_DOUBLES;
      fw[F_EA * 64 + l] = Ea;
      fw[F_LA * 64 + l] = La;
      const double L2s = __shfl(L2, (i << 3) | (j & 3));
      fw[F_EL2 * 64 + l] = j < 4 ? E2 : L2s;                     // [Eu Eth 0 e | Lu Lth Lnu Le]
      if (i == 0) fw[F_GAIN * 64 + j] = Gxu_c * iD;
      if (l == 8) { fw[F_GAIN * 64 + 8] = ku_th; fw[F_GAIN * 64 + 9] = ku_nu; fw[F_GAIN * 64 + 10] = ku_0; }
      if (PIPE) cur_ = nxt_;
    }
    bad = __any(bad);
    int ok = !bad;
    if (ok) {
      const double itl = rcp(s.th - d.tlb), itu = rcp(d.tub - s.th);
      const double rthp = rth + mu * (itu - itl);
      const double a11 = s.zlt * itl + s.zut * itu + dw + B.Oth, a12 = B.Otn, a22 = d.term == 2 ? -1.0 : B.Onn;    // (terminal 2: no r.v = 0; a unit pivot closes the nu3 row)
      const double b1 = -(rthp + B.oth), b2 = -B.onu;
      const double det = a11 * a22 - a12 * a12;
      if (det < 0.0) {
        const double idet = 1.0 / det;
        dth = (b1 * a22 - a12 * b2) * idet;
        dnu3 = (a11 * b2 - a12 * b1) * idet;
      } else {
        ok = 0;
      }
    }
    if (ok) break;
    if (probe) {
      if (l == 0) { sc[X_STATE] = ST_DONE; sc[X_STATUS] = ASCENT_REGULARISATION_FAILED; }
      return;
    }
    dw = next_delta_w(dw, dw_last);
    refac++;
    if (dw > 1e10) {
      if (l == 0) { sc[X_STATUS] = ASCENT_REGULARISATION_FAILED; sc[X_STATE] = ST_DONE; }
      return;
    }
  }
  // ---- forward substitution: dz, du, dlambda; bound-multiplier steps; fraction to the boundary -----------------------
  const double tau = fmax(0.99, 1.0 - mu);
  double *st = w + g.off_st();
  double xi = 0.0;                     // dz_{k-1}, column-indexed (component j in lane (*, j))
  double rmax = 0.0, gsum = 0.0, adu = 1.0, cl = 0.0, dzK = 0.0, gmove = 0.0;
  // (the inputs of a step are loaded while the step before is computed, as in the backward recursion)
  struct FwdIn { double Ea, La, EL2, kg, ku_th, ku_nu, ku_0, a_, m_, u_, zb[6], V, lam_i, pp, pn, zp, zn, lu; };
  auto load_fwd = [&](int k, FwdIn &q) __attribute__((always_inline)) {
    const double *fw = fwd0 + (size_t)k * FWD_DOUBLES;
    q.Ea = fw[F_EA * 64 + l]; q.La = fw[F_LA * 64 + l]; q.EL2 = fw[F_EL2 * 64 + l];
    q.kg = fw[F_GAIN * 64 + j];
    q.ku_th = fw[F_GAIN * 64 + 8]; q.ku_nu = fw[F_GAIN * 64 + 9]; q.ku_0 = fw[F_GAIN * 64 + 10];
    q.a_ = it[(O_Z + IA) * K + k]; q.m_ = it[(O_Z + IM) * K + k]; q.u_ = it[O_U * K + k];
    ASC_UNROLL
    for (int b = 0; b < 6; b++) q.zb[b] = it[(O_ZB + b) * K + k];
    q.V = (w + g.off_rec() + (size_t)k * NGRID * 64)[G_V * 64 + l];
    q.lam_i = i < 7 ? it[(O_L + i) * K + k] : g.dc ? it[O_LU * K + k] : 0.0;
    if (g.dc) { q.pp = it[O_PP * K + k]; q.pn = it[O_PN * K + k]; q.zp = it[O_ZP * K + k]; q.zn = it[O_ZN * K + k]; q.lu = it[O_LU * K + k]; }
    else { q.pp = q.pn = 1.0; q.zp = q.zn = q.lu = 0.0; }
  };
  FwdIn fc_, fn_;
  if (PIPE) { load_fwd(0, fc_); fn_ = fc_; }
  for (int k = 0; k < K; k++) {
    if (!PIPE) load_fwd(k, fc_);
    else if (k + 1 < K) load_fwd(k + 1, fn_);
    const double Ea = fc_.Ea, La = fc_.La, EL2 = fc_.EL2;
    const double kg = fc_.kg;
    const double ku_th = fc_.ku_th, ku_nu = fc_.ku_nu, ku_0 = fc_.ku_0;
    const double du = -(rowsum(kg * xi) + ku_th * dth + ku_nu * dnu3 + ku_0);
    const double yv = j == 0 ? du : j == 1 ? dth : j == 2 ? dnu3 : j == 3 ? 1.0 : 0.0;       // (u, theta, nu, 1)
    const double yl = j == 4 ? du : j == 5 ? dth : j == 6 ? dnu3 : j == 7 ? 1.0 : 0.0;
    const double dz = -(rowsum(Ea * xi) + rowsum(EL2 * yv));        // row-indexed: dz_i in lane (i,*)
    const double dl = rowsum(La * xi) + rowsum(EL2 * yl);
    if (j == 0 && i < 7) { st[(O_Z + i) * K + k] = dz; st[(O_L + i) * K + k] = dl; }
    double du_ = du;                     // the step of the control: the stage's control, or the eighth state
    if (g.dc) {
      const double dlu = pick(dl, 7, 0);
      du_ = pick(dz, 7, 0);
      const double pp = fc_.pp, pn = fc_.pn, zp = fc_.zp, zn = fc_.zn, lu = fc_.lu;
      const double ip = rcp(pp), in_ = rcp(pn);
      // the slack with the larger curvature from its own row (well conditioned), the other from delta = p - n (its own row
      // divides a difference of two nearly equal numbers by a curvature that vanishes for an inactive slack)
      const double sgp = zp * ip + dw, sgn_ = zn * in_ + dw;
      double dpp, dpn;
      if (sgp >= sgn_) { dpp = (dlu - (dcw - mu * ip - lu)) * rcp(sgp); dpn = dpp - du; }
      else { dpn = (-dlu - (dcw - mu * in_ + lu)) * rcp(sgn_); dpp = du + dpn; }
      const double dzp = ip * (mu - zp * dpp) - zp, dzn = in_ * (mu - zn * dpn) - zn;
      ASC_FTBR(rmax, ip, dpp); ASC_FTBR(rmax, in_, dpn);
      ASC_FTB(adu, zp, dzp); ASC_FTB(adu, zn, dzn);
      gsum -= dpp * ip + dpn * in_;
      gmove += dpp + dpn;
#ifdef DENSE_TRACE
      if (l == 0 && p == 0 && (fabs(dpp) > 1.0 || fabs(dpn) > 1.0 || fabs(du) > 1.0 || fabs(du_) > 1.0))
        printf("[dense fwd] k %d ddelta %.3e du %.3e dlu %.3e | pp %.3e pn %.3e zp %.3e zn %.3e lu %.3e | dpp %.3e dpn %.3e | gains th %.3e nu %.3e 0 %.3e dw %.1e\n", k, du, du_, dlu, pp, pn, zp, zn, lu, dpp, dpn, ku_th, ku_nu, ku_0, dw);
#endif
      if (l == 0) { st[O_LU * K + k] = dlu; st[O_PP * K + k] = dpp; st[O_PN * K + k] = dpn; st[O_ZP * K + k] = dzp; st[O_ZN * K + k] = dzn; }
    }
    if (l == 0) st[O_U * K + k] = du_;
    // bound multipliers and both fraction-to-boundary rules (every lane redundantly)
    const double dza = pick(dz, IA, 0), dzm = pick(dz, IM, 0);
    const double a_ = fc_.a_, m_ = fc_.m_, u_ = fc_.u_;
    const double id[6] = {rcp(a_), rcp(d.aub - a_), rcp(m_), rcp(1.0 - m_), rcp(u_ + 1.0), rcp(1.0 - u_)};
    ASC_FTBR(rmax, id[0], dza); ASC_FTBR(rmax, id[1], -dza);
    ASC_FTBR(rmax, id[2], dzm); ASC_FTBR(rmax, id[3], -dzm);
    ASC_FTBR(rmax, id[4], du_); ASC_FTBR(rmax, id[5], -du_);
    gsum += dza * (id[1] - id[0]) + dzm * (id[3] - id[2]) + du_ * (id[5] - id[4]);
    const double dx3[3] = {dza, dzm, du_};
    ASC_UNROLL
    for (int b = 0; b < 3; b++) {
      const double zl = fc_.zb[2 * b], zu = fc_.zb[2 * b + 1];
      const double dzl = id[2 * b] * (mu - zl * dx3[b]) - zl, dzu = id[2 * b + 1] * (mu + zu * dx3[b]) - zu;
      ASC_FTB(adu, zl, dzl);
      ASC_FTB(adu, zu, dzu);
      if (l == 0) { st[(O_ZB + 2 * b) * K + k] = dzl; st[(O_ZB + 2 * b + 1) * K + k] = dzu; }
    }
    // c'(lambda + dlambda) for the curvature estimate of the merit function
    const double V = fc_.V;
    const double cvec = rowvec_t(V, 0);
    const double lam_i = fc_.lam_i;
    cl += j == 0 ? cvec * (lam_i + dl) : 0.0;
    xi = __shfl(dz, j << 3);           // next step's xi_j = dz_j
    if (k == K - 1) dzK = dz;
    if (PIPE) fc_ = fn_;
  }
  cl = wsum(cl);
  // ---- scalars of the step, merit bookkeeping ---------------------------------------------------------------------------
  const double dzKx = pick(dzK, IX, 0), dzKy = pick(dzK, IY, 0), dzKvx = pick(dzK, IVX, 0), dzKvy = pick(dzK, IVY, 0);
  Scal ds;
  ds.th = dth; ds.nu3 = dnu3;
  ds.s1 = (tm.g1 - s.s1) + tm.g1g[0] * dzKx + tm.g1g[1] * dzKy;
  ds.s2 = (tm.g2 - s.s2) + tm.g2g[0] * dzKvx + tm.g2g[1] * dzKvy;
  if (d.term == 2) {
    ds.s1 += tm.g1v[0] * dzKvx + tm.g1v[1] * dzKvy;
    ds.s2 += tm.g2p[0] * dzKx + tm.g2p[1] * dzKy;
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
  double gd = mu * gsum + dcw * gmove;
  gd += ds.th * (1.0 - mu / dl_ + mu / dU) - mu * ds.s1 / s.s1 - mu * ds.s2 / s.s2;
  cl += tm.e3 * (s.nu3 + ds.nu3) + (tm.g1 - s.s1) * (s.nu1 + ds.nu1) + (tm.g2 - s.s2) * (s.nu2 + ds.nu2);
  const double curv = -gd + cl;
  if (c1 > 0.0) {
    const double need = (gd + 0.5 * fmax(curv, 0.0)) / (0.9 * c1);
    if (nu_pen < need) nu_pen = need + 1.0;
  }
  if (l == 0) {
    store_scal(sc, X_D, ds);
    sc[X_MU] = mu; sc[X_NUP] = nu_pen; sc[X_DWL] = dw; sc[X_REFAC] += refac;
    sc[X_DM] = gd - nu_pen * c1;
    sc[X_PHI0] = (s.th + mv) - mu * sl + nu_pen * c1;
    sc[X_ALPHA] = apr; sc[X_ADU] = adu;
    if (probe) { sc[X_STATE] = ST_DONE; sc[X_STATUS] = 0; }
    else atomicAdd(&counters[0], 1);
  }
}


// ==============================================================================================================
// PCR variant of the Newton solve (small batches / long grids): the same Newton system, ordered by collocation node into
// 15x15 blocks (7 states, the control, the 7 defect multipliers), solved by parallel cyclic reduction over the nodes on
// v_mfma_f64_16x16x4_f64 tiles (ascent_blocktri.hip) instead of the serial Riccati recursion: log2(K) levels of one
// wavefront per node.  Inside a node the EQUATIONS are ordered (defect rows, control row, state rows) against the
// UNKNOWNS (states, control, multipliers): the diagonal block then starts with the step Jacobian Jb ~ I, so the
// unpivoted Gauss-Jordan inverse of ascent_blocktri meets unit-size pivots (in the symmetric order the leading block
// would be the Lagrangian Hessian, which is tiny and indefinite).  PCR exposes no inertia; this variant regularises on
// the curvature dx'(W + Sigma + delta)dx along the step, which the merit function needs anyway.
// ==============================================================================================================
// With the move penalty (DC = 1) the control is the eighth state and the stage's control delta = p - n is eliminated into the
// movement equation (d delta = (d lambda_u - g)/R: a diagonal entry -1/R): 16x16 blocks, rows (8 constraint rows, 8
// stationarity rows) against unknowns (8 states, 8 multipliers).
constexpr int PC_NB = 2, PC_BS = 15, PC_BSMAX = 16;
template <int DC>
__global__ __launch_bounds__(WAVE) void pc_assemble(const ascent_params *params, long batch, DGeo g, const double *ws, double *bt,
                                                    double *bcols, size_t node_doubles) {
  const long p = blockIdx.y;
  const int k = blockIdx.x, K = g.K, l = threadIdx.x;
  const double *w = ws + (size_t)p * g.nlp_doubles();
  const double *sc = w + g.off_sc();
  if ((int)sc[X_STATE] != ST_NEWTON) return;
  const Der d = derive_t(params[p], g.terminal);
  const Scal s = load_scal(sc, X_S);
  const double mu = sc[X_MU], dw = sc[X_DW];
  const double *it = w + g.off_it((int)sc[X_CUR]), *nv = w + g.off_nv();
  const double *rk = w + g.off_rec() + (size_t)k * NGRID * 64;
  const double *rn = rk + NGRID * 64;                         // record of step k+1 (valid if k+1 < K)
  const bool nxt = k + 1 < K, last = k == K - 1;
  const double hT = (1.0 / K) * d.T, dt = hT * s.th;
  const double a_ = it[(O_Z + IA) * K + k], m_ = it[(O_Z + IM) * K + k], u_ = it[O_U * K + k];
  const double id0 = rcp(a_), id1 = rcp(d.aub - a_), id2 = rcp(m_), id3 = rcp(1.0 - m_), id4 = rcp(u_ + 1.0), id5 = rcp(1.0 - u_);
  const double siga = it[(O_ZB + 0) * K + k] * id0 + it[(O_ZB + 1) * K + k] * id1;
  const double sigm = it[(O_ZB + 2) * K + k] * id2 + it[(O_ZB + 3) * K + k] * id3;
  const double sigu = it[(O_ZB + 4) * K + k] * id4 + it[(O_ZB + 5) * K + k] * id5;
  double QT[28], rt4[4] = {0.0, 0.0, 0.0, 0.0}, e3g[4] = {0.0, 0.0, 0.0, 0.0};
  ASC_UNROLL
  for (int q = 0; q < 28; q++) QT[q] = 0.0;
  if (last) {
    double zK[7];
    for (int q = 0; q < 7; q++) zK[q] = it[(O_Z + q) * K + K - 1];
    const Terminal tm = terminal_of(d, zK);
    const double is1 = rcp(s.s1), is2 = rcp(s.s2);
    const double sig1 = s.zs1 * is1 + dw, sig2 = s.zs2 * is2 + dw, rs1 = -mu * is1 - s.nu1, rs2 = -mu * is2 - s.nu2;
    const double w1 = s.nu1 + sig1 * (tm.g1 - s.s1) + rs1, w2 = s.nu2 + sig2 * (tm.g2 - s.s2) + rs2;
    if (d.term == 2) {      // burnout anywhere on the ellipse: the generalised terminal block; no r.v = 0 (e3g stays zero)
      terminal_hessian_any(QT, tm, s.nu1, s.nu2, sig1, sig2);
      terminal_grad_any(tm, w1, w2, rt4);
    } else {
      terminal_hessian(QT, tm, s.nu3, s.nu1, s.nu2, sig1, sig2);
      rt4[0] = s.nu3 * tm.e3g[0] + w1 * tm.g1g[0]; rt4[1] = s.nu3 * tm.e3g[1] + w1 * tm.g1g[1];
      rt4[2] = s.nu3 * tm.e3g[2] + w2 * tm.g2g[0]; rt4[3] = s.nu3 * tm.e3g[3] + w2 * tm.g2g[1];
    }
    ASC_UNROLL
    for (int q = 0; q < 4; q++) e3g[q] = tm.e3g[q];
  }
  double *o = bt + ((size_t)p * K + k) * node_doubles;
  const int c = l & 15;
  double iR = 0.0, gR = 0.0;                                  // (DC) 1/R and g/R of the reduced slack pair
  if (DC) {
    const double pp = it[O_PP * K + k], pn = it[O_PN * K + k], lu = it[O_LU * K + k], dcw = params[p].dcost;
    const double ip = rcp(pp), in_ = rcp(pn);
    const double isp = rcp(it[O_ZP * K + k] * ip + dw), isn = rcp(it[O_ZN * K + k] * in_ + dw);
    iR = isp + isn;
    gR = (dcw - mu * ip - lu) * isp - (dcw - mu * in_ + lu) * isn;
  }
  ASC_UNROLL
  for (int q = 0; q < 4; q++) {
    const int rp = (l >> 4) + 4 * q;                          // physical row: 0-6 defect rows, 7 control row, 8-14 state rows
    double L = 0.0, D = 0.0, U = 0.0, R = 0.0;
    if (DC) {
      if (rp < 8) {                                           // constraint row i (7: the movement equation, delta eliminated)
        const int i = rp;
        if (c < 8) { D = rk[G_JB * 64 + i * 8 + c]; if (k > 0) L = rk[G_JA * 64 + i * 8 + c]; }
        else if (i == 7 && c == 15) D = -iR;
        if (c == 0) R = -rk[G_V * 64 + 0 * 8 + i] - (i == 7 ? gR : 0.0);
        else if (c == 1) R = rk[G_V * 64 + 2 * 8 + i];
      } else {                                                // stationarity with respect to state i (7: the control)
        const int i = rp - 8;
        if (c < 8) {
          D = rk[G_HBB * 64 + i * 8 + c] + (nxt ? rn[G_HAA * 64 + i * 8 + c] : 0.0);
          if (i == c) D += (i == IA ? siga : i == IM ? sigm : i == 7 ? sigu : 0.0) + dw;
          if (last) {
            ASC_UNROLL
            for (int a = 0; a < 4; a++) {
              ASC_UNROLL
              for (int b = 0; b < 4; b++) D += (i == a && c == b) ? QT[sid(a, b)] : 0.0;
            }
          }
          if (k > 0) L = rk[G_HAB * 64 + c * 8 + i];
          if (nxt) U = rn[G_HAB * 64 + i * 8 + c];
        } else {
          D = rk[G_JB * 64 + (c - 8) * 8 + i];
          if (nxt) U = rn[G_JA * 64 + (c - 8) * 8 + i];
        }
        if (c == 0) {
          double rx = nv[(NV_GB + i) * K + k] + (nxt ? nv[(NV_GA + i) * K + k + 1] : 0.0);
          if (i == IA) rx += mu * (id1 - id0);
          if (i == IM) rx += mu * (id3 - id2);
          if (i == 7) rx += mu * (id5 - id4);
          ASC_UNROLL
          for (int a = 0; a < 4; a++) rx += (last && i == a) ? rt4[a] : 0.0;
          R = -rx;
        } else if (c == 1) {
          R = rk[G_V * 64 + 4 * 8 + i] + (nxt ? rn[G_V * 64 + 3 * 8 + i] : 0.0);
        } else if (c == 2) {
          ASC_UNROLL
          for (int a = 0; a < 4; a++) R += (last && i == a) ? e3g[a] : 0.0;
        }
      }
    } else
    if (rp < 7) {                                             // defect row i: Ja dz_{k-1} + Jb dz_k + Ju du + Jth dth = -c
      const int i = rp;
      if (c < 7) { D = rk[G_JB * 64 + i * 8 + c]; if (k > 0) L = rk[G_JA * 64 + i * 8 + c]; }
      else if (c == 7) D = rk[G_V * 64 + 1 * 8 + i];
      if (c == 0) R = -rk[G_V * 64 + 0 * 8 + i];
      else if (c == 1) R = rk[G_V * 64 + 2 * 8 + i];
    } else if (rp == 7) {                                     // control row
      if (c == 7) D = sigu + dw;
      else if (c >= 8 && c < 15) D = rk[G_V * 64 + 1 * 8 + (c - 8)];
      if (c == 0) R = -(-dt * d.alpha * it[(O_L + IW) * K + k] + mu * (id5 - id4));
      else if (c == 1) R = nv[(NV_P + 8) * K + k];
    } else if (rp < 15) {                                     // state row i: stationarity with respect to z_k
      const int i = rp - 8;
      if (c < 7) {
        D = rk[G_HBB * 64 + i * 8 + c] + (nxt ? rn[G_HAA * 64 + i * 8 + c] : 0.0);
        if (i == c) D += (i == IA ? siga : i == IM ? sigm : 0.0) + dw;
        if (last) {       // (compile-time indices: a runtime index into QT would put it into scratch memory)
          ASC_UNROLL
          for (int a = 0; a < 4; a++) {
            ASC_UNROLL
            for (int b = 0; b < 4; b++) D += (i == a && c == b) ? QT[sid(a, b)] : 0.0;
          }
        }
        if (k > 0) L = rk[G_HAB * 64 + c * 8 + i];            // (Hab_k)'
        if (nxt) U = rn[G_HAB * 64 + i * 8 + c];
      } else if (c >= 8 && c < 15) {
        D = rk[G_JB * 64 + (c - 8) * 8 + i];                  // Jb'
        if (nxt) U = rn[G_JA * 64 + (c - 8) * 8 + i];         // Ja_{k+1}'
      }
      if (c == 0) {
        double rx = nv[(NV_GB + i) * K + k] + (nxt ? nv[(NV_GA + i) * K + k + 1] : 0.0);
        if (i == IA) rx += mu * (id1 - id0);
        if (i == IM) rx += mu * (id3 - id2);
        ASC_UNROLL
        for (int a = 0; a < 4; a++) rx += (last && i == a) ? rt4[a] : 0.0;
        R = -rx;
      } else if (c == 1) {
        R = rk[G_V * 64 + 4 * 8 + i] + (nxt ? rn[G_V * 64 + 3 * 8 + i] : 0.0);
      } else if (c == 2) {
        ASC_UNROLL
        for (int a = 0; a < 4; a++) R += (last && i == a) ? e3g[a] : 0.0;
      }
    } else {
      D = c == 15 ? 1.0 : 0.0;
    }
    o[0 * 256 + q * WAVE + l] = L; o[1 * 256 + q * WAVE + l] = D; o[2 * 256 + q * WAVE + l] = U; o[3 * 256 + q * WAVE + l] = R;
  }
  // the two border rows of the (symmetric) matrix over this node's unknowns (z, u, lambda), for the Schur complement
  if (l < 32) {
    const int b = l >> 4, cc = l & 15;
    double v = 0.0;
    if (b == 0 && DC) {
      if (cc < 8) v = rk[G_V * 64 + 4 * 8 + cc] + (nxt ? rn[G_V * 64 + 3 * 8 + cc] : 0.0);
      else v = rk[G_V * 64 + 2 * 8 + (cc - 8)];
    } else if (b == 0) {
      if (cc < 7) v = rk[G_V * 64 + 4 * 8 + cc] + (nxt ? rn[G_V * 64 + 3 * 8 + cc] : 0.0);
      else if (cc == 7) v = nv[(NV_P + 8) * K + k];
      else if (cc < 15) v = rk[G_V * 64 + 2 * 8 + (cc - 8)];
    } else {
      ASC_UNROLL
      for (int a = 0; a < 4; a++) v += (last && cc == a) ? e3g[a] : 0.0;
    }
    bcols[((size_t)p * K + k) * 32 + l] = v;
  }
}

// border Schur complement, the step of every node from the PCR solution, bound-multiplier steps, fraction to the boundary,
// merit bookkeeping; regularisation on the curvature along the step.  One wavefront per NLP, lanes stride over the nodes.
template <int DC>
__global__ __launch_bounds__(WAVE) void pc_step(const ascent_params *params, long batch, DGeo g, double *ws, const double *Y,
                                                const double *bcols, const int *flags, int probe, int *counters) {
  constexpr int BS = DC ? 16 : PC_BS;
  const long p = blockIdx.x;
  const int l = threadIdx.x, K = g.K;
  double *w = ws + (size_t)p * g.nlp_doubles();
  double *sc = w + g.off_sc();
  if ((int)sc[X_STATE] != ST_NEWTON) return;
  const Der d = derive_t(params[p], g.terminal);
  const Scal s = load_scal(sc, X_S);
  const double mu = sc[X_MU], dw = sc[X_DW], rth = sc[X_RTH], c1 = sc[X_C1], sl = sc[X_SL];
  double nu_pen = sc[X_NUP];
  const double *it = w + g.off_it((int)sc[X_CUR]), *nv = w + g.off_nv();
  double *st = w + g.off_st();
  const double *Yp = Y + (size_t)p * K * BS * 3, *Bp = bcols + (size_t)p * K * 32;
  const double dcw = DC ? params[p].dcost : 0.0, mv = DC ? sc[X_MV] : 0.0;
  double zK[7];
  for (int q = 0; q < 7; q++) zK[q] = it[(O_Z + q) * K + K - 1];
  const Terminal tm = terminal_of(d, zK);
  // ---- Schur complement of the two border unknowns (theta, nu3) ---------------------------------------------------------
  double hth = 0.0, s11 = 0.0, s12 = 0.0, s21 = 0.0, s22 = 0.0, t1 = 0.0, t2 = 0.0;
  for (int k = l; k < K; k += WAVE) {
    hth += nv[(NV_P + 9) * K + k];
    const double *y = Yp + (size_t)k * BS * 3, *b = Bp + (size_t)k * 32;
    for (int r = 0; r < BS; r++) {
      const double bt_ = b[r], bn = b[16 + r], yr = y[r * 3], yt = y[r * 3 + 1], yn = y[r * 3 + 2];
      s11 -= bt_ * yt; s12 -= bt_ * yn; t1 -= bt_ * yr;
      s21 -= bn * yt; s22 -= bn * yn; t2 -= bn * yr;
    }
  }
  hth = wsum(hth); s11 = wsum(s11); s12 = wsum(s12); s21 = wsum(s21); s22 = wsum(s22); t1 = wsum(t1); t2 = wsum(t2);
  const double itl = rcp(s.th - d.tlb), itu = rcp(d.tub - s.th);
  s11 += s.zlt * itl + s.zut * itu + dw + hth;
  t1 += -(rth + mu * (itu - itl));
  t2 += -tm.e3;
  if (d.term == 2) s22 = -1.0;                  // (no r.v = 0 row: its border column is zero; a unit pivot closes nu3, d nu3 = 0)
  const double det = s11 * s22 - s12 * s21;
  const double dth = (t1 * s22 - s12 * t2) / det, dnu3 = (s11 * t2 - s21 * t1) / det;
  int ok = isfinite(dth) && isfinite(dnu3) && !flags[p];
  // ---- the step of every node; bound multipliers; fraction to the boundary ------------------------------------------------
  const double tau = fmax(0.99, 1.0 - mu);
  double rmax = 0.0, gsum = 0.0, adu = 1.0, cl = 0.0, dx2 = 0.0, gmove = 0.0;
  for (int k = l; k < K; k += WAVE) {
    const double *y = Yp + (size_t)k * BS * 3;
    double x[BS];
    ASC_UNROLL
    for (int r = 0; r < BS; r++) x[r] = y[r * 3] - y[r * 3 + 1] * dth - y[r * 3 + 2] * dnu3;
    ASC_UNROLL
    for (int q = 0; q < 7; q++) { st[(O_Z + q) * K + k] = x[q]; st[(O_L + q) * K + k] = x[8 + q]; dx2 += x[q] * x[q]; }
    st[O_U * K + k] = x[7];
    dx2 += x[7] * x[7];
    const double dza = x[IA], dzm = x[IM], du = x[7];
    const double a_ = it[(O_Z + IA) * K + k], m_ = it[(O_Z + IM) * K + k], u_ = it[O_U * K + k];
    const double id[6] = {rcp(a_), rcp(d.aub - a_), rcp(m_), rcp(1.0 - m_), rcp(u_ + 1.0), rcp(1.0 - u_)};
    ASC_FTBR(rmax, id[0], dza); ASC_FTBR(rmax, id[1], -dza);
    ASC_FTBR(rmax, id[2], dzm); ASC_FTBR(rmax, id[3], -dzm);
    ASC_FTBR(rmax, id[4], du); ASC_FTBR(rmax, id[5], -du);
    gsum += dza * (id[1] - id[0]) + dzm * (id[3] - id[2]) + du * (id[5] - id[4]);
    const double dx3[3] = {dza, dzm, du};
    ASC_UNROLL
    for (int b = 0; b < 3; b++) {
      const double zl = it[(O_ZB + 2 * b) * K + k], zu = it[(O_ZB + 2 * b + 1) * K + k];
      const double dzl = id[2 * b] * (mu - zl * dx3[b]) - zl, dzu = id[2 * b + 1] * (mu + zu * dx3[b]) - zu;
      ASC_FTB(adu, zl, dzl);
      ASC_FTB(adu, zu, dzu);
      st[(O_ZB + 2 * b) * K + k] = dzl; st[(O_ZB + 2 * b + 1) * K + k] = dzu;
    }
    const double *rk = w + g.off_rec() + (size_t)k * NGRID * 64;
    ASC_UNROLL
    for (int q = 0; q < 7; q++) cl += rk[G_V * 64 + q] * (it[(O_L + q) * K + k] + x[8 + q]);
    if (DC) {       // the slack pair of the movement equation (as in d_newton): d delta from the equation itself
      const double cu = rk[G_V * 64 + 7], dlu = x[15];
      double dup = 0.0;
      if (k > 0) { const double *yp = Yp + (size_t)(k - 1) * BS * 3; dup = yp[7 * 3] - yp[7 * 3 + 1] * dth - yp[7 * 3 + 2] * dnu3; }
      const double ddel = x[7] - dup + cu;
      const double pp = it[O_PP * K + k], pn = it[O_PN * K + k], zp = it[O_ZP * K + k], zn = it[O_ZN * K + k], lu = it[O_LU * K + k];
      const double ip = rcp(pp), in_ = rcp(pn);
      const double sgp = zp * ip + dw, sgn_ = zn * in_ + dw;
      double dpp, dpn;
      if (sgp >= sgn_) { dpp = (dlu - (dcw - mu * ip - lu)) * rcp(sgp); dpn = dpp - ddel; }
      else { dpn = (-dlu - (dcw - mu * in_ + lu)) * rcp(sgn_); dpp = ddel + dpn; }
      const double dzp = ip * (mu - zp * dpp) - zp, dzn = in_ * (mu - zn * dpn) - zn;
      ASC_FTBR(rmax, ip, dpp); ASC_FTBR(rmax, in_, dpn);
      ASC_FTB(adu, zp, dzp); ASC_FTB(adu, zn, dzn);
      gsum -= dpp * ip + dpn * in_;
      gmove += dpp + dpn;
      dx2 += dpp * dpp + dpn * dpn;
      cl += cu * (lu + dlu);
      st[O_LU * K + k] = dlu; st[O_PP * K + k] = dpp; st[O_PN * K + k] = dpn; st[O_ZP * K + k] = dzp; st[O_ZN * K + k] = dzn;
    }
  }
  rmax = wmax(rmax); gsum = wsum(gsum); adu = wmin(adu); cl = wsum(cl); dx2 = wsum(dx2); gmove = wsum(gmove);
  // ---- scalars of the step, merit bookkeeping (as d_newton) ---------------------------------------------------------------
  const double is1 = rcp(s.s1), is2 = rcp(s.s2);
  const double sig1 = s.zs1 * is1 + dw, sig2 = s.zs2 * is2 + dw, rs1 = -mu * is1 - s.nu1, rs2 = -mu * is2 - s.nu2;
  double dzKx = 0.0, dzKy = 0.0, dzKvx = 0.0, dzKvy = 0.0;
  {
    const double *y = Yp + (size_t)(K - 1) * BS * 3;
    dzKx = y[IX * 3] - y[IX * 3 + 1] * dth - y[IX * 3 + 2] * dnu3; dzKy = y[IY * 3] - y[IY * 3 + 1] * dth - y[IY * 3 + 2] * dnu3;
    dzKvx = y[IVX * 3] - y[IVX * 3 + 1] * dth - y[IVX * 3 + 2] * dnu3; dzKvy = y[IVY * 3] - y[IVY * 3 + 1] * dth - y[IVY * 3 + 2] * dnu3;
  }
  Scal ds;
  ds.th = dth; ds.nu3 = dnu3;
  ds.s1 = (tm.g1 - s.s1) + tm.g1g[0] * dzKx + tm.g1g[1] * dzKy;
  ds.s2 = (tm.g2 - s.s2) + tm.g2g[0] * dzKvx + tm.g2g[1] * dzKvy;
  if (d.term == 2) {
    ds.s1 += tm.g1v[0] * dzKvx + tm.g1v[1] * dzKvy;
    ds.s2 += tm.g2p[0] * dzKx + tm.g2p[1] * dzKy;
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
  double gd = mu * gsum + dcw * gmove;
  gd += ds.th * (1.0 - mu / dl_ + mu / dU) - mu * ds.s1 / s.s1 - mu * ds.s2 / s.s2;
  cl += tm.e3 * (s.nu3 + ds.nu3) + (tm.g1 - s.s1) * (s.nu1 + ds.nu1) + (tm.g2 - s.s2) * (s.nu2 + ds.nu2);
  const double curv = -gd + cl;                 // dx'(W + Sigma + delta_w)dx by the Newton identity
  dx2 += dth * dth + ds.s1 * ds.s1 + ds.s2 * ds.s2;
  ok = ok && isfinite(curv) && curv >= 1e-11 * dx2;
  if (!ok) {                                    // regularise and solve again (this NLP only; state stays ST_NEWTON)
    if (l == 0) {
      if (probe) { sc[X_STATE] = ST_DONE; sc[X_STATUS] = ASCENT_REGULARISATION_FAILED; return; }
      const double ndw = next_delta_w(dw, sc[X_DWL]);
      if (ndw > 1e10) { sc[X_STATUS] = ASCENT_REGULARISATION_FAILED; sc[X_STATE] = ST_DONE; }
      else { sc[X_DW] = ndw; sc[X_REFAC] += 1.0; atomicAdd(&counters[0], 1); }
    }
    return;
  }
  if (c1 > 0.0) {
    const double need = (gd + 0.5 * fmax(curv, 0.0)) / (0.9 * c1);
    if (nu_pen < need) nu_pen = need + 1.0;
  }
  if (l == 0) {
    store_scal(sc, X_D, ds);
    sc[X_NUP] = nu_pen; sc[X_DWL] = dw;
    sc[X_DM] = gd - nu_pen * c1;
    sc[X_PHI0] = (s.th + mv) - mu * sl + nu_pen * c1;
    sc[X_ALPHA] = apr; sc[X_ADU] = adu;
    if (probe) { sc[X_STATE] = ST_DONE; sc[X_STATUS] = 0; }
    else { sc[X_STATE] = ST_TRIAL; atomicAdd(&counters[0], 1); }
  }
}

// ==============================================================================================================
// d_finish: results in the external layouts.  Lane = (NLP, step).
// ==============================================================================================================
__global__ __launch_bounds__(WAVE) void d_finish(const ascent_params *params, long batch, DGeo g, double *ws, double *traj,
                                                 double *tf_out, int *status_out, int *iters_out, double *blob,
                                                 int step_instead) {
  const long p = blockIdx.y;
  const int k = blockIdx.x * WAVE + threadIdx.x, K = g.K, nt = K + 1;
  if (k >= K) return;
  double *w = ws + (size_t)p * g.nlp_doubles();
  double *sc = w + g.off_sc();
  const Der d = derive_t(params[p], g.terminal);
  const double *it = step_instead ? w + g.off_st() : w + g.off_it((int)sc[X_CUR]);
  if (k == 0) {
    const Scal s = load_scal(sc, step_instead ? X_D : X_S);
    if (tf_out) tf_out[p] = s.th;
    if (status_out) status_out[p] = (int)sc[X_STATUS];
    if (iters_out) iters_out[p] = (int)sc[X_ITERS];
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
  double z[7];
  ASC_UNROLL
  for (int q = 0; q < 7; q++) z[q] = it[(O_Z + q) * K + k];
  const double u = it[O_U * K + k];
  if (blob) {
    ASC_UNROLL
    for (int q = 0; q < 7; q++) {
      blob[(7L * k + q) * batch + p] = z[q];
      blob[(8L * K + 7L * k + q) * batch + p] = it[(O_L + q) * K + k];
    }
    blob[(7L * K + k) * batch + p] = u;
    ASC_UNROLL
    for (int b = 0; b < 6; b++) blob[(15L * K + 6L * k + b) * batch + p] = it[(O_ZB + b) * K + k];
  }
  if (traj) {
    double ax, ay;
    accel<0>(d, z[IX], z[IY], z[IA], z[IM], 0.0, 0.0, ax, ay, nullptr, nullptr);
    const double v[10] = {z[IX], z[IY], z[IVX], z[IVY], ax, ay, z[IA], z[IW], u, z[IM]};
    ASC_UNROLL
    for (int f = 0; f < 10; f++) traj[((long)f * nt + k + 1) * batch + p] = v[f];
  }
}

// parity surface: the stage record of every step in a flat layout, [batch][K][6][64]
__global__ __launch_bounds__(WAVE) void d_dump_records(long batch, DGeo g, const double *ws, double *out) {
  const long p = blockIdx.y;
  const int k = blockIdx.x;
  const double *w = ws + (size_t)p * g.nlp_doubles();
  const double *rec = w + g.off_rec() + (size_t)k * NGRID * 64;
  double *o = out + ((size_t)p * g.K + k) * NGRID * 64;
  for (int r = 0; r < NGRID; r++) o[r * 64 + threadIdx.x] = rec[r * 64 + threadIdx.x];
}

// ==============================================================================================================
// Coast arc (the second phase of BASELINE config 5): Kepler-exact two-body propagation of every NLP's burnout state
// to the next apoapsis of its orbit, sampled uniformly in time.  Lane = (NLP, node): every node is an independent
// solve of Kepler's equation (the explicit-Euler propagator of the reference's v1 script, PDF p28-29, stepped 6.6 million
// times serially for the same picture).
// ==============================================================================================================
__global__ __launch_bounds__(WAVE) void k_coast(const ascent_params *params, long batch, const double *state4, int nc,
                                                double *coast, double *theta2, double *apsides) {
  const long p = (long)blockIdx.x * WAVE + threadIdx.x;
  const int jn = blockIdx.y;                  // node 0..nc
  if (p >= batch) return;
  const ascent_params &prm = params[p];
  const double S = prm.r_peri, GM = prm.G * prm.M;
  const double X = state4[0 * batch + p] * S, Y = state4[1 * batch + p] * S + prm.R0;
  const double VX = state4[2 * batch + p] * S, VY = state4[3 * batch + p] * S;
  const double r = sqrt(X * X + Y * Y), v2 = VX * VX + VY * VY, rv = X * VX + Y * VY;
  const double a = 1.0 / (2.0 / r - v2 / GM);
  const double h = X * VY - Y * VX;                     // signed angular momentum (the ascent flies towards -x: h > 0)
  // eccentricity vector
  const double ex = (v2 / GM - 1.0 / r) * X - rv / GM * VX, ey = (v2 / GM - 1.0 / r) * Y - rv / GM * VY;
  const double e = sqrt(ex * ex + ey * ey);
  const double n = sqrt(GM / (a * a * a));
  // eccentric anomaly of the burnout state:  r = a(1 - e cos E),  r.v = sqrt(GM a) e sin E
  const double E0 = e > 1e-12 ? atan2(rv / sqrt(GM * a), 1.0 - r / a) : 0.0;
  const double M0 = E0 - e * sin(E0);
  double Mend = M_PI;                                   // apoapsis
  if (M0 > M_PI) Mend += 2.0 * M_PI;
  const double T2 = (Mend - M0) / n;
  const double M = M0 + (Mend - M0) * (double)jn / (double)nc;
  double E = M + e * sin(M);
  for (int it = 0; it < 12; it++) E -= (E - e * sin(E) - M) / (1.0 - e * cos(E));
  // perifocal frame: P along the eccentricity vector, Q = h x P / |h| (90 degrees ahead in the direction of motion)
  double px = 1.0, py = 0.0;
  if (e > 1e-12) { px = ex / e; py = ey / e; } else { px = X / r; py = Y / r; }
  const double sg = h >= 0.0 ? 1.0 : -1.0;
  const double qx = -sg * py, qy = sg * px;
  const double cE = cos(E), sE = sin(E), b = a * sqrt(fmax(0.0, 1.0 - e * e));
  const double xp = a * (cE - e), yp = b * sE;
  const double rr = a * (1.0 - e * cE);
  const double vxp = -sqrt(GM * a) / rr * sE, vyp = sqrt(GM * a) / rr * sqrt(fmax(0.0, 1.0 - e * e)) * cE;
  const double Xn = xp * px + yp * qx, Yn = xp * py + yp * qy, VXn = vxp * px + vyp * qx, VYn = vxp * py + vyp * qy;
  const long npts = nc + 1;
  coast[((long)0 * npts + jn) * batch + p] = Xn / S;
  coast[((long)1 * npts + jn) * batch + p] = (Yn - prm.R0) / S;
  coast[((long)2 * npts + jn) * batch + p] = VXn / S;
  coast[((long)3 * npts + jn) * batch + p] = VYn / S;
  if (jn == 0) {
    theta2[p] = T2 / prm.T_scale;
    apsides[0 * batch + p] = a * (1.0 - e) - prm.R0;
    apsides[1 * batch + p] = a * (1.0 + e) - prm.R0;
  }
}

}  // namespace

namespace ascent {

size_t dense_ws_bytes(int K, long batch) {
  DGeo g{K, 0, 0};
  return (size_t)batch * g.nlp_doubles() * sizeof(double) + 64;
}

// PCR variant: the dense workspace, then two block images, the PCR solution, the border rows, per-NLP flags
struct PcrWs { double *bt_a, *bt_b, *Y, *bcols; int *flags; };
static size_t pcr_extra_doubles(int K, long batch) {
  return (size_t)batch * K * (2 * blocktri_node_doubles() + PC_BSMAX * 3 + 32) + (size_t)batch + 16;
}
size_t dense_pcr_ws_bytes(int K, long batch) { return dense_ws_bytes(K, batch) + pcr_extra_doubles(K, batch) * sizeof(double); }
static PcrWs pcr_ws(double *ws, int K, long batch) {
  PcrWs q;
  double *base = (double *)((char *)ws + dense_ws_bytes(K, batch));
  const size_t nb = (size_t)batch * K * blocktri_node_doubles();
  q.bt_a = base; q.bt_b = base + nb; q.Y = base + 2 * nb; q.bcols = q.Y + (size_t)batch * K * PC_BSMAX * 3;
  q.flags = (int *)(q.bcols + (size_t)batch * K * 32);
  return q;
}

#define DCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { snprintf(err, errlen, "%s: %s", #call, hipGetErrorString(e_)); return ASCENT_E_HIP; } } while (0)

static int pcr_newton(const ascent_params *dp, long batch, DGeo g, double *ws, int probe, int *counters, hipStream_t stream,
                      char *err, size_t errlen) {
  const int K = g.K;
  const PcrWs q = pcr_ws(ws, K, batch);
  DCHK(hipMemsetAsync(q.flags, 0, (size_t)batch * sizeof(int), stream));
  if (g.dc) hipLaunchKernelGGL(pc_assemble<1>, dim3((unsigned)K, (unsigned)batch), dim3(WAVE), 0, stream, dp, batch, g, (const double *)ws, q.bt_a, q.bcols, blocktri_node_doubles());
  else hipLaunchKernelGGL(pc_assemble<0>, dim3((unsigned)K, (unsigned)batch), dim3(WAVE), 0, stream, dp, batch, g, (const double *)ws, q.bt_a, q.bcols, blocktri_node_doubles());
  DCHK(hipGetLastError());
  int rc = blocktri_pcr_assembled(batch, K, g.dc ? 16 : PC_BS, PC_NB, q.bt_a, q.bt_b, q.Y, q.flags, stream, err, errlen);
  if (rc) return rc;
  if (g.dc) hipLaunchKernelGGL(pc_step<1>, dim3((unsigned)batch), dim3(WAVE), 0, stream, dp, batch, g, ws, (const double *)q.Y, (const double *)q.bcols, (const int *)q.flags, probe, counters);
  else hipLaunchKernelGGL(pc_step<0>, dim3((unsigned)batch), dim3(WAVE), 0, stream, dp, batch, g, ws, (const double *)q.Y, (const double *)q.bcols, (const int *)q.flags, probe, counters);
  DCHK(hipGetLastError());
  return ASCENT_OK;
}

int dense_run(const ascent_params *dp, long batch, int K, int scheme, int terminal, double *ws, const double *dguess, int warm,
              int max_iter, double tol, double mu0, double *dtraj, double *dtf, int *dstatus, int *diters, double *dblob,
              hipStream_t stream, char *err, size_t errlen, int pcr, int move_penalty) {
  DGeo g{K, scheme, terminal, move_penalty ? 1 : 0};
  int *counters = (int *)((char *)ws + (size_t)batch * g.nlp_doubles() * sizeof(double));
  static int *host_cnt_dev[64] = {nullptr};
  int dev_ = 0;
  DCHK(hipGetDevice(&dev_));
  dev_ &= 63;
  if (!host_cnt_dev[dev_]) DCHK(hipHostMalloc((void **)&host_cnt_dev[dev_], 4 * sizeof(int)));
  int *host_cnt = host_cnt_dev[dev_];
  const dim3 ngrid((unsigned)((K + WAVE - 1) / WAVE), (unsigned)batch);
  hipLaunchKernelGGL(d_init, ngrid, dim3(WAVE), 0, stream, dp, batch, g, ws, dguess, warm, mu0, 0, (const double *)nullptr);
  DCHK(hipGetLastError());
  const int burst = 4;
  for (long round = 0;;) {
    if (round > 50L * (max_iter + 2)) { snprintf(err, errlen, "dense path did not terminate within %ld rounds", round); return ASCENT_E_NOTERM; }
    for (int r = 0; r < burst; r++, round++) {
      DCHK(hipMemsetAsync(counters, 0, sizeof(int), stream));
      hipLaunchKernelGGL(d_eval, ngrid, dim3(WAVE), 0, stream, dp, batch, g, ws);
      if (!pcr) {
        if (batch <= 1024)       // (at most one wavefront per SIMD: the pipelined variant)
          hipLaunchKernelGGL((d_newton<0, 1>), dim3((unsigned)batch), dim3(WAVE), 0, stream, dp, batch, g, ws, max_iter, tol, 0,
                             (const double *)nullptr, counters);
        else
          hipLaunchKernelGGL((d_newton<0, 0>), dim3((unsigned)batch), dim3(WAVE), 0, stream, dp, batch, g, ws, max_iter, tol, 0,
                             (const double *)nullptr, counters);
      } else {
        hipLaunchKernelGGL(d_newton<1>, dim3((unsigned)batch), dim3(WAVE), 0, stream, dp, batch, g, ws, max_iter, tol, 0,
                           (const double *)nullptr, counters);
        // (an NLP that goes on is counted by both kernels; the host only asks whether the counter is zero)
        const int rc = pcr_newton(dp, batch, g, ws, 0, counters, stream, err, errlen);
        if (rc) return rc;
      }
      DCHK(hipGetLastError());
    }
    DCHK(hipMemcpyAsync(host_cnt, counters, sizeof(int), hipMemcpyDeviceToHost, stream));
    DCHK(hipStreamSynchronize(stream));
    if (host_cnt[0] == 0) break;          // no NLP asked for another trial point in the last round
  }
  hipLaunchKernelGGL(d_finish, ngrid, dim3(WAVE), 0, stream, dp, batch, g, ws, dtraj, dtf, dstatus, diters, dblob, 0);
  DCHK(hipGetLastError());
  return ASCENT_OK;
}

int dense_probe(const ascent_params *dp, long batch, int K, int scheme, int terminal, double *ws, const double *diterate,
                const double *dmu, const double *ddw, bool step_too, double *dstep, int *dinertia, double *drecords,
                hipStream_t stream, char *err, size_t errlen, int pcr, int move_penalty) {
  DGeo g{K, scheme, terminal, move_penalty ? 1 : 0};
  int *counters = (int *)((char *)ws + (size_t)batch * g.nlp_doubles() * sizeof(double));
  const dim3 ngrid((unsigned)((K + WAVE - 1) / WAVE), (unsigned)batch);
  hipLaunchKernelGGL(d_init, ngrid, dim3(WAVE), 0, stream, dp, batch, g, ws, diterate, 2, 0.1, 1, dmu);
  hipLaunchKernelGGL(d_eval, ngrid, dim3(WAVE), 0, stream, dp, batch, g, ws);
  DCHK(hipGetLastError());
  if (drecords) {
    hipLaunchKernelGGL(d_dump_records, dim3((unsigned)K, (unsigned)batch), dim3(WAVE), 0, stream, batch, g, ws, drecords);
    DCHK(hipGetLastError());
  }
  if (step_too && pcr) {
    hipLaunchKernelGGL(d_newton<1>, dim3((unsigned)batch), dim3(WAVE), 0, stream, dp, batch, g, ws, 1 << 30, -1.0, 1, ddw, counters);
    const int rc = pcr_newton(dp, batch, g, ws, 1, counters, stream, err, errlen);
    if (rc) return rc;
    hipLaunchKernelGGL(d_finish, ngrid, dim3(WAVE), 0, stream, dp, batch, g, ws, (double *)nullptr, (double *)nullptr, dinertia,
                       (int *)nullptr, dstep, 1);
    DCHK(hipGetLastError());
  } else if (step_too) {
    hipLaunchKernelGGL(d_newton<0>, dim3((unsigned)batch), dim3(WAVE), 0, stream, dp, batch, g, ws, 1 << 30, -1.0, 1, ddw, counters);
    hipLaunchKernelGGL(d_finish, ngrid, dim3(WAVE), 0, stream, dp, batch, g, ws, (double *)nullptr, (double *)nullptr, dinertia,
                       (int *)nullptr, dstep, 1);
    DCHK(hipGetLastError());
  }
  return ASCENT_OK;
}

int coast_run(const ascent_params *dp, long batch, const double *dstate4, int nc, double *dcoast, double *dtheta2,
              double *dapsides, hipStream_t stream, char *err, size_t errlen) {
  hipLaunchKernelGGL(k_coast, dim3((unsigned)((batch + WAVE - 1) / WAVE), (unsigned)(nc + 1)), dim3(WAVE), 0, stream, dp, batch,
                     dstate4, nc, dcoast, dtheta2, dapsides);
  DCHK(hipGetLastError());
  return ASCENT_OK;
}

}  // namespace ascent
