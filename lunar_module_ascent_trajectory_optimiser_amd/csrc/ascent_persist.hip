// Persistent solver kernel of the batched ascent NLP solver for medium batches (gfx950): ONE launch per grid level, one
// wavefront owns four NLPs for the whole interior-point solve.
//
// The split pipeline (ascent_pipeline.hip) runs an interior-point iteration as five launches; the Jacobian / Hessian blocks
// of every collocation node (/root/reference/Launch_Optimiser.py:114-136) are materialised in HBM by one kernel and read
// back by four others (55 rows per node and iteration, 8.8x the algorithmic bytes at batch 4096), and the host steers
// the rounds.  Here a wavefront keeps its four NLPs from the initial point to convergence:
//   * the factorisation sweep is the 16-lanes-per-NLP sweep of the split pipeline (lanes 0-6 one column each of the 7x7
//     value function, lanes 7-9 the three right-hand sides; DPP row broadcasts, an LDS transpose per step); the forward and
//     adjoint sweeps are affine recursions whose node-local matrices the node-parallel phase forms: one row of a
//     matrix-vector product per lane and step;
//   * the node-parallel work -- trial point, defects, Jacobian / Hessian blocks, recursion matrices, merit and KKT-error
//     pieces, bound-multiplier steps, adjoint right-hand sides -- is done by the SAME wavefront, 64 lanes = 4 NLPs x 16
//     consecutive nodes, one 16-node chunk at a time, and handed to the sweep through LDS: the blocks of a chunk are
//     produced into LDS, consumed by the 16 serial steps of that chunk and overwritten by the next chunk.  They never reach
//     HBM; phases that need them again (forward, adjoint) evaluate them again;
//   * HBM holds, per NLP and node, the two iterate buffers, the step and the ten feedback gains of the factorisation
//     ([NLP][row][node], node contiguous: a chunk row of an NLP is one 128-byte line);
//   * line-search rejections, inertia corrections and the barrier schedule are per-NLP state in LDS; wavefronts do not
//     wait for one another and the host is not involved until the level is finished; the grid levels of the nested
//     iteration hand over inside this layout (p_transfer).
// Backward Euler (the reference's NODES=2; p_solve<0,0>), the trapezoid (<1,0>) and the v1 formulation (<0,1>).
#include <hip/hip_runtime.h>

// Floating-point contraction within a statement only (the front end's choice, the same in every inlined copy of a function).
// The kernel below evaluates the Jacobian blocks of a node again in every phase instead of storing them; with the default
// (contraction across statements, decided per copy by the back end) two copies of the SAME expression may round
// differently, the implicit block A of the factor phase is then not bit for bit the A of the forward and adjoint phases, and
// near the solution -- where the eliminated terminal slacks put sigma = z/s ~ 1e12 on the velocity rows -- that 1e-16 becomes
// 1e-8 in the multiplier steps: a floor of a few 1e-9 under the dual infeasibility (seen with the trapezoid: stragglers at
// tol 1e-9, failures at 1e-10).  Costs nothing measurable (4.19 ms either way for the config-3 batch).
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


// ==============================================================================================================
// p_init / p_transfer / p_finish: starting points and results.  Lane = (NLP, node) in p_init and p_transfer.
// ==============================================================================================================
// Move penalty: the slack pair of the movement equation u_k - u_{k-1} = p_k - n_k at a starting point -- around the guess's own
// movement (u, up: the pushed controls of nodes k and k-1), multipliers that zero the pair's stationarity rows
// (d_init of the dense-block path, solve_one of the C restatement)
ASC_DEV void start_move(double *w, int Kp, int k, double u, double up, bool warm, double dcw) {
  const double dl = u - up, eps = warm ? 1e-4 : 1e-2;
  w[(R_IT + O_LU) * Kp + k] = 0.0;
  w[(R_IT + O_PP) * Kp + k] = fmax(dl, 0.0) + eps;
  w[(R_IT + O_PN) * Kp + k] = fmax(-dl, 0.0) + eps;
  w[(R_IT + O_ZP) * Kp + k] = dcw;
  w[(R_IT + O_ZN) * Kp + k] = dcw;
}
// The reduced slack pair of a stage: with Sigma_p = z_p/p + dw, Sigma_n = z_n/n + dw and the stationarity residuals
// r_p = dcw - mu/p - lambda_u, r_n = dcw - mu/n + lambda_u the stage's control delta = p - n has curvature
// Rd = 1/(1/Sigma_p + 1/Sigma_n) and gradient gdl = Rd (r_p/Sigma_p - r_n/Sigma_n);  d lambda_u = Rd d delta + gdl.
// (One function for every phase that needs the pair: the same bits everywhere, see the note on contraction above.)
struct MovePivot { double ip, in_, sgp, sgn, rp, rn, Rd, gdl; };
ASC_DEV MovePivot move_pivot(double pp, double pn, double zp, double zn, double lu, double dcw, double mu, double dw) {
  MovePivot m;
  m.ip = rcp(pp); m.in_ = rcp(pn);
  m.sgp = zp * m.ip + dw; m.sgn = zn * m.in_ + dw;
  m.rp = dcw - mu * m.ip - lu; m.rn = dcw - mu * m.in_ + lu;
  const double isp = rcp(m.sgp), isn = rcp(m.sgn);
  m.Rd = rcp(isp + isn);
  m.gdl = m.Rd * (m.rp * isp - m.rn * isn);
  return m;
}

// The starting point at node kk: the built-in straight-line guess (warm == 0) or the caller's / the coarser grid's values in
// z, l, u, zb, pushed into the interior (warm 1: primal only, warm 2: primal-dual; a probe takes them as they are)
ASC_DEV void start_node(const Der &d, int K, int kk, int warm, bool probe, int form, double *z, double *l, double *zb, double &u) {
  if (!warm) {
    const double tf0 = 0.9, dr = 0.166, aend = 0.5, vp = sqrt(d.vp2), dt0 = (1.0 / K) * d.T * tf0;
    const double sdr = sin(dr), cdr = cos(dr);
    const double xf = -d.rhof * sdr, yf = d.rhof * cdr - d.rho0;
    const double fr = (double)(kk + 1) / K;
    z[IX] = fr * xf; z[IY] = fr * yf; z[IVX] = -fr * vp * cdr; z[IVY] = -fr * vp * sdr; z[IA] = fr * aend;
    z[IW] = aend / (K * dt0); z[IM] = d.mrate * dt0 * (kk + 1);
    u = 0.0;
    if (form == 1) {       // v1: the angle is the control: angle = (ub/2)(u+1), no angular rate
      z[IW] = 0.0;
      u = z[IA] / (0.5 * d.aub) - 1.0;
    }
  }
  if (!probe) {
    z[IA] = push_in(z[IA], 0.0, d.aub);
    z[IM] = push_in(z[IM], 0.0, 1.0);
    u = push_in(u, -1.0, 1.0);
  }
  ASC_UNROLL
  for (int b = 0; b < 6; b++) zb[b] = probe ? zb[b] : warm == 2 ? fmax(zb[b], 1e-12) : 1.0;
  ASC_UNROLL
  for (int i = 0; i < 7; i++) l[i] = warm == 2 ? l[i] : 0.0;
}
// ... and its scalars (s holds the guess if warm); zK = the last node of the starting point
ASC_DEV void start_scal(const Der &d, int warm, bool probe, const double *zK, Scal &s) {
  if (!warm) s.th = 0.9;
  if (!probe) s.th = push_in(s.th, d.tlb, d.tub);
  const Terminal tm = terminal_of(d, zK);
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
}
ASC_DEV void store_start(double *w, int Kp, int k, const double *z, const double *l, const double *zb, double u) {
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { w[(R_IT + O_Z + i) * Kp + k] = z[i]; w[(R_IT + O_L + i) * Kp + k] = l[i]; }
  w[(R_IT + O_U) * Kp + k] = u;
  ASC_UNROLL
  for (int b = 0; b < 6; b++) w[(R_IT + O_ZB + b) * Kp + k] = zb[b];
  // (the second iterate buffer, the step and the gains are written before they are read)
}

__global__ __launch_bounds__(WAVE) void p_init(const ascent_params *params, long batch, PGeo g, double *ws, const double *guess,
                                               int warm, double mu_init, const double *probe_mu, const double *probe_dw, int probe_kind) {
  const long p = blockIdx.y;
  const int k = blockIdx.x * WAVE + threadIdx.x, K = g.K, Kp = g.Kp;
  if (k >= Kp) return;
  double *w = ws + (size_t)p * g.nlp_doubles();
  const Der d = derive_t(params[p], g.term);
  const int asked_warm = warm;
  if (warm && !(guess[(21L * K + S_TH) * batch + p] > 0.0)) warm = 0;
  const bool probe = probe_mu != nullptr;          // the iterate is taken as it is
  double z[7], l[7], zb[6], u = 0.0;
  const int kk = k < K ? k : K - 1;                  // padding nodes replicate the last node (never read by anything that counts)
  if (warm) {
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { z[i] = guess[(7L * kk + i) * batch + p]; l[i] = guess[(8L * K + 7L * kk + i) * batch + p]; }
    u = guess[(7L * K + kk) * batch + p];
    ASC_UNROLL
    for (int b = 0; b < 6; b++) zb[b] = guess[(15L * K + 6L * kk + b) * batch + p];
  }
  start_node(d, K, kk, warm, probe, g.form, z, l, zb, u);
  store_start(w, Kp, k, z, l, zb, u);
  if (g.mp) {       // (v1: the MV is the angle = (aub/2)(u + 1), starts at 0 and carries DCOST itself)
    const double ha = 0.5 * d.aub, f0 = (double)kk / K * 0.5;             // (cold start: the straight-line guess of node kk-1)
    double up = kk == 0 ? (g.form == 1 ? -1.0 : 0.0) : warm ? guess[(7L * K + kk - 1) * batch + p] : (g.form == 1 ? f0 / ha - 1.0 : 0.0);
    if (kk > 0 && !probe) up = push_in(up, -1.0, 1.0);
    start_move(w, Kp, k, u, up, warm != 0, g.form == 1 ? params[p].dcost * ha : params[p].dcost);
  }
  if (k != K - 1) return;
  double *sc = w + (size_t)g.nrows() * Kp;
  Scal s;
  if (warm) {
    const double *gs = guess + (21L * K) * batch + p;
    s.th = gs[S_TH * batch]; s.zlt = gs[S_ZLT * batch]; s.zut = gs[S_ZUT * batch]; s.s1 = gs[S_S1 * batch];
    s.s2 = gs[S_S2 * batch]; s.zs1 = gs[S_ZS1 * batch]; s.zs2 = gs[S_ZS2 * batch]; s.nu3 = gs[S_NU3 * batch];
    s.nu1 = gs[S_NU1 * batch]; s.nu2 = gs[S_NU2 * batch];
  }
  start_scal(d, warm, probe, z, s);
  for (int r = 0; r < NSCAL; r++) sc[r] = 0.0;
  put_scal(sc, X_S, s);
  sc[X_STATE] = ST_TRIAL; sc[X_FIRST] = 1.0; sc[X_STATUS] = ASCENT_MAX_ITER;
  sc[X_MU] = (asked_warm && !warm) ? 0.1 : mu_init; sc[X_NUP] = 1.0;
  if (probe) { sc[X_MU] = probe_mu[p]; sc[X_PDW] = probe_dw[p]; sc[X_PROBE] = (double)probe_kind; }
}

// Nested iteration, from one grid to the next finer one without leaving the kernel's own layout: the converged primal-dual
// solution of the coarse grid (workspace wsc) is prolonged -- linear in tau; node 0 is the fixed initial state for the states
// and the first node for everything else; bound multipliers scale with the step; scalars are copied: the arithmetic of
// k_prolong in ascent_solver.hip -- and becomes the primal-dual warm start (mu) of the fine grid (workspace wsf).  A problem whose
// coarse solve did not converge starts cold.  The iterations spent so far travel along.
__global__ __launch_bounds__(WAVE) void p_transfer(const ascent_params *params, long batch, PGeo gc, const double *wsc, PGeo gf,
                                                   double *wsf, double mu) {
  const long p = blockIdx.y;
  const int k = blockIdx.x * WAVE + threadIdx.x, Kc = gc.K, Kf = gf.K, Kpc = gc.Kp, Kpf = gf.Kp;
  if (k >= Kpf) return;
  const double *wc = wsc + (size_t)p * gc.nlp_doubles(), *scc = wc + (size_t)gc.nrows() * Kpc;
  double *wf = wsf + (size_t)p * gf.nlp_doubles();
  const Der d = derive_t(params[p], gf.term);
  const int warm = (int)scc[X_STATUS] == ASCENT_CONVERGED ? 2 : 0;
  const double *ic = wc + (size_t)((int)scc[X_CUR] * gc.nit()) * Kpc;
  double z[7], l[7], zb[6], u = 0.0, up = 0.0;
  const int kk = k < Kf ? k : Kf - 1;
  if (warm && gf.mp && kk > 0) {      // the prolonged control of the node before (for the slack pair of this node's movement)
    const double x = (double)kk / (double)Kf * (double)Kc;
    int j = (int)x;
    if (j > Kc - 1) j = Kc - 1;
    const double wt = x - (double)j;
    const int ja = j ? j - 1 : 0;
    const double a = ic[O_U * Kpc + ja], b = ic[O_U * Kpc + j];
    up = push_in(fma(wt, b - a, a), -1.0, 1.0);
  }
  if (warm) {
    const double x = (double)(kk + 1) / (double)Kf * (double)Kc;
    int j = (int)x;
    if (j > Kc - 1) j = Kc - 1;
    const double wt = x - (double)j;
    const int ja = j ? j - 1 : 0;
    const double zsc = (double)Kc / (double)Kf;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      const double a = j ? ic[(O_Z + i) * Kpc + ja] : ((gf.form == 1 && i == IA) ? ic[(O_Z + i) * Kpc] : 0.0), b = ic[(O_Z + i) * Kpc + j];
      z[i] = fma(wt, b - a, a);
      const double la = ic[(O_L + i) * Kpc + ja], lb = ic[(O_L + i) * Kpc + j];
      l[i] = fma(wt, lb - la, la);
    }
    { const double a = ic[O_U * Kpc + ja], b = ic[O_U * Kpc + j]; u = fma(wt, b - a, a); }
    ASC_UNROLL
    for (int b6 = 0; b6 < 6; b6++) {
      const double a = ic[(O_ZB + b6) * Kpc + ja], b = ic[(O_ZB + b6) * Kpc + j];
      zb[b6] = fma(wt, b - a, a) * zsc;
    }
  }
  start_node(d, Kf, kk, warm, false, gf.form, z, l, zb, u);
  store_start(wf, Kpf, k, z, l, zb, u);
  if (gf.mp) {
    if (kk == 0) up = gf.form == 1 ? -1.0 : 0.0;
    else if (!warm && gf.form == 1) up = push_in((double)kk / Kf * 0.5 / (0.5 * d.aub) - 1.0, -1.0, 1.0);      // (cold: the guess of node kk-1)
    start_move(wf, Kpf, k, u, up, warm != 0, gf.form == 1 ? params[p].dcost * 0.5 * d.aub : params[p].dcost);
  }
  if (k != Kf - 1) return;
  double *sc = wf + (size_t)gf.nrows() * Kpf;
  Scal s;
  if (warm) s = lds_scal(scc, X_S);
  start_scal(d, warm, false, z, s);
  for (int r = 0; r < NSCAL; r++) sc[r] = 0.0;
  put_scal(sc, X_S, s);
  sc[X_STATE] = ST_TRIAL; sc[X_FIRST] = 1.0; sc[X_STATUS] = ASCENT_MAX_ITER;
  sc[X_MU] = warm ? mu : 0.1; sc[X_NUP] = 1.0;
  sc[X_ITB] = scc[X_ITB] + scc[X_ITERS];
}

// Results in the external layouts ([field][node][NLP], NLP contiguous).  Lane = NLP, a block = 64 NLPs x one 16-node chunk, one
// wavefront per FIN_SUB nodes of it: the loads gather one 8-byte word per NLP, but the 16 nodes of the chunk come out of the
// same 128-byte line (same block, same L1); the stores are full lines.
constexpr int FIN_WAVES = 8, FIN_SUB = CH / FIN_WAVES;
__global__ __launch_bounds__(WAVE * FIN_WAVES) void p_finish(const ascent_params *params, long batch, PGeo g, const double *ws, double *traj,
                                                 double *tf_out, int *status_out, int *iters_out, double *blob) {
  const long p = (long)blockIdx.y * WAVE + threadIdx.x;
  if (p >= batch) return;
  const int c = blockIdx.x, K = g.K, Kp = g.Kp, nt = K + 1;
  const double *w = ws + (size_t)p * g.nlp_doubles();
  const double *sc = w + (size_t)g.nrows() * Kp;
  const Der d = derive_t(params[p], g.term);
  const double *it = w + (size_t)((int)sc[X_CUR] * g.nit()) * Kp;
  if (c == 0 && threadIdx.y == 0) {
    const Scal s = lds_scal(sc, X_S);
    tf_out[p] = s.th;
    status_out[p] = (int)sc[X_STATUS];
    iters_out[p] = (int)sc[X_ITERS] + (int)sc[X_ITB];
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
  for (int k = c * CH + threadIdx.y * FIN_SUB; k < min(K, c * CH + (threadIdx.y + 1) * FIN_SUB); k++) {
    double z[7];
    ASC_UNROLL
    for (int q = 0; q < 7; q++) z[q] = it[(O_Z + q) * Kp + k];
    const double u = it[O_U * Kp + k];
    if (blob) {
      ASC_UNROLL
      for (int q = 0; q < 7; q++) {
        blob[(7L * k + q) * batch + p] = z[q];
        blob[(8L * K + 7L * k + q) * batch + p] = it[(O_L + q) * Kp + k];
      }
      blob[(7L * K + k) * batch + p] = u;
      ASC_UNROLL
      for (int b = 0; b < 6; b++) blob[(15L * K + 6L * k + b) * batch + p] = it[(O_ZB + b) * Kp + k];
    }
    if (traj) {
      double ax, ay;
      accel<0>(d, z[IX], z[IY], z[IA], z[IM], 0.0, 0.0, ax, ay, nullptr, nullptr);
      const double v[10] = {z[IX], z[IY], z[IVX], z[IVY], ax, ay, z[IA], z[IW], u, z[IM]};
      ASC_UNROLL
      for (int f = 0; f < 10; f++) traj[((long)f * nt + k + 1) * batch + p] = v[f];
    }
  }
}

// p_probe_out: the Newton step one probe round left behind, in the external blob layout; inertia[p] = 1: refused
__global__ __launch_bounds__(WAVE) void p_probe_out(long batch, PGeo g, const double *ws, double *step, int *inertia) {
  const long p = blockIdx.y;
  const int k = blockIdx.x * WAVE + threadIdx.x, K = g.K, Kp = g.Kp;
  if (k >= K) return;
  const double *w = ws + (size_t)p * g.nlp_doubles();
  const double *sc = w + (size_t)g.nrows() * Kp, *stp = w + (size_t)g.r_st() * Kp;
  const bool ok = (int)sc[X_STATE] == ST_TRIAL;
  if (k == 0) {
    inertia[p] = ok ? 0 : 1;
    for (int r = 0; r < 10; r++) step[(21L * K + r) * batch + p] = ok ? sc[X_D + r] : 0.0;
  }
  ASC_UNROLL
  for (int i = 0; i < 7; i++) {
    step[(7L * k + i) * batch + p] = ok ? stp[(O_Z + i) * Kp + k] : 0.0;
    step[(8L * K + 7L * k + i) * batch + p] = ok ? stp[(O_L + i) * Kp + k] : 0.0;
  }
  step[(7L * K + k) * batch + p] = ok ? stp[O_U * Kp + k] : 0.0;
  ASC_UNROLL
  for (int b = 0; b < 6; b++) step[(15L * K + 6L * k + b) * batch + p] = ok ? stp[(O_ZB + b) * Kp + k] : 0.0;
}

// p_probe_rows_out: the node rows one probe round (kind 2) dumped from LDS, in the layout of ascent_eval_nodes: defects
// [7K][batch], Jacobian blocks [8K][batch], Hessian blocks [10K][batch] -- the bound-barrier curvature that the factor phase
// folds into the (angle, angle) and (mass, mass) entries is taken out again (as q_probe_out does for the split pipeline)
__global__ __launch_bounds__(WAVE) void p_probe_rows_out(const ascent_params *params, long batch, PGeo g, const double *ws, double *defects,
                                                         double *jac, double *hess) {
  const long p = blockIdx.y;
  const int k = blockIdx.x * WAVE + threadIdx.x, K = g.K, Kp = g.Kp;
  if (k >= K) return;
  const double *w = ws + (size_t)p * g.nlp_doubles();
  const double *sc = w + (size_t)g.nrows() * Kp, *rows = w + (size_t)g.r_st() * Kp;
  const double *it = w + (size_t)((int)sc[X_CUR] * g.nit()) * Kp;
  const Der d = derive_t(params[p], g.term);
  const double a_ = it[(O_Z + IA) * Kp + k], m_ = it[(O_Z + IM) * Kp + k];
  const double siga = it[(O_ZB + 0) * Kp + k] * rcp(a_) + it[(O_ZB + 1) * Kp + k] * rcp(d.aub - a_);
  const double sigm = it[(O_ZB + 2) * Kp + k] * rcp(m_) + it[(O_ZB + 3) * Kp + k] * rcp(1.0 - m_);
  ASC_UNROLL
  for (int i = 0; i < 8; i++) jac[(8L * k + i) * batch + p] = rows[i * Kp + k];
  ASC_UNROLL
  for (int i = 0; i < 10; i++) hess[(10L * k + i) * batch + p] = rows[(8 + i) * Kp + k] - (i == 7 ? siga : i == 9 ? sigm : 0.0);
  ASC_UNROLL
  for (int i = 0; i < 7; i++) defects[(7L * k + i) * batch + p] = rows[(18 + i) * Kp + k];
}

// ==============================================================================================================
// p_solve: the whole interior-point loop of one grid level
// ==============================================================================================================

// Rows xdot and ydot of A^-1, A = I - dt df/dz (see solveA in ascent_device.hpp).  The other rows follow from them:
// row x = e_x + dt row xdot, row y = e_y + dt row ydot, row angle = e_angle + dt e_angledot; rows angledot and mass are unit vectors.
template <int FORM>
ASC_DEV void ainv_vrows(const double *G, const double *E, double dt, double *rvx, double *rvy) {
  double c1[7], c2[7];
  c1[IX] = dt * G[0]; c1[IY] = dt * G[1]; c1[IVX] = 1.0; c1[IVY] = 0.0; c1[IA] = dt * G[2]; c1[IW] = FORM == 1 ? 0.0 : dt * c1[IA]; c1[IM] = dt * G[3];
  c2[IX] = dt * G[4]; c2[IY] = dt * G[5]; c2[IVX] = 0.0; c2[IVY] = 1.0; c2[IA] = dt * G[6]; c2[IW] = FORM == 1 ? 0.0 : dt * c2[IA]; c2[IM] = dt * G[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { rvx[i] = E[0] * c1[i] + E[1] * c2[i]; rvy[i] = E[2] * c1[i] + E[3] * c2[i]; }
}


// The trial point x + alpha dx at node k (iterate n, step dn), stored into the other iterate buffer, and its pieces of the
// l1 merit function and of the KKT error (Launch_Optimiser.py:114-136 evaluated once, with first derivatives) -- in two parts:
// everything that depends on the primal variables and the bound multipliers only (trial_primal), and the rows that need the
// equality multipliers of the trial point (trial_dual).  The adjoint phase knows the primal step of a chunk before its sweep and
// the multiplier step after it: it runs the first part before the sweep and carries TrialKeep (21 values) across it instead of
// the node's whole iterate and step (70).  Every accumulator of Part sees the same operations in the same order as in one pass.
struct TrialKeep { double G[8], F[7], dza, dzm, zb4, zb5, zpp, zpn; };

template <int SCHEME, int FORM, int MP = 0, int TERM = 0>
ASC_DEV void trial_primal(const Der &d, int K, int Kp, int k, const NodeIn &n, const NodeIn &dn, const TrialCtx &t, bool live,
                          double *in, Part &P, TrialKeep &kp) {
  const double alpha = t.alpha;
  double z[7], zp[7], zb[6];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { z[i] = n.z[i] + alpha * dn.z[i]; zp[i] = n.zp[i] + alpha * dn.zp[i]; }
  const double u = n.u + alpha * dn.u;
  const double dist[6] = {z[IA], d.aub - z[IA], z[IM], 1.0 - z[IM], u + 1.0, 1.0 - u};
  ASC_UNROLL
  for (int b = 0; b < 6; b++) {
    const double id = rcp(dist[b]);
    zb[b] = t.first ? n.zb[b] : fmin(fmax(n.zb[b] + t.adu * dn.zb[b], t.mlo * id), t.mhi * id);
  }
  if (live) {
    ASC_UNROLL
    for (int i = 0; i < 7; i++) in[(O_Z + i) * Kp + k] = z[i];
    in[O_U * Kp + k] = u;
    ASC_UNROLL
    for (int b = 0; b < 6; b++) in[(O_ZB + b) * Kp + k] = zb[b];
  }
  kp.zpp = 0.0; kp.zpn = 0.0;
  if constexpr (MP) {      // the movement equation of the step, its slack pair and their rows of the KKT error
    const double up = n.up + alpha * dn.up, pp = n.pp + alpha * dn.pp, pn = n.pn + alpha * dn.pn;
    const double ip = rcp(pp), in_ = rcp(pn);
    const double zpp = t.first ? n.zpp : fmin(fmax(n.zpp + t.adu * dn.zpp, t.mlo * ip), t.mhi * ip);
    const double zpn = t.first ? n.zpn : fmin(fmax(n.zpn + t.adu * dn.zpn, t.mlo * in_), t.mhi * in_);
    if (live) { in[O_PP * Kp + k] = pp; in[O_PN * Kp + k] = pn; in[O_ZP * Kp + k] = zpp; in[O_ZN * Kp + k] = zpn; }
    const double cu = u - up - pp + pn;
    P.c1 += fabs(cu);
    P.cinf = fmax(P.cinf, fabs(cu));
    const double prp = pp * zpp, prn = pn * zpn;
    P.pmin = fmin(P.pmin, fmin(prp, prn)); P.pmax = fmax(P.pmax, fmax(prp, prn));
    P.zsum += zpp + zpn;
    P.mv += pp + pn;
    const double ps = pp * pn;
    P.sl += ps > 0.0 ? log(ps) : NAN;
    kp.zpp = zpp; kp.zpn = zpn;
  }
  double ax, ay;
  accel<1>(d, z[IX], z[IY], z[IA], z[IM], 0.0, 0.0, ax, ay, kp.G, nullptr);
  rhs_f<FORM>(d, z, u, ax, ay, kp.F);
  if (SCHEME == 1) {
    double Fb[7], axp, ayp;
    accel<0>(d, zp[IX], zp[IY], zp[IA], zp[IM], 0.0, 0.0, axp, ayp, nullptr, nullptr);
    rhs_f<FORM>(d, zp, u, axp, ayp, Fb);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) kp.F[i] = 0.5 * (kp.F[i] + Fb[i]);
  }
  ASC_UNROLL
  for (int i = 0; i < 7; i++) {
    const double cc = (FORM == 1 && i == IA) ? z[IA] - 0.5 * d.aub * (u + 1.0) : z[i] - zp[i] - t.dt * kp.F[i];
    P.c1 += fabs(cc);
    P.cinf = fmax(P.cinf, fabs(cc));
  }
  if (k == K - 1) {
    const Scal &stt = t.stt;
    const Terminal tt = TERM == 2 ? terminal_eval_any(d, z) : terminal_eval(d, z);
    const double e1 = fabs(tt.e3), e2 = fabs(tt.g1 - stt.s1), e3 = fabs(tt.g2 - stt.s2);
    P.cinf = fmax(P.cinf, fmax(e1, fmax(e2, e3)));
    P.c1 += e1 + e2 + e3;
    const double ps = ((stt.th - d.tlb) * (d.tub - stt.th)) * (stt.s1 * stt.s2);
    P.sl += ps > 0.0 ? log(ps) : NAN;
  }
  ASC_UNROLL
  for (int b = 0; b < 6; b++) { const double pr = dist[b] * zb[b]; P.pmin = fmin(P.pmin, pr); P.pmax = fmax(P.pmax, pr); P.zsum += zb[b]; }
  const double pa = dist[0] * dist[1], pm = dist[2] * dist[3], pu = dist[4] * dist[5];
  P.sl += (pa > 0.0 && pm > 0.0 && pu > 0.0) ? log(pa * pm * pu) : NAN;
  kp.dza = zb[1] - zb[0]; kp.dzm = zb[3] - zb[2]; kp.zb4 = zb[4]; kp.zb5 = zb[5];
}

// ... and the rows with the trial multipliers l (node k), ln (node k+1), lu / lun (move penalty: of the movement equations)
template <int SCHEME, int FORM, int MP = 0, int TERM = 0>
ASC_DEV void trial_dual(const Der &d, int K, int Kp, int k, const TrialKeep &kp, const double *l, const double *ln, double lu, double lun,
                        const TrialCtx &t, bool live, double *in, Part &P) {
  if (live) {
    ASC_UNROLL
    for (int i = 0; i < 7; i++) in[(O_L + i) * Kp + k] = l[i];
    if constexpr (MP) in[O_LU * Kp + k] = lu;
  }
  if constexpr (MP) {
    P.l1 += fabs(lu);
    P.rd = fmax(P.rd, fmax(fabs(t.dcw - lu - kp.zpp), fabs(t.dcw + lu - kp.zpn)));
  }
  // (scheme 1, the trapezoid with the control held over the step: defect z_k - z_{k-1} - dt/2 [f(z_k,u_k) + f(z_{k-1},u_k)]; node k then
  //  carries the multipliers of steps k and k+1 in its stationarity row, each with half the step)
  const double cs = SCHEME == 1 ? 0.5 * t.dt : t.dt;
  double fl[7], lt[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) lt[i] = SCHEME == 1 ? l[i] + ln[i] : l[i];
  fzt_lambda<FORM>(kp.G, lt, fl);
  double r[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) {
    r[i] = (FORM == 1 && i == IA) ? l[i] - cs * fl[i] : l[i] - cs * fl[i] - ln[i];
    P.rth -= t.hT * kp.F[i] * l[i];
    P.l1 += fabs(l[i]);
  }
  r[IA] += kp.dza;
  r[IM] += kp.dzm;
  if (k == K - 1) {       // (the last node's trial state comes back from the buffer this lane has just written)
    const Scal &stt = t.stt;
    double z[7];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) z[i] = in[(O_Z + i) * Kp + k];
    if constexpr (TERM == 2) {
      const Terminal tt = terminal_eval_any(d, z);
      double g4[4];
      terminal_grad_any(tt, stt.nu1, stt.nu2, g4);
      r[IX] += g4[0]; r[IY] += g4[1]; r[IVX] += g4[2]; r[IVY] += g4[3];
    } else {
      const Terminal tt = terminal_eval(d, z);
      r[IX] += stt.nu3 * tt.e3g[0] + stt.nu1 * tt.g1g[0];
      r[IY] += stt.nu3 * tt.e3g[1] + stt.nu1 * tt.g1g[1];
      r[IVX] += stt.nu3 * tt.e3g[2] + stt.nu2 * tt.g2g[0];
      r[IVY] += stt.nu3 * tt.e3g[3] + stt.nu2 * tt.g2g[1];
    }
  }
  ASC_UNROLL
  for (int i = 0; i < 7; i++) P.rd = fmax(P.rd, fabs(r[i]));
  double ruv = (FORM == 1 ? -0.5 * d.aub * l[IA] : -t.be * l[IW]) - kp.zb4 + kp.zb5;
  if constexpr (MP) ruv += lu - lun;
  P.rd = fmax(P.rd, fabs(ruv));
}

// both parts in one go (line-search retries and the first point of a level)
template <int SCHEME, int FORM, int MP = 0, int TERM = 0>
ASC_DEV void trial_node(const Der &d, int K, int Kp, int k, const NodeIn &n, const NodeIn &dn, const TrialCtx &t, bool live,
                        double *in, Part &P) {
  TrialKeep kp;
  trial_primal<SCHEME, FORM, MP, TERM>(d, K, Kp, k, n, dn, t, live, in, P, kp);
  double l[7], ln[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { l[i] = n.l[i] + t.alpha * dn.l[i]; ln[i] = n.ln[i] + t.alpha * dn.ln[i]; }
  const double lu = MP ? n.lu + t.alpha * dn.lu : 0.0, lun = MP ? n.lun + t.alpha * dn.lun : 0.0;
  trial_dual<SCHEME, FORM, MP, TERM>(d, K, Kp, k, kp, l, ln, lu, lun, t, live, in, P);
}


template <int SCHEME, int FORM, int MP = 0, int TERM = 0, int WIDE = 0>
__global__ __launch_bounds__(WAVE) void p_solve(const ascent_params *params, long batch, PGeo g, double *ws, int max_iter, double tol) {
  using L = Lay<MP>;
  constexpr int NS = L::NS, NIT = L::NIT, R_ST = L::R_ST, R_KA = L::R_KA, R_K0 = L::R_K0, NROWS = L::NROWS;
  constexpr int S_G = L::S_G, S_E = L::S_E, S_H = L::S_H, S_F = L::S_F, S_C = L::S_C, S_RZ = L::S_RZ, S_GT = L::S_GT, S_SC = L::S_SC;
  constexpr int RL = NS;                                // the first of the three right-hand-side lanes of the factorisation sweep
  static_assert(FORM == 0 || SCHEME == 0, "the v1 formulation is restated with backward Euler");
  __shared__ double stage[L::S_ROWS * LDW];
  __shared__ double outb[L::OUT_ROWS * LDW];
  __shared__ double lds_t[NPW][NS][NS];
  __shared__ double lds_d[NPW][3][8];                   // (row 2 stays zero)
  __shared__ double lsc[NPW][NSCAL];
  __shared__ double lds_c[NPW][8];                      // adjoint phase: the multiplier step of the first node of the chunk above
  // WIDE = 1 (batches that leave SIMDs idle: one NLP per wavefront): the node-parallel phases take 64 nodes at a time on the 64
  // lanes instead of 4 NLPs x 16 nodes; the four 16-lane groups then run the SAME serial sweeps side by side (identical values,
  // identical stores) over the 64 steps of a chunk, and sums over the nodes run over the whole wavefront.
  constexpr int CHN = WIDE ? 64 : CH;                   // nodes per chunk
  const int lane = threadIdx.x, grp = lane >> 4, role = lane & 15;
  const int nl = WIDE ? lane : role;                    // this lane's node within a chunk
  const int cbase = WIDE ? 0 : grp * 16;                // first column of this lane's NLP in the LDS stage
  const long p = WIDE ? (long)blockIdx.x : (long)blockIdx.x * NPW + grp;
  const bool live = p < batch;
  const long pc = live ? p : batch - 1;                 // dead groups shadow the last NLP and never store
  const int K = g.K, Kp = g.Kp, nch = g.nch;
  double *w = ws + (size_t)pc * g.nlp_doubles();
  double *gsc = w + (size_t)NROWS * Kp;
  double *sc = lsc[WIDE ? 0 : grp];
  const Der d = TERM == 2 ? derive_t(params[pc], 2) : derive(params[pc]);      // (TERM = 2: burnout anywhere on the (r_peri, r_apo) ellipse)
  static_assert(TERM == 0 || (TERM == 2 && FORM == 0), "terminal 2 is carried for the current formulation");
  for (int r = role; r < NSCAL; r += 16) sc[r] = gsc[r];
  if (role < 8) lds_d[grp][2][role] = 0.0;
  wsync();
  if (!live && role == 0) sc[X_STATE] = ST_DONE;
  wsync();
  const int col = grp * 16 + role;                      // this lane's column of the LDS stage in node-parallel phases
  const double hT = (1.0 / K) * d.T;
  constexpr int IB = FORM == 1 ? IA : IW;          // the defect row the control enters (v1: the algebraic angle row)
  // weight of the l1 move penalty on u and the control before node 0 (v1: the MV is the angle = (aub/2)(u + 1), DCOST on the angle, angle_0 = 0)
  const double dcw = MP ? (FORM == 1 ? params[pc].dcost * 0.5 * d.aub : params[pc].dcost) : 0.0;
  constexpr double UINIT = FORM == 1 ? -1.0 : 0.0;

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
      const double dt = hT * stt.th, be = dt * d.alpha;
      const double mlo = mu * 1e-10, mhi = mu * 1e10;
      Part P;
      if (sc[X_TEVAL] != 0.0) {             // evaluated by the adjoint phase of the previous round
        P.rd = sc[X_P + 0]; P.cinf = sc[X_P + 1]; P.pmin = sc[X_P + 2]; P.pmax = sc[X_P + 3]; P.l1 = sc[X_P + 4];
        P.zsum = sc[X_P + 5]; P.rth = sc[X_P + 6]; P.c1 = sc[X_P + 7]; P.sl = sc[X_P + 8]; P.mv = sc[X_P + 9];
      } else {
        TrialCtx t;
        t.alpha = alpha; t.adu = adu; t.mlo = mlo; t.mhi = mhi; t.dt = dt; t.be = be; t.hT = hT; t.first = first; t.stt = stt; t.dcw = dcw;
        P.clear();
        for (int c = 0; c < nch; c++) {
          const int k = c * CHN + nl;
          if (k < K) {
            NodeIn n, dn;
            load_node<MP>(ic, Kp, K, k, n, UINIT);
            if (first) dn = NodeIn{};             // (no step yet; the step rows are not initialised)
            else load_node<MP>(stp, Kp, K, k, dn);
            trial_node<SCHEME, FORM, MP, TERM>(d, K, Kp, k, n, dn, t, live, in, P);
          }
        }
        P.template reduceW<MP, WIDE>();
      }
      double rd = P.rd, cinf = P.cinf, pmin = P.pmin, pmax = P.pmax, l1 = P.l1, zsum = P.zsum;
      const double rth = 1.0 + P.rth, c1 = P.c1, sl = P.sl;
      // ---- decisions (all 16 lanes of the NLP alike; lane 0 writes) -------------------------------------------------
      double nu_pen = sc[X_NUP], iters = sc[X_ITERS], mu2 = mu;
      int nstate = ST_FACTOR;
      bool accepted = true;
      if (!first) {
        const double phi0 = sc[X_PHI0], Dm = sc[X_DM];
        const double phit = (MP ? stt.th + dcw * P.mv : stt.th) - mu * sl + nu_pen * c1;
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
        e.sd = fmax(100.0, (l1 + zsum) / (double)((MP ? 16 : 13) * K + 7)) * 0.01;
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
#ifdef PERSIST_TRACE      // diagnostic build: the iteration history of one NLP (scripts/persist_trace.py)
        if (role == 0 && p == PERSIST_TRACE && K > 100)
          printf("[persist] K=%d iter %2d mu %.1e E0 %.2e (dual %.1e primal %.1e compl %.1e..%.1e s_d %.2g) alpha %.3g adu %.3g ls %d dw %.1e nu_pen %.2g c1 %.2e\n", K, (int)iters, mu,
                 e.err(0.0), e.rd, e.cinf, e.pmin, e.pmax, e.sd, alpha, adu, (int)sc[X_LS], sc[X_DWL], nu_pen, c1);
        if (role == 0 && p == PERSIST_TRACE && K > 100)
          printf("[persist]      dual rows: nodes %.2e | th %.2e s1 %.2e s2 %.2e\n", rd, fabs(rth - stt.zlt + stt.zut), fabs(-stt.nu1 - stt.zs1), fabs(-stt.nu2 - stt.zs2));
#endif
        wsync();
        if (role == 0) {
          put_scal(sc, X_S, stt);
          sc[X_CUR] = 1 - cur; sc[X_FIRST] = 0.0; sc[X_ITERS] = iters; sc[X_LS] = 0.0; sc[X_C1] = c1; sc[X_SL] = sl; sc[X_RTH] = rth;
          sc[X_MU] = mu2; sc[X_NUP] = nu_pen; sc[X_DW] = probe ? sc[X_PDW] : 0.0; sc[X_STATE] = nstate; sc[X_TEVAL] = 0.0;
          if (MP) sc[X_MV] = P.mv;
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
      const double dt = hT * s.th, be = dt * d.alpha, ith = rcp(s.th), cs = SCHEME == 1 ? 0.5 * dt : dt;
      const double ha = 0.5 * d.aub, bu = FORM == 1 ? ha : be;          // (v1: angle = ha (u + 1), the control enters the angle row)
      const double hTc = SCHEME == 1 ? 0.5 * hT : hT;
      // what this lane gathers from a step's blocks for row i of its vector (factor phase), as in q_factor_wide
      // (MP: eight column lanes -- the control is the eighth state, its bound terms S_SC+0 its diagonal entry -- and the
      //  right-hand-side lanes 8-10; the stage's scalar control is delta = p - n with the reduced pair's curvature and gradient)
      int grow[NS];
      double gsgn[NS];
      {
        constexpr int hmap[8] = {0, 1, -1, -1, 2, -1, 3, -1};
        constexpr int hrow[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};
        ASC_UNROLL
        for (int i = 0; i < NS; i++) {
          int row = S_H; double sgn = 0.0;
          if (role < 7) {
            ASC_UNROLL
            for (int c = 0; c < 7; c++)
              if (c == role && hmap[i] >= 0 && hmap[c] >= 0) { row = S_H + hrow[hmap[i]][hmap[c]]; sgn = 1.0; }
          } else if (MP && role == 7) { if (i == 7) { row = S_SC; sgn = 1.0; } }
          else if (role == RL) { row = S_RZ + i; sgn = -1.0; }
          else if (role == RL + 1) { row = S_GT + i; sgn = -1.0; }
          grow[i] = row * LDW;
          gsgn[i] = sgn;
        }
      }
      const double bsc = role == RL ? -mu : 0.0;
      const int rowA = (role < 8 ? S_G + role : role < 12 ? S_E + role - 8
                        : MP ? (role == 12 ? S_SC + 1 : role == 13 ? S_SC + 2 : S_SC)
                        : (role == 12 ? S_SC : role == 13 ? S_SC + 1 : role == 14 ? S_SC + 4 : S_SC + 2)) * LDW;
      const int rowB = (role < 7 ? S_C + role
                        : MP ? (role < 13 ? S_F + role - 7 : role == 15 ? S_C + 7 : S_SC)
                        : (role < 14 ? S_F + role - 7 : role == 14 ? S_SC + 3 : S_SC)) * LDW;
      const int rowK = (role < NS + 3 ? role : NS + 3) * LDW;      // out rows 0 .. NS-1 kap, NS .. NS+2 k0, NS+3 dummy / pivot
      const bool colr = role < NS;
      const int drow = role == RL ? 0 : role == RL + 1 ? 1 : 2;           // what a lane subtracts after the pivot: P c, P rc or nothing
      double a[NS];
      ASC_UNROLL
      for (int i = 0; i < NS; i++) a[i] = 0.0;
      double U = 0.0, V = 0.0, k10 = 0.0, k11 = 0.0, k12 = 0.0, k20 = 0.0, k22 = 0.0;
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
          if (i < 4) { v = role == RL ? -r0[i] : v; v = role == RL + 2 ? -tm.e3g[i] : v; }
          a[i] = v;
        }
      }
      const bool probe_rows = sc[X_PROBE] == 2.0;      // parity probe of the node evaluation: dump the stage rows, no sweep
      for (int c = nch - 1; c >= 0; c--) {
        // ---- node-parallel: the blocks of the 16 nodes of this chunk -------------------------------------------------
        {
          const int k = c * CHN + nl;
          if (k < K && act) {
            NodeIn n;
            load_node<MP>(it, Kp, K, k, n, UINIT);
            double G[8], E[4], H[10], F[7], fl[7], lt[7], ax, ay;
            ASC_UNROLL
            for (int i = 0; i < 7; i++) lt[i] = SCHEME == 1 ? n.l[i] + n.ln[i] : n.l[i];
            accel<2>(d, n.z[IX], n.z[IY], n.z[IA], n.z[IM], -cs * lt[IVX], -cs * lt[IVY], ax, ay, G, H);
            rhs_f<FORM>(d, n.z, n.u, ax, ay, F);
            if (SCHEME == 1) {       // second evaluation point of the step: f(z_{k-1}, u_k)
              double Fb[7], axp, ayp;
              accel<0>(d, n.zp[IX], n.zp[IY], n.zp[IA], n.zp[IM], 0.0, 0.0, axp, ayp, nullptr, nullptr);
              rhs_f<FORM>(d, n.zp, n.u, axp, ayp, Fb);
              ASC_UNROLL
              for (int i = 0; i < 7; i++) F[i] = 0.5 * (F[i] + Fb[i]);
            }
            implicit_block(G, cs, E);
            fzt_lambda<FORM>(G, lt, fl);
            const double dist[6] = {n.z[IA], d.aub - n.z[IA], n.z[IM], 1.0 - n.z[IM], n.u + 1.0, 1.0 - n.u};
            double id[6];
            ASC_UNROLL
            for (int b = 0; b < 6; b++) id[b] = rcp(dist[b]);
            H[7] += n.zb[0] * id[0] + n.zb[1] * id[1];
            H[9] += n.zb[2] * id[2] + n.zb[3] * id[3];
            ASC_UNROLL
            for (int i = 0; i < 8; i++) stage[(S_G + i) * LDW + col] = G[i];
            ASC_UNROLL
            for (int i = 0; i < 4; i++) stage[(S_E + i) * LDW + col] = E[i];
            ASC_UNROLL
            for (int i = 0; i < 10; i++) stage[(S_H + i) * LDW + col] = H[i];
            ASC_UNROLL
            for (int i = 0; i < 7; i++) {
              if (!MP || i != IM) stage[(S_F + i) * LDW + col] = hT * F[i];            // (pre-scaled: the sweep uses hT F only)
              stage[(S_C + i) * LDW + col] = (FORM == 1 && i == IA) ? n.z[IA] - ha * (n.u + 1.0) : n.z[i] - n.zp[i] - dt * F[i];
              stage[(S_RZ + i) * LDW + col] = (FORM == 1 && i == IA) ? n.l[i] - cs * fl[i] : n.l[i] - cs * fl[i] - n.ln[i];
              stage[(S_GT + i) * LDW + col] = -hTc * fl[i];
            }
            const double scr[5] = {n.zb[4] * id[4] + n.zb[5] * id[5], FORM == 1 ? -ha * n.l[IA] : -be * n.l[IW], id[1] - id[0], id[3] - id[2],
                                   id[5] - id[4]};
            if constexpr (MP) {      // row 7 of the stage: the movement equation, the control's stationarity row, the reduced slack pair
              const MovePivot mvp = move_pivot(n.pp, n.pn, n.zpp, n.zpn, n.lu, dcw, mu, dw);
              stage[(S_C + 7) * LDW + col] = n.u - n.up - n.pp + n.pn;
              stage[(S_RZ + 7) * LDW + col] = (scr[1] + (n.lu - n.lun)) + mu * scr[4];
              stage[(S_GT + 7) * LDW + col] = FORM == 1 ? 0.0 : -hT * d.alpha * n.l[IW];
              stage[S_SC * LDW + col] = scr[0];
              stage[(S_SC + 1) * LDW + col] = mvp.Rd;
              stage[(S_SC + 2) * LDW + col] = mvp.gdl;
              stage[(S_RZ + IA) * LDW + col] += mu * scr[2];       // (the barrier gradients folded into the residual rows)
              stage[(S_RZ + IM) * LDW + col] += mu * scr[3];
            } else {
              ASC_UNROLL
              for (int i = 0; i < 5; i++) stage[(S_SC + i) * LDW + col] = scr[i];
            }
          }
        }
        wsync();
        PROF(1);
        if (probe_rows) {      // the rows the sweep would gather -- Jacobian block, Hessian block (bound terms included), defects -- as they
          const int k = c * CHN + nl;                    // stand in LDS, into the step / gain rows of the workspace (p_probe_rows_out)
          if (k < K && act && live) {
            ASC_UNROLL
            for (int i = 0; i < 8; i++) w[(size_t)(R_ST + i) * Kp + k] = stage[(S_G + i) * LDW + col];
            ASC_UNROLL
            for (int i = 0; i < 10; i++) w[(size_t)(R_ST + 8 + i) * Kp + k] = stage[(S_H + i) * LDW + col];
            ASC_UNROLL
            for (int i = 0; i < 7; i++) w[(size_t)(R_ST + 18 + i) * Kp + k] = stage[(S_C + i) * LDW + col];
          }
          wsync();
          continue;
        }
        // ---- serial: the 16 steps of the chunk, backwards; 16 lanes per NLP (the arithmetic of q_factor_wide) -----------
        if (act) {
          for (int jj = CHN - 1; jj >= 0; jj--) {
            const int k = c * CHN + jj;
            if (k >= K) continue;
            const int cj = cbase + jj;
            double gq[NS];
            ASC_UNROLL
            for (int i = 0; i < NS; i++) gq[i] = stage[grow[i] + cj];
            const double gA = stage[rowA + cj], gB = stage[rowB + cj];
            const double G[8] = {bcast16<0>(gA), bcast16<1>(gA), bcast16<2>(gA), bcast16<3>(gA),
                                 bcast16<4>(gA), bcast16<5>(gA), bcast16<6>(gA), bcast16<7>(gA)};
            const double E[4] = {bcast16<8>(gA), bcast16<9>(gA), bcast16<10>(gA), bcast16<11>(gA)};
            const double R0 = bcast16<12>(gA), ru0 = bcast16<13>(gA), bur = bcast16<14>(gA);      // (MP: R0, ru0 = curvature and gradient of the reduced slack pair;
            const double bza = bcast16<15>(gA), bzm = bcast16<14>(gB);                            //  bur, bza, bzm are not used)
            const double cc[8] = {bcast16<0>(gB), bcast16<1>(gB), bcast16<2>(gB), bcast16<3>(gB),
                                  bcast16<4>(gB), bcast16<5>(gB), bcast16<6>(gB), MP ? bcast16<15>(gB) : 0.0};
            const double rc1[7] = {bcast16<7>(gB), bcast16<8>(gB), bcast16<9>(gB), bcast16<10>(gB), bcast16<11>(gB),
                                   bcast16<12>(gB), MP ? hT * d.mrate : bcast16<13>(gB)};
            if (FORM == 1 && k < K - 1) {     // step k+1 does not see angle_k: drop its row and column
              a[IA] = 0.0;
              if (role == IA) {
                ASC_UNROLL
                for (int i = 0; i < NS; i++) a[i] = 0.0;
              }
            }
            if (SCHEME == 1 && k < K - 1) {   // pull the value function of step k+1 back through Abar = I + cs F_z(z_k): Abar' on every
              double t[7];                    // column and right-hand side, transpose, Abar' on the columns again
              fzt_lambda(G, a, t);            // (MP: the control passes through a step unchanged)
              ASC_UNROLL
              for (int i = 0; i < 7; i++) a[i] += cs * t[i];
              if (colr) {
                ASC_UNROLL
                for (int i = 0; i < NS; i++) lds_t[grp][role][i] = a[i];
                wsync();
                double r[NS];
                ASC_UNROLL
                for (int l2 = 0; l2 < NS; l2++) r[l2] = lds_t[grp][l2][role];
                fzt_lambda(G, r, t);
                ASC_UNROLL
                for (int i = 0; i < 7; i++) a[i] = r[i] + cs * t[i];
                if constexpr (MP) a[7] = r[7];
              }
              wsync();                        // lds_t is written again below
            }
            ASC_UNROLL
            for (int i = 0; i < NS; i++) a[i] += gsgn[i] * gq[i];
            if constexpr (!MP) {
              a[IA] += bsc * bza;
              a[IM] += bsc * bzm;
            }
            if (dw != 0.0) {
              ASC_UNROLL
              for (int i = 0; i < NS; i++) a[i] += role == i ? dw : 0.0;
            }
            // (MP: the extended step Jacobian is [[A, -bu e_b], [0, 1]], b the defect row the control enters: its inverse transpose acts as
            //  A^-T on the states and adds bu times that row's component to the control's entry)
            double b[NS];
            solveAT<FORM>(G, E, cs, a, b);
            if constexpr (MP) b[7] = a[7] + bu * b[IB];
            if (colr) {       // N <- A^-T N A^-1: the columns, transposed through LDS (in order within a wavefront), the columns again
              ASC_UNROLL
              for (int i = 0; i < NS; i++) lds_t[grp][role][i] = b[i];
              wsync();
              double t[NS];
              ASC_UNROLL
              for (int l2 = 0; l2 < NS; l2++) t[l2] = lds_t[grp][l2][role];
              solveAT<FORM>(G, E, cs, t, b);
              if constexpr (MP) b[7] = t[7] + bu * b[IB];
            }
            double mw[NS], D, coef;
            if constexpr (MP) {       // the stage's control delta enters the movement equation (row 7) with coefficient one
              ASC_UNROLL
              for (int i = 0; i < NS; i++) mw[i] = bcast16<7>(b[i]);
              D = R0 + mw[7];
            } else {
              ASC_UNROLL
              for (int i = 0; i < NS; i++) mw[i] = bu * bcast16<IB>(b[i]);
              D = R0 + dw + bu * mw[IB];
            }
            const double iD = rcp(D);
            if constexpr (MP) {
              const double rsel = role == RL ? ru0 : 0.0;
              coef = (b[7] - rsel) * iD;
            } else {
              const double ru = ru0 + mu * bur, gu = FORM == 1 ? 0.0 : ru0 * ith;
              const double rsel = role == 7 ? ru : role == 8 ? gu : 0.0;
              coef = (bu * b[IB] - rsel) * iD;
            }
            ASC_UNROLL
            for (int i = 0; i < NS; i++) a[i] = b[i] - mw[i] * coef;
            outb[rowK + cj] = role < NS + 3 ? coef : D;      // rows 0 .. NS-1 kap, NS .. NS+2 k0, NS+3: the pivot (for the flush below)
            if (colr) {
              double d0 = 0.0, d1 = 0.0;
              ASC_UNROLL
              for (int i = 0; i < NS; i++) {
                d0 -= a[i] * cc[i];
                if (i < 7) d1 += a[i] * rc1[i];
              }
              lds_d[grp][0][role] = d0;
              lds_d[grp][1][role] = d1;
            }
            wsync();
            {     // right-hand-side lanes: a <- a - P c (the column lanes read zeros and keep their a; their U, V are never used)
              double prc[NS];
              ASC_UNROLL
              for (int i = 0; i < NS; i++) prc[i] = lds_d[grp][drow][i];
              double uu = 0.0, vv = 0.0;
              ASC_UNROLL
              for (int i = 0; i < NS; i++) {
                const double pj = a[i] - prc[i], sj = a[i] + pj;
                if (i < 7) uu += rc1[i] * sj;
                vv += cc[i] * sj;
                a[i] = pj;
              }
              U += uu; V += vv;
            }
          }
        }
        wsync();
        PROF(2);
        // ---- flush the feedback gains of the chunk (node-parallel) ---------------------------------------------------------
        {
          const int k = c * CHN + nl;
          if (k < K && act) {
            if (live) {
              ASC_UNROLL
              for (int i = 0; i < NS + 3; i++) w[(size_t)(R_KA + i) * Kp + k] = outb[i * LDW + col];
            }
            // the node's terms of the border's Schur complement and the sign of its pivot (no recurrence: summed here, 16 nodes at a time)
            const double k00 = outb[NS * LDW + col], k01 = outb[(NS + 1) * LDW + col], k02 = outb[(NS + 2) * LDW + col], D = outb[(NS + 3) * LDW + col];
            const double Dk1 = D * k01, Dk2 = D * k02;
            k10 += Dk1 * k00; k11 += Dk1 * k01; k12 += Dk1 * k02; k20 += Dk2 * k00; k22 += Dk2 * k02;
            if (!(D > 0.0)) bad = 1;
          }
        }
        wsync();
        PROF(3);
      }
      // ---- border: the 2x2 Schur complement in (theta, nu3); inertia ---------------------------------------------------------
      const double U0 = bcast16<RL>(U), U1 = bcast16<RL + 1>(U), V1 = bcast16<RL + 1>(V), U2 = bcast16<RL + 2>(U), V2 = bcast16<RL + 2>(V);
      k10 = gsumW<WIDE>(k10); k11 = gsumW<WIDE>(k11); k12 = gsumW<WIDE>(k12); k20 = gsumW<WIDE>(k20); k22 = gsumW<WIDE>(k22);
      bad = (int)gmaxW<WIDE>((double)bad);
      if (probe_rows) {
        if (act && role == 0) sc[X_STATE] = ST_DONE;
      } else if (act) {
        const double S10 = k10 + 0.5 * (U0 - V1), S11 = k11 + U1, S12 = k12 + 0.5 * U2, S20 = k20 - 0.5 * V2, S22 = k22;
        int ok = !bad;
        double dth = 0.0, dnu3 = 0.0;
        if (ok) {
          const double itl = rcp(s.th - d.tlb), itu = rcp(d.tub - s.th);
          const double rthp = sc[X_RTH] + mu * (itu - itl);
          const double sth = s.zlt * itl + s.zut * itu + dw;
          const double a11 = sth - S11, a12 = -S12, a22 = TERM == 2 ? -1.0 : -S22;      // (TERM 2: no r.v = 0 row; a unit pivot closes nu3)
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
        if (role == 0) {
          if (ok) {
            sc[X_DTH] = dth; sc[X_DNU3] = dnu3; sc[X_SIG1] = sig1; sc[X_SIG2] = sig2; sc[X_RS1] = rs1; sc[X_RS2] = rs2;
            // the violations of the two terminal inequalities, g_i - s_i, are multiplied by sigma_i = z_i/s_i (1e14 for an active
            // constraint at mu = 1e-10) wherever the eliminated slacks re-enter: every phase of this iteration must use the SAME
            // bits -- a re-evaluation whose fused multiply-adds the compiler contracts differently differs by 1e-19, i.e. by
            // 1e-5 in the multiplier step, and the Newton iteration then cycles at that level instead of converging
            sc[X_CG1] = tm.g1 - s.s1; sc[X_CG2] = tm.g2 - s.s2;
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
      const double dt = hT * s.th, be = dt * d.alpha, cs = SCHEME == 1 ? 0.5 * dt : dt, hTc = SCHEME == 1 ? 0.5 * hT : hT;
      const double ha = 0.5 * d.aub, bu = FORM == 1 ? ha : be;
      const double tau = fmax(0.99, 1.0 - mu);
      // ---- forward -----------------------------------------------------------------------------------------------
      {
        // The forward recursion  dz_k = A_k^-1 (dz_{k-1} + x0_k + be du_k e_w),  du_k = du0_k - ka_k . (dz_{k-1} + x0_k)  is affine in
        // dz_{k-1} with node-local coefficients: dz_k = M_k dz_{k-1} + v_k,  M_k = A_k^-1 - be (A_k^-1 e_w) ka_k'.  The node-parallel
        // phase forms M_k and v_k (16 nodes at a time); the serial step is then one row of a 6x6 matrix-vector product per lane
        // (lanes 0-5: x y xdot ydot angle angledot, lane 6: du) -- 6 broadcasts and 7 multiply-adds on a dependency chain of 4
        // instead of ~90 instructions on a chain of ~25.  The mass component has no feedback (dz_m,k = dz_m,k-1 + x0_m,k): it is a
        // prefix sum over the nodes, done in the node-parallel phase.
        // MP: the control is a state -- du_k = du_{k-1} - cu_k + ddelta_k with the stage's control ddelta_k = dd0_k - ka_k . (dx_{k-1} + x0_k)
        // over the eight states: one more coefficient per row (the previous control step), lane 6 carries du_k; ddelta_k = du_k - du_{k-1} + cu_k
        // is taken from the result in the node-parallel phase.
        constexpr int NC = 6 + MP, FS = NC + 1;                         // coefficients per row; rows per lane (coefficients, then v)
        const int fbase = (role < 7 ? FS * role : 0) * LDW;             // rows FS i .. FS i + NC - 1: M[i][..] (lane 6: -ka), row FS i + NC: v[i] (du00)
        const int fout = (role < 6 ? role : role == 6 ? 7 : 8) * LDW;   // out rows 0-5 dz, 6 dz_m (from the scan), 7 du, 8 dummy
        double yown = 0.0, carry_m = 0.0, carry_u = 0.0;
        double rmax = 0.0, gsum = 0.0, adu = 1.0, gmove = 0.0, clu = 0.0;
        double dzK[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < nch; c++) {
          const int kn = c * CHN + nl;
          const bool on = kn < K && act;
          double a_ = 0.5, m_ = 0.5, u_ = 0.0, zb[6] = {1, 1, 1, 1, 1, 1};
          double G[8], Gp[8], E[4], x0[7], ka[NS], du00 = 0.0;       // (Gp: scheme 1 only, the Jacobian block of node k-1)
          double x0u = 0.0, pp_ = 1.0, pn_ = 1.0, zpp_ = 0.0, zpn_ = 0.0, lu_ = 0.0;      // (MP only)
          ASC_UNROLL
          for (int i = 0; i < 7; i++) x0[i] = 0.0;
          ASC_UNROLL
          for (int i = 0; i < NS; i++) ka[i] = 0.0;
          if (on) {
            double z[7], zp[7], F[7], ax, ay;
            ASC_UNROLL
            for (int i = 0; i < 7; i++) { z[i] = it[(O_Z + i) * Kp + kn]; zp[i] = kn > 0 ? it[(O_Z + i) * Kp + kn - 1] : 0.0; }
            u_ = it[O_U * Kp + kn];
            a_ = z[IA]; m_ = z[IM];
            ASC_UNROLL
            for (int b = 0; b < 6; b++) zb[b] = it[(O_ZB + b) * Kp + kn];
            ASC_UNROLL
            for (int i = 0; i < NS; i++) ka[i] = w[(size_t)(R_KA + i) * Kp + kn];
            if constexpr (MP) {
              pp_ = it[O_PP * Kp + kn]; pn_ = it[O_PN * Kp + kn]; zpp_ = it[O_ZP * Kp + kn]; zpn_ = it[O_ZN * Kp + kn]; lu_ = it[O_LU * Kp + kn];
              x0u = -(u_ - (kn > 0 ? it[O_U * Kp + kn - 1] : UINIT) - pp_ + pn_);
            }
            du00 = w[(size_t)R_K0 * Kp + kn] + w[(size_t)(R_K0 + 1) * Kp + kn] * dth + w[(size_t)(R_K0 + 2) * Kp + kn] * dnu3;
            accel<1>(d, z[IX], z[IY], z[IA], z[IM], 0.0, 0.0, ax, ay, G, nullptr);
            rhs_f<FORM>(d, z, u_, ax, ay, F);
            if (SCHEME == 1) {       // second evaluation point f(z_{k-1}, u_k); its Jacobian is the Abar_k = I + cs F_z(z_{k-1}) of the step
              double Fb[7], axp, ayp;
              accel<1>(d, zp[IX], zp[IY], zp[IA], zp[IM], 0.0, 0.0, axp, ayp, Gp, nullptr);
              rhs_f<FORM>(d, zp, u_, axp, ayp, Fb);
              ASC_UNROLL
              for (int i = 0; i < 7; i++) F[i] = 0.5 * (F[i] + Fb[i]);
            }
            implicit_block(G, cs, E);
            ASC_UNROLL
            for (int i = 0; i < 7; i++) {
              x0[i] = hT * F[i] * dth - ((FORM == 1 && i == IA) ? z[IA] - ha * (u_ + 1.0) : z[i] - zp[i] - dt * F[i]);
              du00 -= ka[i] * x0[i];
            }
            if constexpr (MP) du00 -= ka[7] * x0u;
          }
          // dz_m: inclusive prefix sum of x0_m over the nodes of the NLP (16 here, the chunks before in carry_m)
          double incl = x0[IM];
          ASC_UNROLL
          for (int sft = 1; sft < CHN; sft *= 2) {
            const double t = __shfl_up(incl, sft, CHN);
            if (nl >= sft) incl += t;
          }
          const double dzm_k = carry_m + incl, dzm_p = dzm_k - x0[IM];
          carry_m += WIDE ? __shfl(incl, 63) : bcast16<15>(incl);
          if (on) {
            const double du00p = du00 - ka[IM] * dzm_p;
            double rvx[7], rvy[7];
            ainv_vrows<FORM>(G, E, cs, rvx, rvy);
            // row i of M = (row i of A^-1) - be aW[i] ka',  v[i] = (row i of A^-1) . x0 + be aW[i] du00' + A^-1[i][m] dz_m,k-1,  aW = A^-1 e_w
            // (MP: du_k = (x0u + dd0) - ka . dx_{k-1} + (1 - ka_u) du_{k-1} takes the place of du_k: one more column, be aW[i] (1 - ka_u))
            auto emit = [&](int i, const double *r, double aw) {
              const double bw = bu * aw;
              if constexpr (MP) stage[(FS * i + 6) * LDW + col] = bw * (1.0 - ka[7]);
              if (SCHEME == 1) {       // dz_k = M Abar_k dz_{k-1} + v: the row times Abar (= row + cs F_z' row), its mass entry folded into v
                double m7[7], t[7];
                ASC_UNROLL
                for (int j = 0; j < 7; j++) m7[j] = r[j] - bw * ka[j];
                fzt_lambda(Gp, m7, t);
                ASC_UNROLL
                for (int j = 0; j < 7; j++) m7[j] += cs * t[j];
                double v = bw * (MP ? x0u + du00 : du00) + m7[IM] * dzm_p;
                ASC_UNROLL
                for (int j = 0; j < 7; j++) v += r[j] * x0[j];
                ASC_UNROLL
                for (int j = 0; j < 6; j++) stage[(FS * i + j) * LDW + col] = m7[j];
                stage[(FS * i + NC) * LDW + col] = v;
                return;
              }
              double v = bw * (MP ? x0u + du00p : du00p) + r[IM] * dzm_p;
              ASC_UNROLL
              for (int j = 0; j < 7; j++) v += r[j] * x0[j];
              ASC_UNROLL
              for (int j = 0; j < 6; j++) stage[(FS * i + j) * LDW + col] = (FORM == 1 && j == IA) ? 0.0 : r[j] - bw * ka[j];     // (v1: no coupling to angle_{k-1})
              stage[(FS * i + NC) * LDW + col] = v;
            };
            // (aW above is A^-1 e_b with b the row the control enters: angledot, or the algebraic angle row of the v1 formulation)
            emit(IVX, rvx, rvx[IB]);
            emit(IVY, rvy, rvy[IB]);
            {
              double r[7];
              ASC_UNROLL
              for (int j = 0; j < 7; j++) r[j] = (j == IX ? 1.0 : 0.0) + cs * rvx[j];
              emit(IX, r, cs * rvx[IB]);
              ASC_UNROLL
              for (int j = 0; j < 7; j++) r[j] = (j == IY ? 1.0 : 0.0) + cs * rvy[j];
              emit(IY, r, cs * rvy[IB]);
              ASC_UNROLL
              for (int j = 0; j < 7; j++) r[j] = j == IA ? 1.0 : (j == IW && FORM == 0) ? cs : 0.0;
              emit(IA, r, FORM == 1 ? 1.0 : cs);
              ASC_UNROLL
              for (int j = 0; j < 7; j++) r[j] = j == IW ? 1.0 : 0.0;
              emit(IW, r, FORM == 1 ? 0.0 : 1.0);
            }
            if constexpr (MP) {        // row of du_k (lane 6)
              double kj[7], v0;
              if (SCHEME == 1) {
                double t[7];
                fzt_lambda(Gp, ka, t);
                ASC_UNROLL
                for (int j = 0; j < 7; j++) kj[j] = ka[j] + cs * t[j];
                v0 = du00 - kj[IM] * dzm_p;
              } else {
                ASC_UNROLL
                for (int j = 0; j < 7; j++) kj[j] = ka[j];
                v0 = du00p;
              }
              ASC_UNROLL
              for (int j = 0; j < 6; j++) stage[(FS * 6 + j) * LDW + col] = (FORM == 1 && j == IA) ? 0.0 : -kj[j];     // (v1: no coupling to angle_{k-1})
              stage[(FS * 6 + 6) * LDW + col] = 1.0 - ka[7];
              stage[(FS * 6 + NC) * LDW + col] = x0u + v0;
            } else if (SCHEME == 1) {         // du_k = du00 - ka' Abar_k dz_{k-1}
              double t[7];
              fzt_lambda(Gp, ka, t);
              ASC_UNROLL
              for (int j = 0; j < 6; j++) stage[(42 + j) * LDW + col] = -(ka[j] + cs * t[j]);
              stage[48 * LDW + col] = du00 - (ka[IM] + cs * t[IM]) * dzm_p;
            } else {
              ASC_UNROLL
              for (int j = 0; j < 6; j++) stage[(42 + j) * LDW + col] = (FORM == 1 && j == IA) ? 0.0 : -ka[j];
              stage[48 * LDW + col] = du00p;
            }
            outb[6 * LDW + col] = dzm_k;
          }
          wsync();
          PROF(4);
          if (act) {
            const int jn = min(CHN, K - c * CHN);
            for (int jj = 0; jj < jn; jj++) {
              const int cj = cbase + jj;
              const double *sj = stage + fbase + cj;
              const double m0 = sj[0], m1 = sj[LDW], m2 = sj[2 * LDW], m3 = sj[3 * LDW], m4 = sj[4 * LDW], m5 = sj[5 * LDW], vv = sj[NC * LDW];
              const double b0 = bcast16<0>(yown), b1 = bcast16<1>(yown), b2 = bcast16<2>(yown), b3 = bcast16<3>(yown),
                           b4 = bcast16<4>(yown), b5 = bcast16<5>(yown);
              const double e0 = (vv + m0 * b0) + m2 * b2, e1 = m1 * b1 + m3 * b3, e2 = m4 * b4 + m5 * b5;
              if constexpr (MP) {
                const double m6 = sj[6 * LDW], b6 = bcast16<6>(yown);
                yown = ((e0 + e1) + e2) + m6 * b6;
              } else {
                yown = (e0 + e1) + e2;
              }
              outb[fout + cj] = yown;
            }
          }
          wsync();
          PROF(5);
          // ---- node-parallel: store the primal step, bound-multiplier steps, fraction to the boundary ----------------------
          if (kn < K && act) {
            double dzn[8];
            ASC_UNROLL
            for (int i = 0; i < 8; i++) dzn[i] = outb[i * LDW + col];
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
            if constexpr (MP) {
              // The movement multiplier from the stage control's own stationarity row, d lambda_u = Rd ddelta + gdl, then the slack pair:
              // the slack with the larger curvature from its own row (well conditioned), the other one from ddelta = dp - dn (its own row
              // divides a difference of two nearly equal numbers by a curvature that vanishes for an inactive slack)
              const double du_p = nl > 0 ? outb[7 * LDW + col - 1] : carry_u;
              const double ddel = (du - du_p) - x0u;
              const MovePivot mvp = move_pivot(pp_, pn_, zpp_, zpn_, lu_, dcw, mu, dw);
              const double dlu = mvp.Rd * ddel + mvp.gdl;
              double dpp, dpn;
              if (mvp.sgp >= mvp.sgn) { dpp = (dlu - mvp.rp) * rcp(mvp.sgp); dpn = dpp - ddel; }
              else { dpn = (-dlu - mvp.rn) * rcp(mvp.sgn); dpp = ddel + dpn; }
              const double dzp = mvp.ip * (mu - zpp_ * dpp) - zpp_, dzn_ = mvp.in_ * (mu - zpn_ * dpn) - zpn_;
              ASC_FTBR(rmax, mvp.ip, dpp); ASC_FTBR(rmax, mvp.in_, dpn);
              ASC_FTB(adu, zpp_, dzp); ASC_FTB(adu, zpn_, dzn_);
              gsum -= dpp * mvp.ip + dpn * mvp.in_;
              gmove += dpp + dpn;
              clu -= x0u * (lu_ + dlu);           // c_u (lambda_u + d lambda_u) of the merit function's curvature estimate
              if (live) {
                stp[O_LU * Kp + kn] = dlu; stp[O_PP * Kp + kn] = dpp; stp[O_PN * Kp + kn] = dpn; stp[O_ZP * Kp + kn] = dzp; stp[O_ZN * Kp + kn] = dzn_;
              }
            }
          }
          if constexpr (MP) carry_u = outb[7 * LDW + cbase + CHN - 1];       // the control step of the chunk's last node
          wsync();
          PROF(6);
        }
        rmax = gmaxW<WIDE>(rmax); gsum = gsumW<WIDE>(gsum); adu = gminW<WIDE>(adu);
        if constexpr (MP) { gmove = gsumW<WIDE>(gmove); clu = gsumW<WIDE>(clu); }
        ASC_UNROLL
        for (int i = 0; i < 7; i++) dzK[i] = gsumW<WIDE>(dzK[i]);          // only the lane of the last node holds non-zeros
        // ---- the scalars of the step and the step lengths: known once the primal step is (the adjoint below only adds the
        //      multiplier steps), so that the adjoint phase can evaluate the first trial point of the line search as it goes ----
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
        TrialCtx tc;
        tc.alpha = apr; tc.adu = adu; tc.mlo = mu * 1e-10; tc.mhi = mu * 1e10; tc.hT = hT; tc.first = false; tc.dcw = dcw;
        tc.stt = trial_scal(d, s, ds, apr, adu, mu, false);
        tc.dt = hT * tc.stt.th; tc.be = tc.dt * d.alpha;
        double *in = w + (size_t)((1 - (int)sc[X_CUR]) * NIT) * Kp;
        Part P;
        P.clear();
        // ---- adjoint (backwards over the chunks) ---------------------------------------------------------------------------
        // The adjoint recursion  dl_k = A_k^-T (r_k + dl_{k+1})  in the same affine form: dl_k = N_k dl_{k+1} + w_k with the node-local
        // N_k = A_k^-T and w_k = A_k^-T r_k.  Columns angledot and mass of A^-T are unit vectors (rows angledot and mass of A are), so a
        // lane needs five coefficients, its own previous value (lanes 5 and 6) and w: rows 6i .. 6i+4: N[i][0..4], row 6i+5: w[i].
        const int abase = (role < 7 ? 6 * role : 0) * LDW;
        const int aout = (role < 7 ? role : 8) * LDW;
        const double aself = (role == IW || role == IM) ? 1.0 : 0.0;
        double lown = 0.0;
        double cl = 0.0, ccl = 0.0;
        for (int c = nch - 1; c >= 0; c--) {
          const int kn = c * CHN + nl;
          double ccn[7] = {0, 0, 0, 0, 0, 0, 0};
          TrialKeep kp;                       // what the trial point's dual rows need of its primal part (evaluated before the sweep)
          if (kn < K && act) {
            NodeIn n, dn;
            load_node<MP>(it, Kp, K, kn, n, UINIT);
            ASC_UNROLL
            for (int i = 0; i < 7; i++) { dn.z[i] = stp[(O_Z + i) * Kp + kn]; dn.zp[i] = kn > 0 ? stp[(O_Z + i) * Kp + kn - 1] : 0.0; }
            dn.u = stp[O_U * Kp + kn];
            ASC_UNROLL
            for (int b = 0; b < 6; b++) dn.zb[b] = stp[(O_ZB + b) * Kp + kn];
            if constexpr (MP) {
              dn.up = kn > 0 ? stp[O_U * Kp + kn - 1] : 0.0;
              dn.lu = stp[O_LU * Kp + kn];
              dn.lun = kn + 1 < K ? stp[O_LU * Kp + kn + 1] : 0.0;
              dn.pp = stp[O_PP * Kp + kn]; dn.pn = stp[O_PN * Kp + kn]; dn.zpp = stp[O_ZP * Kp + kn]; dn.zpn = stp[O_ZN * Kp + kn];
            }
            const double *dz = dn.z;
            double G[8], E[4], H[10], F[7], fl[7], lt[7], ax, ay;
            ASC_UNROLL
            for (int i = 0; i < 7; i++) lt[i] = SCHEME == 1 ? n.l[i] + n.ln[i] : n.l[i];
            accel<2>(d, n.z[IX], n.z[IY], n.z[IA], n.z[IM], -cs * lt[IVX], -cs * lt[IVY], ax, ay, G, H);
            rhs_f<FORM>(d, n.z, n.u, ax, ay, F);
            if (SCHEME == 1) {
              double Fb[7], axp, ayp;
              accel<0>(d, n.zp[IX], n.zp[IY], n.zp[IA], n.zp[IM], 0.0, 0.0, axp, ayp, nullptr, nullptr);
              rhs_f<FORM>(d, n.zp, n.u, axp, ayp, Fb);
              ASC_UNROLL
              for (int i = 0; i < 7; i++) F[i] = 0.5 * (F[i] + Fb[i]);
            }
            implicit_block(G, cs, E);
            fzt_lambda<FORM>(G, lt, fl);
            const double id0 = rcp(n.z[IA]), id1 = rcp(d.aub - n.z[IA]), id2 = rcp(n.z[IM]), id3 = rcp(1.0 - n.z[IM]);
            H[7] += n.zb[0] * id0 + n.zb[1] * id1;
            H[9] += n.zb[2] * id2 + n.zb[3] * id3;
            double r[7];
            ASC_UNROLL
            for (int i = 0; i < 7; i++) {
              const double rz = (FORM == 1 && i == IA) ? n.l[i] - cs * fl[i] : n.l[i] - cs * fl[i] - n.ln[i], gt = -hTc * fl[i];
              r[i] = -rz - gt * dth - dw * dz[i];
              ccn[i] = (FORM == 1 && i == IA) ? n.z[IA] - ha * (n.u + 1.0) : n.z[i] - n.zp[i] - dt * F[i];
              ccl += ccn[i] * n.l[i];
            }
            r[IA] -= mu * (id1 - id0);
            r[IM] -= mu * (id3 - id2);
            r[IX] -= H[0] * dz[IX] + H[1] * dz[IY] + H[2] * dz[IA] + H[3] * dz[IM];
            r[IY] -= H[1] * dz[IX] + H[4] * dz[IY] + H[5] * dz[IA] + H[6] * dz[IM];
            r[IA] -= H[2] * dz[IX] + H[5] * dz[IY] + H[7] * dz[IA] + H[8] * dz[IM];
            r[IM] -= H[3] * dz[IX] + H[6] * dz[IY] + H[8] * dz[IA] + H[9] * dz[IM];
            if (kn == K - 1) {
              double QT[28], qd[7];
              const Terminal tm = TERM == 2 ? terminal_eval_any(d, n.z) : terminal_eval(d, n.z);
              ASC_UNROLL
              for (int i = 0; i < 28; i++) QT[i] = 0.0;
              if constexpr (TERM == 2) terminal_hessian_any(QT, tm, s.nu1, s.nu2, sig1, sig2);
              else terminal_hessian(QT, tm, s.nu3, s.nu1, s.nu2, sig1, sig2);
              symv(QT, dz, qd);
              ASC_UNROLL
              for (int i = 0; i < 7; i++) r[i] -= qd[i];
              const double w1 = s.nu1 + sig1 * sc[X_CG1] + rs1, w2 = s.nu2 + sig2 * sc[X_CG2] + rs2;
              if constexpr (TERM == 2) {
                double g4[4];
                terminal_grad_any(tm, w1, w2, g4);
                r[IX] -= g4[0]; r[IY] -= g4[1]; r[IVX] -= g4[2]; r[IVY] -= g4[3];
              } else {
              r[IX] -= s.nu3 * tm.e3g[0] + w1 * tm.g1g[0] + tm.e3g[0] * dnu3;
              r[IY] -= s.nu3 * tm.e3g[1] + w1 * tm.g1g[1] + tm.e3g[1] * dnu3;
              r[IVX] -= s.nu3 * tm.e3g[2] + w2 * tm.g2g[0] + tm.e3g[2] * dnu3;
              r[IVY] -= s.nu3 * tm.e3g[3] + w2 * tm.g2g[1] + tm.e3g[3] * dnu3;
              }
            }
            double wv[7];
            solveAT<FORM>(G, E, cs, r, wv);
            {     // N = A^-T: N[i][j] = A^-1[j][i], j = x y xdot ydot angle (the rows of A^-1 above)
              double rvx[7], rvy[7];
              ainv_vrows<FORM>(G, E, cs, rvx, rvy);
              ASC_UNROLL
              for (int i = 0; i < 7; i++) {
                double nr[7];             // row i of A^-T
                nr[IVX] = rvx[i]; nr[IVY] = rvy[i];
                nr[IX] = (i == IX ? 1.0 : 0.0) + cs * rvx[i];
                nr[IY] = (i == IY ? 1.0 : 0.0) + cs * rvy[i];
                nr[IA] = FORM == 1 ? 0.0 : i == IA ? 1.0 : i == IW ? cs : 0.0;      // (v1: no coupling to the angle multiplier of step k+1)
                nr[IW] = i == IW ? 1.0 : 0.0;
                nr[IM] = i == IM ? 1.0 : 0.0;
                if (SCHEME == 1) {      // dl_k = A_k^-T (r_k + Abar_{k+1}' dl_{k+1}),  Abar_{k+1} = I + cs F_z(z_k): the row times Abar'
                  double t[7];
                  fz_mul(G, nr, t);
                  ASC_UNROLL
                  for (int j = 0; j < 7; j++) nr[j] += cs * t[j];
                }
                ASC_UNROLL
                for (int j = 0; j < 5; j++) stage[(6 * i + j) * LDW + col] = nr[j];
              }
            }
            ASC_UNROLL
            for (int i = 0; i < 7; i++) stage[(6 * i + 5) * LDW + col] = wv[i];
            // the primal step of the chunk is known: the primal part of its trial point at the first step length, before the sweep
            trial_primal<SCHEME, FORM, MP, TERM>(d, K, Kp, kn, n, dn, tc, live, in, P, kp);
          }
          wsync();
          PROF(7);
          if (act) {
            const int jj0 = min(CHN, K - c * CHN) - 1;
            for (int jj = jj0; jj >= 0; jj--) {
              const int cj = cbase + jj;
              const double *sj = stage + abase + cj;
              const double n0 = sj[0], n1 = sj[LDW], n2 = sj[2 * LDW], n3 = sj[3 * LDW], n4 = sj[4 * LDW], wv = sj[5 * LDW];
              const double b0 = bcast16<0>(lown), b1 = bcast16<1>(lown), b2 = bcast16<2>(lown), b3 = bcast16<3>(lown), b4 = bcast16<4>(lown);
              const double e0 = (wv + aself * lown) + n0 * b0, e1 = n1 * b1 + n2 * b2, e2 = n3 * b3 + n4 * b4;
              lown = (e0 + e1) + e2;
              outb[aout + cj] = lown;
            }
          }
          wsync();
          double dl[7], dln[7];
          if (kn < K && act) {
            ASC_UNROLL
            for (int i = 0; i < 7; i++) {
              dl[i] = outb[i * LDW + col];
              dln[i] = kn + 1 < K ? (nl < CHN - 1 ? outb[i * LDW + col + 1] : lds_c[WIDE ? 0 : grp][i]) : 0.0;
              ccl += ccn[i] * dl[i];            // c . dlambda: no recurrence, summed here
            }
            if (live) {
              ASC_UNROLL
              for (int i = 0; i < 7; i++) stp[(O_L + i) * Kp + kn] = dl[i];
            }
          }
          wsync();
          PROF(8);
          // ---- node-parallel: the step of the chunk is complete -> the dual rows of its trial point (the iterate's multipliers
          //      come from HBM again rather than being held across the sweep) ------------------------------------------------------
          if (kn < K && act) {
            if (nl == 0) {
              ASC_UNROLL
              for (int i = 0; i < 7; i++) lds_c[WIDE ? 0 : grp][i] = dl[i];
            }
            double l[7], ln[7], lu = 0.0, lun = 0.0;
            ASC_UNROLL
            for (int i = 0; i < 7; i++) {
              l[i] = it[(O_L + i) * Kp + kn] + tc.alpha * dl[i];
              ln[i] = (kn + 1 < K ? it[(O_L + i) * Kp + kn + 1] : 0.0) + tc.alpha * dln[i];
            }
            if constexpr (MP) {
              lu = it[O_LU * Kp + kn] + tc.alpha * stp[O_LU * Kp + kn];
              lun = (kn + 1 < K ? it[O_LU * Kp + kn + 1] : 0.0) + tc.alpha * (kn + 1 < K ? stp[O_LU * Kp + kn + 1] : 0.0);
            }
            trial_dual<SCHEME, FORM, MP, TERM>(d, K, Kp, kn, kp, l, ln, lu, lun, tc, live, in, P);
          }
          PROF(0);
        }
        P.template reduceW<MP, WIDE>();
        // ---- scalars of the step, merit bookkeeping -------------------------------------------------------------------------
        ccl = gsumW<WIDE>(ccl);
        if (act) {
          cl += ccl;
          const Terminal &tm = tmK;
          double gd = MP ? mu * gsum + dcw * gmove : mu * gsum;
          if constexpr (MP) cl += clu;
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
            put_scal(sc, X_D, ds);
            sc[X_NUP] = nu_pen;
            sc[X_DM] = gd - nu_pen * c1;
            sc[X_PHI0] = (MP ? s.th + dcw * sc[X_MV] : s.th) - mu * slog + nu_pen * c1;
            sc[X_ALPHA] = apr; sc[X_ADU] = adu; sc[X_LS] = 0.0;
            sc[X_P + 0] = P.rd; sc[X_P + 1] = P.cinf; sc[X_P + 2] = P.pmin; sc[X_P + 3] = P.pmax; sc[X_P + 4] = P.l1;
            sc[X_P + 5] = P.zsum; sc[X_P + 6] = P.rth; sc[X_P + 7] = P.c1; sc[X_P + 8] = P.sl; sc[X_P + 9] = P.mv;
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
  if (live)
    for (int r = role; r < NSCAL; r += 16) gsc[r] = sc[r];
}

}  // namespace

namespace ascent {

// wide: one NLP per wavefront, 64-node chunks (node arrays padded to a multiple of 64); otherwise four NLPs per wavefront, 16-node chunks.
// scheme 2 (Hermite-Simpson, ascent_hs.hip): chunks of 48 / 12 nodes.
static PGeo geo_of(int K, int form = 0, int mp = 0, int term = 0, int wide = 0, int scheme = 0) {
  PGeo g;
  const int ch = scheme == 2 ? hs_chunk_nodes(wide) : wide ? 64 : CH;
  g.K = K; g.nch = (K + ch - 1) / ch; g.Kp = (g.nch * ch + 15) / 16 * 16;      // (rows of an NLP start on 128-byte lines: see PGeo::nlp_doubles)
  g.form = form; g.mp = mp ? 1 : 0; g.term = term == 2 ? 2 : 0; g.wide = wide ? 1 : 0;
  return g;
}
// Batches that cannot give every SIMD a wavefront of four NLPs (MI355X: 256 CUs x 4 SIMDs) run one NLP per wavefront
// (every variant of the kernel has both forms).
static int use_wide(long batch, int scheme, int form, int mp, int term) {
  (void)scheme; (void)form; (void)mp; (void)term;
  if (const char *e = getenv("ASCENT_PERSIST_WIDE")) return e[0] == '1';
  return batch <= 1024;
}

// One grid level's workspace, rounded up to a multiple of 256 bytes: the two regions of the nested iteration are laid out
// back to back with exactly these sizes (persist_region1_offset below is the one place that says where the second one starts).
// Sized for the largest padding of the node arrays any kernel form uses (chunks of 64, 48, 16 or 12 nodes): enough for each.
static size_t level_doubles(int K, int mp) {
  size_t m = 0;
  for (int scheme = 0; scheme <= 2; scheme += 2)
    for (int wide = 0; wide <= 1; wide++) {
      const size_t n = geo_of(K, 0, mp, 0, wide, scheme).nlp_doubles();
      m = n > m ? n : m;
    }
  return m;
}
size_t persist_ws_bytes(int K, long batch, int mp) {
  const size_t b = (size_t)batch * level_doubles(K, mp) * sizeof(double) + 64;
  return (b + 255) & ~(size_t)255;
}
size_t persist_region1_offset(const int *levels, long batch, int mp) { return persist_ws_bytes(levels[0] - 1, batch, mp); }
size_t persist_level_bytes_used(int K, long batch, int mp) { return (size_t)batch * level_doubles(K, mp) * sizeof(double); }
size_t persist_ws_bytes_nested(const int *levels, int nlev, long batch, int mp) {
  size_t b = persist_region1_offset(levels, batch, mp);
  if (nlev > 1) b += persist_ws_bytes(levels[1] - 1, batch, mp);
  return b;
}

#define PCHK2(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { snprintf(err, errlen, "%s: %s", #call, hipGetErrorString(e_)); return ASCENT_E_HIP; } } while (0)

static void launch_solve(int scheme, int form, int mp, long batch, hipStream_t stream, const ascent_params *dp, const PGeo &g, double *w,
                         int max_iter, double tol) {
  if (scheme == 2) {      // Hermite-Simpson: ascent_hs.hip
    hs_launch_solve(batch, stream, dp, g.K, g.Kp, g.nch, g.term, g.wide, w, max_iter, tol);
    return;
  }
  if (g.wide) {      // one NLP per wavefront
    const dim3 gw((unsigned)batch), bw(WAVE);
    if (g.term == 2) {      // burnout anywhere on the ellipse (formulation 0)
      if (mp && scheme == 1) hipLaunchKernelGGL((p_solve<1, 0, 1, 2, 1>), gw, bw, 0, stream, dp, batch, g, w, max_iter, tol);
      else if (mp) hipLaunchKernelGGL((p_solve<0, 0, 1, 2, 1>), gw, bw, 0, stream, dp, batch, g, w, max_iter, tol);
      else if (scheme == 1) hipLaunchKernelGGL((p_solve<1, 0, 0, 2, 1>), gw, bw, 0, stream, dp, batch, g, w, max_iter, tol);
      else hipLaunchKernelGGL((p_solve<0, 0, 0, 2, 1>), gw, bw, 0, stream, dp, batch, g, w, max_iter, tol);
      return;
    }
    if (mp && scheme == 1) hipLaunchKernelGGL((p_solve<1, 0, 1, 0, 1>), gw, bw, 0, stream, dp, batch, g, w, max_iter, tol);
    else if (mp && form == 1) hipLaunchKernelGGL((p_solve<0, 1, 1, 0, 1>), gw, bw, 0, stream, dp, batch, g, w, max_iter, tol);
    else if (form == 1) hipLaunchKernelGGL((p_solve<0, 1, 0, 0, 1>), gw, bw, 0, stream, dp, batch, g, w, max_iter, tol);
    else if (mp) hipLaunchKernelGGL((p_solve<0, 0, 1, 0, 1>), gw, bw, 0, stream, dp, batch, g, w, max_iter, tol);
    else if (scheme == 1) hipLaunchKernelGGL((p_solve<1, 0, 0, 0, 1>), gw, bw, 0, stream, dp, batch, g, w, max_iter, tol);
    else hipLaunchKernelGGL((p_solve<0, 0, 0, 0, 1>), gw, bw, 0, stream, dp, batch, g, w, max_iter, tol);
    return;
  }
  const dim3 grid((unsigned)((batch + NPW - 1) / NPW)), block(WAVE);
  if (g.term == 2) {      // burnout anywhere on the ellipse (formulation 0)
    if (mp && scheme == 1) hipLaunchKernelGGL((p_solve<1, 0, 1, 2>), grid, block, 0, stream, dp, batch, g, w, max_iter, tol);
    else if (mp) hipLaunchKernelGGL((p_solve<0, 0, 1, 2>), grid, block, 0, stream, dp, batch, g, w, max_iter, tol);
    else if (scheme == 1) hipLaunchKernelGGL((p_solve<1, 0, 0, 2>), grid, block, 0, stream, dp, batch, g, w, max_iter, tol);
    else hipLaunchKernelGGL((p_solve<0, 0, 0, 2>), grid, block, 0, stream, dp, batch, g, w, max_iter, tol);
    return;
  }
  if (mp && scheme == 1) hipLaunchKernelGGL((p_solve<1, 0, 1>), grid, block, 0, stream, dp, batch, g, w, max_iter, tol);
  else if (mp && form == 1) hipLaunchKernelGGL((p_solve<0, 1, 1>), grid, block, 0, stream, dp, batch, g, w, max_iter, tol);
  else if (mp) hipLaunchKernelGGL((p_solve<0, 0, 1>), grid, block, 0, stream, dp, batch, g, w, max_iter, tol);
  else if (scheme == 1) hipLaunchKernelGGL((p_solve<1, 0, 0>), grid, block, 0, stream, dp, batch, g, w, max_iter, tol);
  else if (form == 1) hipLaunchKernelGGL((p_solve<0, 1, 0>), grid, block, 0, stream, dp, batch, g, w, max_iter, tol);
  else hipLaunchKernelGGL((p_solve<0, 0, 0>), grid, block, 0, stream, dp, batch, g, w, max_iter, tol);
}

// All grid levels of the nested iteration (levels[0] = the requested grid, finest first; the coarsest is solved first, cold or
// from the caller's guess): p_init, then per level p_solve and p_transfer to the next finer grid, p_finish at the end.  Two
// workspace regions alternate between the levels.  mp: with the l1 move penalty (schemes 0 / 1, formulation 0).
int persist_run_nested(const ascent_params *dp, long batch, int scheme, int form, int mp, int term, const int *levels, int nlev, double *ws, const double *dguess, int warm,
                       int max_iter, double tol, double tol_coarse, double mu0, double mu_first, double mu_next, double *dtraj,
                       double *dtf, int *dstatus, int *diters, double *dblob, hipStream_t stream, char *err, size_t errlen) {
  if (term == 2 && form != 0) { snprintf(err, errlen, "the persistent kernel carries terminal 2 for formulation 0 only"); return ASCENT_E_ARG; }
  if (scheme == 2 && (form != 0 || mp)) { snprintf(err, errlen, "the persistent Hermite-Simpson kernel has formulation 0 without the move penalty only"); return ASCENT_E_ARG; }
  double *region[2] = {ws, (double *)((char *)ws + persist_region1_offset(levels, batch, mp))};
  const int wide = use_wide(batch, scheme, form, mp, term);
  PGeo g = geo_of(levels[nlev - 1] - 1, form, mp, term, wide, scheme);
  double *w = region[(nlev - 1) & 1];
  hipLaunchKernelGGL(p_init, dim3((unsigned)((g.Kp + WAVE - 1) / WAVE), (unsigned)batch), dim3(WAVE), 0, stream, dp, batch, g, w, dguess,
                     warm, mu0, (const double *)nullptr, (const double *)nullptr, 0);
  for (int l = nlev - 1; l >= 0; l--) {
    launch_solve(scheme, form, mp, batch, stream, dp, g, w, max_iter, l == 0 ? tol : tol_coarse);
    if (l > 0) {
      const PGeo gf = geo_of(levels[l - 1] - 1, form, mp, term, wide, scheme);
      double *wf = region[(l - 1) & 1];
      hipLaunchKernelGGL(p_transfer, dim3((unsigned)((gf.Kp + WAVE - 1) / WAVE), (unsigned)batch), dim3(WAVE), 0, stream, dp, batch, g,
                         (const double *)w, gf, wf, l == nlev - 1 ? mu_first : mu_next);
      g = gf; w = wf;
    }
  }
  hipLaunchKernelGGL(p_finish, dim3((unsigned)((g.K + CH - 1) / CH), (unsigned)((batch + WAVE - 1) / WAVE)), dim3(WAVE, FIN_WAVES), 0, stream, dp, batch, g,
                     (const double *)w, dtraj, dtf, dstatus, diters, dblob);
  PCHK2(hipGetLastError());
  return ASCENT_OK;
}

// One round of p_solve at a caller-supplied iterate (parity surface ascent_kkt_step_path): the iterate as it is, mu and
// delta_w per problem from the caller; p_probe_out hands back the Newton step.  (mp: the slack pairs, which the blob does
// not carry, are set around the iterate's own movement as every warm start sets them.)
int persist_probe(const ascent_params *dp, long batch, int scheme, int form, int mp, int term, int K, double *ws, const double *diterate, const double *dmu, const double *ddw,
                  double *dstep, int *dinertia, hipStream_t stream, char *err, size_t errlen) {
  if (term == 2 && form != 0) { snprintf(err, errlen, "the persistent kernel carries terminal 2 for formulation 0 only"); return ASCENT_E_ARG; }
  if (scheme == 2 && (form != 0 || mp)) { snprintf(err, errlen, "the persistent Hermite-Simpson kernel has formulation 0 without the move penalty only"); return ASCENT_E_ARG; }
  const PGeo g = geo_of(K, form, mp, term, use_wide(batch, scheme, form, mp, term), scheme);
  const dim3 ng((unsigned)((g.Kp + WAVE - 1) / WAVE), (unsigned)batch);
  hipLaunchKernelGGL(p_init, ng, dim3(WAVE), 0, stream, dp, batch, g, ws, diterate, 2, 0.1, dmu, ddw, 1);
  launch_solve(scheme, form, mp, batch, stream, dp, g, ws, 1000, -1.0);
  hipLaunchKernelGGL(p_probe_out, ng, dim3(WAVE), 0, stream, batch, g, (const double *)ws, dstep, dinertia);
  PCHK2(hipGetLastError());
  return ASCENT_OK;
}

// The node rows of the same kernel (parity surface ascent_eval_nodes_path): one round of p_solve up to the point where the
// blocks of every chunk stand in LDS; they are copied out instead of being swept.  dzero: a device array of `batch` zeros
// (mu and delta_w do not enter the rows).
int persist_probe_rows(const ascent_params *dp, long batch, int scheme, int form, int K, double *ws, const double *diterate, const double *dzero,
                       double *ddefects, double *djac, double *dhess, hipStream_t stream, char *err, size_t errlen) {
  const PGeo g = geo_of(K, form, 0, 0, use_wide(batch, scheme, form, 0, 0));
  const dim3 ng((unsigned)((g.Kp + WAVE - 1) / WAVE), (unsigned)batch);
  hipLaunchKernelGGL(p_init, ng, dim3(WAVE), 0, stream, dp, batch, g, ws, diterate, 2, 0.1, dzero, dzero, 2);
  launch_solve(scheme, form, 0, batch, stream, dp, g, ws, 1000, -1.0);
  hipLaunchKernelGGL(p_probe_rows_out, ng, dim3(WAVE), 0, stream, dp, batch, g, (const double *)ws, ddefects, djac, dhess);
  PCHK2(hipGetLastError());
  return ASCENT_OK;
}

}  // namespace ascent
