// libascent: batched primal-dual interior-point solver for the lunar-ascent collocation NLP of
// /root/reference/Launch_Optimiser.py, hand-written for MI355X (gfx950).  Replaces the
// m.solve() call at Launch_Optimiser.py:177 (GEKKO -> APMonitor -> IPOPT/MUMPS).
//
// Kernel structure (one lane = one NLP, see ascent_device.hpp):
//   k_solve       the whole interior-point loop; per iteration
//                   pass E  KKT error (optimality test, barrier update)
//                   pass B  evaluate defects/Jacobian/Hessian blocks of every step and factorise the
//                           bordered block-tridiagonal KKT system backwards in time (Riccati form)
//                   pass F  forward substitution: primal step, fraction-to-boundary, merit slope
//                   pass A  adjoint substitution: multiplier step, bound-multiplier steps
//                   pass T  merit function at trial points (backtracking)
//                   pass U  accept the step
//   k_eval_nodes  per-(step, problem) defects + Jacobian + Hessian blocks (parity surface)
//   k_kkt_step    one Newton step at a caller-supplied iterate (parity surface)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "ascent.h"
#include "ascent_device.hpp"

using namespace ascent;

namespace {

// ---------------------------------------------------------------------------------------------
// workspace: rows of `B` doubles each (B = batch rounded up to the wave size)
// ---------------------------------------------------------------------------------------------
struct Layout {
  int K;
  long B;
  long it, st;                         // iterate blob, step blob (21K+10 rows each)
  long sg, se, sf, sh;                 // per-step G[8], E[4], F[7], H[10]
  long rz, ru, gt, gu, cc;             // barrier-form dual residuals, tf-coupling column, defects
  long rr, qa, qm;                     // R_k, Sigma_angle, Sigma_mass
  long ka, k0;                         // Riccati gains kappa[7], kappa0[3]
  long total;
};

__host__ __device__ inline Layout make_layout(int K, long B) {
  Layout L;
  L.K = K;
  L.B = B;
  long o = 0;
  auto take = [&](long n) { long r = o; o += n; return r; };
  L.it = take(21L * K + NSC);
  L.st = take(21L * K + NSC);
  L.sg = take(8L * K);
  L.se = take(4L * K);
  L.sf = take(7L * K);
  L.sh = take(10L * K);
  L.rz = take(7L * K);
  L.ru = take(K);
  L.gt = take(7L * K);
  L.gu = take(K);
  L.cc = take(7L * K);
  L.rr = take(K);
  L.qa = take(K);
  L.qm = take(K);
  L.ka = take(7L * K);
  L.k0 = take(3L * K);
  L.total = o;
  return L;
}

// per-lane view of the workspace
struct Ctx {
  double *ws;  // already offset by the problem index
  long B;
  int K;
  double h;
  Layout L;
  Der d;
  ASC_DEV double &at(long off, long r) const { return ws[(off + r) * B]; }
  // iterate blob rows
  ASC_DEV double &z(int k, int f) const { return at(L.it, 7L * k + f); }
  ASC_DEV double &u(int k) const { return at(L.it, 7L * K + k); }
  ASC_DEV double &lam(int k, int f) const { return at(L.it, 8L * K + 7L * k + f); }
  ASC_DEV double &zb(int k, int b) const { return at(L.it, 15L * K + 6L * k + b); }
  ASC_DEV double &sc(int j) const { return at(L.it, 21L * K + j); }
  ASC_DEV double &dz(int k, int f) const { return at(L.st, 7L * k + f); }
  ASC_DEV double &du(int k) const { return at(L.st, 7L * K + k); }
  ASC_DEV double &dlam(int k, int f) const { return at(L.st, 8L * K + 7L * k + f); }
  ASC_DEV double &dzb(int k, int b) const { return at(L.st, 15L * K + 6L * k + b); }
  ASC_DEV double &dsc(int j) const { return at(L.st, 21L * K + j); }
};

// scalars of the iterate / step kept in registers
struct Scal {
  double th, zlt, zut, s1, s2, zs1, zs2, nu3, nu1, nu2;
};
ASC_DEV Scal load_scal(const Ctx &c, long off) {
  Scal s;
  const long b = 21L * c.K;
  s.th = c.at(off, b + S_TH); s.zlt = c.at(off, b + S_ZLT); s.zut = c.at(off, b + S_ZUT);
  s.s1 = c.at(off, b + S_S1); s.s2 = c.at(off, b + S_S2); s.zs1 = c.at(off, b + S_ZS1);
  s.zs2 = c.at(off, b + S_ZS2); s.nu3 = c.at(off, b + S_NU3); s.nu1 = c.at(off, b + S_NU1);
  s.nu2 = c.at(off, b + S_NU2);
  return s;
}
ASC_DEV void store_scal(const Ctx &c, long off, const Scal &s) {
  const long b = 21L * c.K;
  c.at(off, b + S_TH) = s.th; c.at(off, b + S_ZLT) = s.zlt; c.at(off, b + S_ZUT) = s.zut;
  c.at(off, b + S_S1) = s.s1; c.at(off, b + S_S2) = s.s2; c.at(off, b + S_ZS1) = s.zs1;
  c.at(off, b + S_ZS2) = s.zs2; c.at(off, b + S_NU3) = s.nu3; c.at(off, b + S_NU1) = s.nu1;
  c.at(off, b + S_NU2) = s.nu2;
}

// ---------------------------------------------------------------------------------------------
// pass E: optimality error pieces.  E(mu) = max(rd/sd, cinf, comp(mu)/sd) with
// comp(mu) = max(|pmax - mu|, |pmin - mu|) over all complementarity products.
// ---------------------------------------------------------------------------------------------
struct ErrParts {
  double rd, cinf, pmin, pmax, sd;
  ASC_DEV double err(double mu) const {
    const double comp = fmax(fabs(pmax - mu), fabs(pmin - mu));
    return fmax(fmax(rd / sd, cinf), comp / sd);
  }
};

ASC_DEV ErrParts pass_error(const Ctx &c, const Scal &s) {
  const Der &d = c.d;
  const int K = c.K;
  const double hT = c.h * d.T, dt = hT * s.th;
  double rd = 0.0, cinf = 0.0, pmin = 1e300, pmax = -1e300, l1 = 0.0, zsum = 0.0, rth = 1.0;
  double z[7], ln[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { z[i] = c.z(K - 1, i); ln[i] = 0.0; }
  for (int k = K - 1; k >= 0; k--) {
    double zp[7], l[7], zb[6], G[8], F[7], fl[7], ax, ay;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { zp[i] = k ? c.z(k - 1, i) : 0.0; l[i] = c.lam(k, i); }
    ASC_UNROLL
    for (int b = 0; b < 6; b++) zb[b] = c.zb(k, b);
    const double u = c.u(k);
    accel<1>(d, z[IX], z[IY], z[IA], z[IM], 0.0, 0.0, ax, ay, G, nullptr);
    rhs_f(d, z, u, ax, ay, F);
    fzt_lambda(G, l, fl);
    double r[7];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      r[i] = l[i] - dt * fl[i] - ln[i];
      rth -= hT * F[i] * l[i];
      l1 += fabs(l[i]);
      cinf = fmax(cinf, fabs(z[i] - zp[i] - dt * F[i]));
    }
    r[IA] += zb[1] - zb[0];
    r[IM] += zb[3] - zb[2];
    if (k == K - 1) {
      const Terminal t = terminal_eval(d, z);
      r[IX] += s.nu3 * t.e3g[0] + s.nu1 * t.g1g[0];
      r[IY] += s.nu3 * t.e3g[1] + s.nu1 * t.g1g[1];
      r[IVX] += s.nu3 * t.e3g[2] + s.nu2 * t.g2g[0];
      r[IVY] += s.nu3 * t.e3g[3] + s.nu2 * t.g2g[1];
      cinf = fmax(cinf, fmax(fabs(t.e3), fmax(fabs(t.g1 - s.s1), fabs(t.g2 - s.s2))));
    }
    ASC_UNROLL
    for (int i = 0; i < 7; i++) rd = fmax(rd, fabs(r[i]));
    rd = fmax(rd, fabs(-dt * d.alpha * l[IW] - zb[4] + zb[5]));
    const double lo[3] = {z[IA], z[IM], u + 1.0}, up[3] = {d.aub - z[IA], 1.0 - z[IM], 1.0 - u};
    ASC_UNROLL
    for (int b = 0; b < 3; b++) {
      const double p1 = lo[b] * zb[2 * b], p2 = up[b] * zb[2 * b + 1];
      pmin = fmin(pmin, fmin(p1, p2));
      pmax = fmax(pmax, fmax(p1, p2));
      zsum += zb[2 * b] + zb[2 * b + 1];
    }
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { ln[i] = l[i]; z[i] = zp[i]; }
  }
  rd = fmax(rd, fabs(rth - s.zlt + s.zut));
  rd = fmax(rd, fmax(fabs(-s.nu1 - s.zs1), fabs(-s.nu2 - s.zs2)));
  const double pr[4] = {(s.th - d.tlb) * s.zlt, (d.tub - s.th) * s.zut, s.s1 * s.zs1, s.s2 * s.zs2};
  ASC_UNROLL
  for (int j = 0; j < 4; j++) { pmin = fmin(pmin, pr[j]); pmax = fmax(pmax, pr[j]); }
  l1 += fabs(s.nu3) + fabs(s.nu1) + fabs(s.nu2);
  zsum += s.zlt + s.zut + s.zs1 + s.zs2;
  ErrParts e;
  e.rd = rd; e.cinf = cinf; e.pmin = pmin; e.pmax = pmax;
  e.sd = fmax(100.0, (l1 + zsum) / (double)(13 * K + 7)) * 0.01;
  return e;
}

// ---------------------------------------------------------------------------------------------
// pass B: evaluate + backward factorisation.  Returns 0, or 1 when the inertia is wrong.
// On success the border unknowns (dtheta, dnu3) are in ds; c1 = ||c||_1.
// ---------------------------------------------------------------------------------------------
struct BorderOut {
  double dth, dnu3, c1, sig1, sig2, rs1, rs2, sth;
};

ASC_DEV int pass_backward(const Ctx &c, const Scal &s, double mu, double dw, BorderOut &out) {
  const Der &d = c.d;
  const int K = c.K;
  const double hT = c.h * d.T, dt = hT * s.th, be = dt * d.alpha;
  double P[28], p0[7], p1[7], p2[7];
  ASC_UNROLL
  for (int i = 0; i < 28; i++) P[i] = 0.0;
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { p0[i] = p1[i] = p2[i] = 0.0; }
  double S10 = 0.0, S11 = 0.0, S12 = 0.0, S20 = 0.0, S22 = 0.0, rth = 1.0, c1 = 0.0;
  double z[7], ln[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) { z[i] = c.z(K - 1, i); ln[i] = 0.0; }
  Terminal tm = terminal_eval(d, z);
  const double sig1 = s.zs1 / s.s1 + dw, sig2 = s.zs2 / s.s2 + dw;
  const double rs1 = -mu / s.s1 - s.nu1, rs2 = -mu / s.s2 - s.nu2;
  const double cg1 = tm.g1 - s.s1, cg2 = tm.g2 - s.s2;
  c1 = fabs(tm.e3) + fabs(cg1) + fabs(cg2);
  for (int k = K - 1; k >= 0; k--) {
    double zp[7], l[7], zb[6], G[8], E[4], H[10], F[7], fl[7], ax, ay;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { zp[i] = k ? c.z(k - 1, i) : 0.0; l[i] = c.lam(k, i); }
    ASC_UNROLL
    for (int b = 0; b < 6; b++) zb[b] = c.zb(k, b);
    const double u = c.u(k);
    accel<2>(d, z[IX], z[IY], z[IA], z[IM], -dt * l[IVX], -dt * l[IVY], ax, ay, G, H);
    rhs_f(d, z, u, ax, ay, F);
    implicit_block(G, dt, E);
    fzt_lambda(G, l, fl);
    double rz[7], gt[7], cc[7];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      rz[i] = l[i] - dt * fl[i] - ln[i];
      gt[i] = -hT * fl[i];
      cc[i] = z[i] - zp[i] - dt * F[i];
      c1 += fabs(cc[i]);
      rth -= hT * F[i] * l[i];
    }
    const double a = z[IA], m = z[IM];
    rz[IA] += -mu / a + mu / (d.aub - a);
    rz[IM] += -mu / m + mu / (1.0 - m);
    const double ru = -be * l[IW] - mu / (u + 1.0) + mu / (1.0 - u);
    const double gu = -hT * d.alpha * l[IW];
    const double R = zb[4] / (u + 1.0) + zb[5] / (1.0 - u) + dw;
    const double qa = zb[0] / a + zb[1] / (d.aub - a), qm = zb[2] / m + zb[3] / (1.0 - m);
    // N = Q_k + P_{k+1}
    double N[28];
    ASC_UNROLL
    for (int i = 0; i < 28; i++) N[i] = P[i];
    N[sid(IX, IX)] += H[0]; N[sid(IX, IY)] += H[1]; N[sid(IX, IA)] += H[2]; N[sid(IX, IM)] += H[3];
    N[sid(IY, IY)] += H[4]; N[sid(IY, IA)] += H[5]; N[sid(IY, IM)] += H[6];
    N[sid(IA, IA)] += H[7] + qa; N[sid(IA, IM)] += H[8]; N[sid(IM, IM)] += H[9] + qm;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) N[sid(i, i)] += dw;
    if (k == K - 1) {
      terminal_hessian(N, tm, s.nu3, s.nu1, s.nu2, sig1, sig2);
      const double w1 = s.nu1 + sig1 * cg1 + rs1, w2 = s.nu2 + sig2 * cg2 + rs2;
      rz[IX] += s.nu3 * tm.e3g[0] + w1 * tm.g1g[0];
      rz[IY] += s.nu3 * tm.e3g[1] + w1 * tm.g1g[1];
      rz[IVX] += s.nu3 * tm.e3g[2] + w2 * tm.g2g[0];
      rz[IVY] += s.nu3 * tm.e3g[3] + w2 * tm.g2g[1];
    }
    // stage data needed by the forward / adjoint passes
    ASC_UNROLL
    for (int i = 0; i < 8; i++) c.at(c.L.sg, 8L * k + i) = G[i];
    ASC_UNROLL
    for (int i = 0; i < 4; i++) c.at(c.L.se, 4L * k + i) = E[i];
    ASC_UNROLL
    for (int i = 0; i < 10; i++) c.at(c.L.sh, 10L * k + i) = H[i];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      c.at(c.L.sf, 7L * k + i) = F[i];
      c.at(c.L.rz, 7L * k + i) = rz[i];
      c.at(c.L.gt, 7L * k + i) = gt[i];
      c.at(c.L.cc, 7L * k + i) = cc[i];
    }
    c.at(c.L.ru, k) = ru; c.at(c.L.gu, k) = gu; c.at(c.L.rr, k) = R;
    c.at(c.L.qa, k) = qa; c.at(c.L.qm, k) = qm;
    // M = A^-T N A^-1 (in place), pivot, gain, P_k
    congruence(N, G, E, dt);
    const double D = R + be * be * N[sid(IW, IW)];
    if (!(D > 0.0)) return 1;
    const double iD = 1.0 / D;
    double mw[7], kap[7];
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { mw[i] = be * N[sid(i, IW)]; kap[i] = mw[i] * iD; }
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      ASC_UNROLL
      for (int j = i; j < 7; j++) P[sid(i, j)] = N[sid(i, j)] - mw[i] * kap[j];
    }
    ASC_UNROLL
    for (int i = 0; i < 7; i++) c.at(c.L.ka, 7L * k + i) = kap[i];
    // three right-hand sides (0: residual, 1: -B_theta, 2: -B_nu3)
    double n[7], nt[7], q0[7], q1[7], q2[7], k00, k01, k02, rc1[7], Prc[7];
    // rhs 0
    ASC_UNROLL
    for (int i = 0; i < 7; i++) n[i] = -rz[i] + p0[i];
    solveAT(G, E, dt, n, nt);
    k00 = (be * nt[IW] - ru) * iD;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { q0[i] = nt[i] - mw[i] * k00; n[i] = -cc[i]; }
    symv(P, n, Prc);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) p0[i] = q0[i] - Prc[i];
    // rhs 1
    ASC_UNROLL
    for (int i = 0; i < 7; i++) n[i] = -gt[i] + p1[i];
    solveAT(G, E, dt, n, nt);
    k01 = (be * nt[IW] - gu) * iD;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { q1[i] = nt[i] - mw[i] * k01; rc1[i] = hT * F[i]; }
    symv(P, rc1, Prc);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) p1[i] = q1[i] - Prc[i];
    // rhs 2 (only the last node has a direct term; no defect / control part)
    ASC_UNROLL
    for (int i = 0; i < 7; i++) n[i] = p2[i];
    if (k == K - 1) { n[IX] -= tm.e3g[0]; n[IY] -= tm.e3g[1]; n[IVX] -= tm.e3g[2]; n[IVY] -= tm.e3g[3]; }
    solveAT(G, E, dt, n, nt);
    k02 = be * nt[IW] * iD;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { q2[i] = nt[i] - mw[i] * k02; p2[i] = q2[i]; }
    c.at(c.L.k0, 3L * k) = k00; c.at(c.L.k0, 3L * k + 1) = k01; c.at(c.L.k0, 3L * k + 2) = k02;
    // Schur-complement entries S_ij = rho_i' K0^-1 rho_j accumulated stage by stage
    double a10 = D * k01 * k00, a11 = D * k01 * k01, a12 = D * k01 * k02, a20 = D * k02 * k00,
           a22 = D * k02 * k02;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      a10 += 0.5 * (rc1[i] * (q0[i] + p0[i]) - cc[i] * (q1[i] + p1[i]));
      a11 += rc1[i] * (q1[i] + p1[i]);
      a12 += 0.5 * rc1[i] * (q2[i] + p2[i]);
      a20 += -0.5 * cc[i] * (q2[i] + p2[i]);
    }
    S10 += a10; S11 += a11; S12 += a12; S20 += a20; S22 += a22;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { ln[i] = l[i]; z[i] = zp[i]; }
  }
  rth += -mu / (s.th - d.tlb) + mu / (d.tub - s.th);
  const double sth = s.zlt / (s.th - d.tlb) + s.zut / (d.tub - s.th) + dw;
  const double a11 = sth - S11, a12 = -S12, a22 = -S22;
  const double b1 = -rth + S10, b2 = -tm.e3 + S20;
  const double det = a11 * a22 - a12 * a12;
  if (!(det < 0.0)) return 1;
  out.dth = (b1 * a22 - a12 * b2) / det;
  out.dnu3 = (a11 * b2 - a12 * b1) / det;
  out.c1 = c1; out.sig1 = sig1; out.sig2 = sig2; out.rs1 = rs1; out.rs2 = rs2; out.sth = sth;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// pass F: forward substitution (primal step), primal fraction-to-boundary, barrier slope
// ---------------------------------------------------------------------------------------------
#define ASC_FTB(a, val, dv) do { const double dv_ = (dv); if (dv_ < 0.0) a = fmin(a, -tau * (val) / dv_); } while (0)

ASC_DEV void pass_forward(const Ctx &c, const Scal &s, double mu, double tau, double dth,
                          double dnu3, double &apr, double &gd, double *dzK) {
  const Der &d = c.d;
  const int K = c.K;
  const double hT = c.h * d.T, dt = hT * s.th, be = dt * d.alpha;
  double dzp[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) dzp[i] = 0.0;
  for (int k = 0; k < K; k++) {
    double G[8], E[4], xi[7], dz[7];
    ASC_UNROLL
    for (int i = 0; i < 8; i++) G[i] = c.at(c.L.sg, 8L * k + i);
    ASC_UNROLL
    for (int i = 0; i < 4; i++) E[i] = c.at(c.L.se, 4L * k + i);
    double du = c.at(c.L.k0, 3L * k) + c.at(c.L.k0, 3L * k + 1) * dth + c.at(c.L.k0, 3L * k + 2) * dnu3;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      xi[i] = dzp[i] - c.at(c.L.cc, 7L * k + i) + hT * c.at(c.L.sf, 7L * k + i) * dth;
      du -= c.at(c.L.ka, 7L * k + i) * xi[i];
    }
    xi[IW] += be * du;
    solveA(G, E, dt, xi, dz);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { c.dz(k, i) = dz[i]; dzp[i] = dz[i]; }
    c.du(k) = du;
    const double a = c.z(k, IA), m = c.z(k, IM), u = c.u(k);
    ASC_FTB(apr, a, dz[IA]); ASC_FTB(apr, d.aub - a, -dz[IA]);
    ASC_FTB(apr, m, dz[IM]); ASC_FTB(apr, 1.0 - m, -dz[IM]);
    ASC_FTB(apr, u + 1.0, du); ASC_FTB(apr, 1.0 - u, -du);
    gd += dz[IA] * (-mu / a + mu / (d.aub - a)) + dz[IM] * (-mu / m + mu / (1.0 - m)) +
          du * (-mu / (u + 1.0) + mu / (1.0 - u));
  }
  ASC_UNROLL
  for (int i = 0; i < 7; i++) dzK[i] = dzp[i];
}

// ---------------------------------------------------------------------------------------------
// pass A: adjoint substitution (multiplier step), bound-multiplier steps, dual fraction-to-boundary,
// and c'(lambda + dlambda) for the curvature estimate
// ---------------------------------------------------------------------------------------------
ASC_DEV void pass_adjoint(const Ctx &c, const Scal &s, double mu, double dw, double tau, double dth,
                          double dnu3, double sig1, double sig2, double &adu, double &cl) {
  const Der &d = c.d;
  const int K = c.K;
  const double hT = c.h * d.T, dt = hT * s.th;
  (void)hT;
  double dln[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) dln[i] = 0.0;
  for (int k = K - 1; k >= 0; k--) {
    double G[8], E[4], H[10], dz[7], r[7], dl[7];
    ASC_UNROLL
    for (int i = 0; i < 8; i++) G[i] = c.at(c.L.sg, 8L * k + i);
    ASC_UNROLL
    for (int i = 0; i < 4; i++) E[i] = c.at(c.L.se, 4L * k + i);
    ASC_UNROLL
    for (int i = 0; i < 10; i++) H[i] = c.at(c.L.sh, 10L * k + i);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) dz[i] = c.dz(k, i);
    const double qa = c.at(c.L.qa, k), qm = c.at(c.L.qm, k);
    ASC_UNROLL
    for (int i = 0; i < 7; i++)
      r[i] = -c.at(c.L.rz, 7L * k + i) - c.at(c.L.gt, 7L * k + i) * dth + dln[i] - dw * dz[i];
    r[IX] -= H[0] * dz[IX] + H[1] * dz[IY] + H[2] * dz[IA] + H[3] * dz[IM];
    r[IY] -= H[1] * dz[IX] + H[4] * dz[IY] + H[5] * dz[IA] + H[6] * dz[IM];
    r[IA] -= H[2] * dz[IX] + H[5] * dz[IY] + (H[7] + qa) * dz[IA] + H[8] * dz[IM];
    r[IM] -= H[3] * dz[IX] + H[6] * dz[IY] + H[8] * dz[IA] + (H[9] + qm) * dz[IM];
    if (k == K - 1) {
      double zK[7], QT[28], qd[7];
      ASC_UNROLL
      for (int i = 0; i < 7; i++) zK[i] = c.z(k, i);
      const Terminal tm = terminal_eval(d, zK);
      ASC_UNROLL
      for (int i = 0; i < 28; i++) QT[i] = 0.0;
      terminal_hessian(QT, tm, s.nu3, s.nu1, s.nu2, sig1, sig2);
      symv(QT, dz, qd);
      ASC_UNROLL
      for (int i = 0; i < 7; i++) r[i] -= qd[i];
      r[IX] -= tm.e3g[0] * dnu3; r[IY] -= tm.e3g[1] * dnu3;
      r[IVX] -= tm.e3g[2] * dnu3; r[IVY] -= tm.e3g[3] * dnu3;
    }
    solveAT(G, E, dt, r, dl);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      c.dlam(k, i) = dl[i];
      dln[i] = dl[i];
      cl += c.at(c.L.cc, 7L * k + i) * (c.lam(k, i) + dl[i]);
    }
    // bound multipliers of angle, mass, u
    const double a = c.z(k, IA), m = c.z(k, IM), u = c.u(k), du = c.du(k);
    const double lo[3] = {a, m, u + 1.0}, up[3] = {d.aub - a, 1.0 - m, 1.0 - u};
    const double dx[3] = {dz[IA], dz[IM], du};
    ASC_UNROLL
    for (int b = 0; b < 3; b++) {
      const double zl = c.zb(k, 2 * b), zu = c.zb(k, 2 * b + 1);
      const double dzl = mu / lo[b] - zl - zl / lo[b] * dx[b];
      const double dzu = mu / up[b] - zu + zu / up[b] * dx[b];
      c.dzb(k, 2 * b) = dzl;
      c.dzb(k, 2 * b + 1) = dzu;
      ASC_FTB(adu, zl, dzl);
      ASC_FTB(adu, zu, dzu);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// pass T: l1 merit function at the trial point iterate + alpha*step
// ---------------------------------------------------------------------------------------------
ASC_DEV double pass_trial(const Ctx &c, const Scal &s, const Scal &ds, double alpha, double mu,
                          double nu_pen) {
  const Der &d = c.d;
  const int K = c.K;
  const double th = s.th + alpha * ds.th, s1 = s.s1 + alpha * ds.s1, s2 = s.s2 + alpha * ds.s2;
  const double dt = c.h * d.T * th;
  double sl = log(th - d.tlb) + log(d.tub - th) + log(s1) + log(s2);
  double c1 = 0.0, zp[7], z[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) zp[i] = 0.0;
  for (int k = 0; k < K; k++) {
    double F[7], ax, ay;
    ASC_UNROLL
    for (int i = 0; i < 7; i++) z[i] = c.z(k, i) + alpha * c.dz(k, i);
    const double u = c.u(k) + alpha * c.du(k);
    accel<0>(d, z[IX], z[IY], z[IA], z[IM], 0.0, 0.0, ax, ay, nullptr, nullptr);
    rhs_f(d, z, u, ax, ay, F);
    ASC_UNROLL
    for (int i = 0; i < 7; i++) { c1 += fabs(z[i] - zp[i] - dt * F[i]); zp[i] = z[i]; }
    sl += log(z[IA]) + log(d.aub - z[IA]) + log(z[IM]) + log(1.0 - z[IM]) + log(u + 1.0) + log(1.0 - u);
  }
  const Terminal tm = terminal_eval(d, z);
  c1 += fabs(tm.e3) + fabs(tm.g1 - s1) + fabs(tm.g2 - s2);
  return th - mu * sl + nu_pen * c1;
}

ASC_DEV double barrier_now(const Ctx &c, const Scal &s, double mu) {
  const Der &d = c.d;
  double sl = log(s.th - d.tlb) + log(d.tub - s.th) + log(s.s1) + log(s.s2);
  for (int k = 0; k < c.K; k++) {
    const double a = c.z(k, IA), m = c.z(k, IM), u = c.u(k);
    sl += log(a) + log(d.aub - a) + log(m) + log(1.0 - m) + log(u + 1.0) + log(1.0 - u);
  }
  return s.th - mu * sl;
}

// ---------------------------------------------------------------------------------------------
// pass U: accept the step
// ---------------------------------------------------------------------------------------------
ASC_DEV double clipz(double zv, double dist, double mu) {
  return fmin(fmax(zv, mu / (1e10 * dist)), 1e10 * mu / dist);
}

ASC_DEV void pass_update(const Ctx &c, double alpha, double adu, double mu) {
  const Der &d = c.d;
  const int K = c.K;
  for (int k = 0; k < K; k++) {
    ASC_UNROLL
    for (int i = 0; i < 7; i++) {
      c.z(k, i) += alpha * c.dz(k, i);
      c.lam(k, i) += alpha * c.dlam(k, i);
    }
    c.u(k) += alpha * c.du(k);
    const double a = c.z(k, IA), m = c.z(k, IM), u = c.u(k);
    const double dist[6] = {a, d.aub - a, m, 1.0 - m, u + 1.0, 1.0 - u};
    ASC_UNROLL
    for (int b = 0; b < 6; b++) c.zb(k, b) = clipz(c.zb(k, b) + adu * c.dzb(k, b), dist[b], mu);
  }
}

// ---------------------------------------------------------------------------------------------
// initial point
// ---------------------------------------------------------------------------------------------
ASC_DEV double push_in(double v, double lb, double ub) {
  const double k1 = 1e-2;
  const double pl = fmin(k1 * fmax(1.0, fabs(lb)), k1 * (ub - lb));
  const double pu = fmin(k1 * fmax(1.0, fabs(ub)), k1 * (ub - lb));
  return fmin(fmax(v, lb + pl), ub - pu);
}

// straight-line states toward a tangential insertion point, u = 0 (cold start)
ASC_DEV void cold_guess(const Ctx &c, Scal &s) {
  const Der &d = c.d;
  const int K = c.K;
  const double tf0 = 0.9, dr = 0.166, aend = 0.5, vp = sqrt(d.vp2), dt = c.h * d.T * tf0;
  const double sdr = sin(dr), cdr = cos(dr);
  const double xf = -d.rhof * sdr, yf = d.rhof * cdr - d.rho0;
  for (int k = 0; k < K; k++) {
    const double fr = (double)(k + 1) / K;
    c.z(k, IX) = fr * xf; c.z(k, IY) = fr * yf;
    c.z(k, IVX) = -fr * vp * cdr; c.z(k, IVY) = -fr * vp * sdr;
    c.z(k, IA) = fr * aend; c.z(k, IW) = aend / (K * dt); c.z(k, IM) = d.mrate * dt * (k + 1);
    c.u(k) = 0.0;
  }
  s.th = tf0;
}

// interior point + multipliers. mode 0/1: primal only (multipliers reset); 2: keep multipliers
ASC_DEV void init_point(const Ctx &c, Scal &s, int mode) {
  const Der &d = c.d;
  const int K = c.K;
  for (int k = 0; k < K; k++) {
    c.z(k, IA) = push_in(c.z(k, IA), 0.0, d.aub);
    c.z(k, IM) = push_in(c.z(k, IM), 0.0, 1.0);
    c.u(k) = push_in(c.u(k), -1.0, 1.0);
    if (mode != 2) {
      ASC_UNROLL
      for (int b = 0; b < 6; b++) c.zb(k, b) = 1.0;
      ASC_UNROLL
      for (int i = 0; i < 7; i++) c.lam(k, i) = 0.0;
    } else {
      ASC_UNROLL
      for (int b = 0; b < 6; b++) c.zb(k, b) = fmax(c.zb(k, b), 1e-12);
    }
  }
  s.th = push_in(s.th, d.tlb, d.tub);
  double zK[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) zK[i] = c.z(K - 1, i);
  const Terminal tm = terminal_eval(d, zK);
  if (mode != 2) {
    s.s1 = fmax(tm.g1, 1e-2); s.s2 = fmax(tm.g2, 1e-2);
    s.zlt = s.zut = s.zs1 = s.zs2 = 1.0;
    s.nu3 = s.nu1 = s.nu2 = 0.0;
  } else {
    s.s1 = fmax(s.s1, 1e-10); s.s2 = fmax(s.s2, 1e-10);
    s.zlt = fmax(s.zlt, 1e-12); s.zut = fmax(s.zut, 1e-12);
    s.zs1 = fmax(s.zs1, 1e-12); s.zs2 = fmax(s.zs2, 1e-12);
  }
}

// Newton step at the current iterate: passes B, F, A.  Returns 0 / 1 (wrong inertia).
struct StepInfo { double apr, adu, gd, cl, c1; };

ASC_DEV int newton_step(const Ctx &c, const Scal &s, double mu, double dw, Scal &ds, StepInfo &si) {
  BorderOut bo;
  if (pass_backward(c, s, mu, dw, bo)) return 1;
  const Der &d = c.d;
  const double tau = fmax(0.99, 1.0 - mu);
  double apr = 1.0, adu = 1.0, gd = 0.0, cl = 0.0, dzK[7];
  pass_forward(c, s, mu, tau, bo.dth, bo.dnu3, apr, gd, dzK);
  pass_adjoint(c, s, mu, dw, tau, bo.dth, bo.dnu3, bo.sig1, bo.sig2, adu, cl);
  // slacks, their multipliers, tf bounds
  double zK[7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) zK[i] = c.z(c.K - 1, i);
  const Terminal tm = terminal_eval(d, zK);
  ds.th = bo.dth; ds.nu3 = bo.dnu3;
  ds.s1 = (tm.g1 - s.s1) + tm.g1g[0] * dzK[IX] + tm.g1g[1] * dzK[IY];
  ds.s2 = (tm.g2 - s.s2) + tm.g2g[0] * dzK[IVX] + tm.g2g[1] * dzK[IVY];
  ds.nu1 = bo.sig1 * ds.s1 + bo.rs1;
  ds.nu2 = bo.sig2 * ds.s2 + bo.rs2;
  ds.zs1 = mu / s.s1 - s.zs1 - s.zs1 / s.s1 * ds.s1;
  ds.zs2 = mu / s.s2 - s.zs2 - s.zs2 / s.s2 * ds.s2;
  const double dl = s.th - d.tlb, dU = d.tub - s.th;
  ds.zlt = mu / dl - s.zlt - s.zlt / dl * ds.th;
  ds.zut = mu / dU - s.zut + s.zut / dU * ds.th;
  ASC_FTB(apr, dl, ds.th); ASC_FTB(apr, dU, -ds.th);
  ASC_FTB(apr, s.s1, ds.s1); ASC_FTB(apr, s.s2, ds.s2);
  ASC_FTB(adu, s.zlt, ds.zlt); ASC_FTB(adu, s.zut, ds.zut);
  ASC_FTB(adu, s.zs1, ds.zs1); ASC_FTB(adu, s.zs2, ds.zs2);
  gd += ds.th * (1.0 - mu / dl + mu / dU) - mu * ds.s1 / s.s1 - mu * ds.s2 / s.s2;
  cl += tm.e3 * (s.nu3 + ds.nu3) + (tm.g1 - s.s1) * (s.nu1 + ds.nu1) + (tm.g2 - s.s2) * (s.nu2 + ds.nu2);
  si.apr = apr; si.adu = adu; si.gd = gd; si.cl = cl; si.c1 = bo.c1;
  return 0;
}

ASC_DEV Ctx make_ctx(double *ws, const Layout &L, long p, const ascent_params *params) {
  Ctx c;
  c.ws = ws + p;
  c.B = L.B;
  c.K = L.K;
  c.h = 1.0 / L.K;
  c.L = L;
  c.d = derive(params[p]);
  return c;
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_solve(const ascent_params *params, long batch, Layout L,
                                              double *ws, const double *guess, int warm, int max_iter,
                                              double tol, double mu_init, double *traj, double *tf_out,
                                              int *status_out, int *iters_out, double *blob_out) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= batch) return;
  const Ctx c = make_ctx(ws, L, p, params);
  const int K = c.K;
  const long rows = 21L * K + NSC;
  Scal s;
  if (warm) {
    for (long r = 0; r < rows; r++) c.at(L.it, r) = guess[r * batch + p];
    s = load_scal(c, L.it);
  } else {
    cold_guess(c, s);
  }
  init_point(c, s, warm);
  double mu = mu_init, nu_pen = 1.0, dw_last = 0.0;
  int status = ASCENT_MAX_ITER, iters = 0;
  for (int iter = 0; iter < max_iter; iter++) {
    const ErrParts e = pass_error(c, s);
    if (e.err(0.0) <= tol) { status = ASCENT_CONVERGED; break; }
    while (mu > tol * 0.1 && e.err(mu) <= 10.0 * mu) {
      mu = fmax(tol * 0.1, fmin(0.2 * mu, mu * sqrt(mu)));
      nu_pen = 1.0;
    }
    double dw = 0.0;
    Scal ds;
    StepInfo si;
    bool fail = false;
    while (newton_step(c, s, mu, dw, ds, si)) {
      dw = dw == 0.0 ? fmax(1e-4, dw_last / 3.0) : dw * 8.0;
      if (dw > 1e10) { fail = true; break; }
    }
    if (fail) { status = ASCENT_REGULARISATION_FAILED; break; }
    dw_last = dw;
    const double curv = -si.gd + si.cl;
    if (si.c1 > 0.0) {
      const double need = (si.gd + 0.5 * fmax(curv, 0.0)) / (0.9 * si.c1);
      if (nu_pen < need) nu_pen = need + 1.0;
    }
    const double Dm = si.gd - nu_pen * si.c1;
    const double phi0 = barrier_now(c, s, mu) + nu_pen * si.c1;
    double alpha = si.apr;
    bool ok = false;
    for (int ls = 0; ls < 40; ls++) {
      const double phit = pass_trial(c, s, ds, alpha, mu, nu_pen);
      if (isfinite(phit) && phit <= phi0 + 1e-8 * alpha * Dm + 2.220446049250313e-15 * fabs(phi0)) { ok = true; break; }
      alpha *= 0.5;
    }
    if (!ok) { status = ASCENT_LINESEARCH_FAILED; break; }
    pass_update(c, alpha, si.adu, mu);
    s.th += alpha * ds.th; s.s1 += alpha * ds.s1; s.s2 += alpha * ds.s2;
    s.nu3 += alpha * ds.nu3; s.nu1 += alpha * ds.nu1; s.nu2 += alpha * ds.nu2;
    s.zlt = clipz(s.zlt + si.adu * ds.zlt, s.th - c.d.tlb, mu);
    s.zut = clipz(s.zut + si.adu * ds.zut, c.d.tub - s.th, mu);
    s.zs1 = clipz(s.zs1 + si.adu * ds.zs1, s.s1, mu);
    s.zs2 = clipz(s.zs2 + si.adu * ds.zs2, s.s2, mu);
    iters = iter + 1;
  }
  store_scal(c, L.it, s);
  tf_out[p] = s.th;
  status_out[p] = status;
  iters_out[p] = iters;
  if (blob_out)
    for (long r = 0; r < rows; r++) blob_out[r * batch + p] = c.at(L.it, r);
  if (traj) {
    const int nt = K + 1;
    for (int k = 0; k < nt; k++) {
      double z[7], u = 0.0, ax, ay;
      ASC_UNROLL
      for (int i = 0; i < 7; i++) z[i] = k ? c.z(k - 1, i) : 0.0;
      if (k) u = c.u(k - 1);
      accel<0>(c.d, z[IX], z[IY], z[IA], z[IM], 0.0, 0.0, ax, ay, nullptr, nullptr);
      const double v[10] = {z[IX], z[IY], z[IVX], z[IVY], ax, ay, z[IA], z[IW], u, z[IM]};
      ASC_UNROLL
      for (int f = 0; f < 10; f++) traj[((long)f * nt + k) * batch + p] = v[f];
    }
  }
}

// thread = (problem, step): defects, Jacobian and Hessian blocks of one collocation step
__global__ __launch_bounds__(256) void k_eval_nodes(const ascent_params *params, long batch, int K,
                                                    const double *it, double *defects, double *jac,
                                                    double *hess) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (p >= batch) return;
  const Der d = derive(params[p]);
  const double th = it[(21L * K + S_TH) * batch + p];
  const double dt = (1.0 / K) * d.T * th;
  double z[7], zp[7], G[8], H[10], F[7], ax, ay;
  ASC_UNROLL
  for (int i = 0; i < 7; i++) {
    z[i] = it[(7L * k + i) * batch + p];
    zp[i] = k ? it[(7L * (k - 1) + i) * batch + p] : 0.0;
  }
  const double u = it[(7L * K + k) * batch + p];
  const double lvx = it[(8L * K + 7L * k + IVX) * batch + p], lvy = it[(8L * K + 7L * k + IVY) * batch + p];
  accel<2>(d, z[IX], z[IY], z[IA], z[IM], -dt * lvx, -dt * lvy, ax, ay, G, H);
  rhs_f(d, z, u, ax, ay, F);
  ASC_UNROLL
  for (int i = 0; i < 7; i++) defects[(7L * k + i) * batch + p] = z[i] - zp[i] - dt * F[i];
  ASC_UNROLL
  for (int i = 0; i < 8; i++) jac[(8L * k + i) * batch + p] = G[i];
  ASC_UNROLL
  for (int i = 0; i < 10; i++) hess[(10L * k + i) * batch + p] = H[i];
}

__global__ __launch_bounds__(64) void k_kkt_step(const ascent_params *params, long batch, Layout L,
                                                 double *ws, const double *it, const double *mu,
                                                 const double *dw, double *step, int *inertia) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= batch) return;
  const Ctx c = make_ctx(ws, L, p, params);
  const long rows = 21L * c.K + NSC;
  for (long r = 0; r < rows; r++) c.at(L.it, r) = it[r * batch + p];
  const Scal s = load_scal(c, L.it);
  Scal ds;
  StepInfo si;
  const int rc = newton_step(c, s, mu[p], dw[p], ds, si);
  inertia[p] = rc;
  if (rc == 0) {
    store_scal(c, L.st, ds);
    for (long r = 0; r < rows; r++) step[r * batch + p] = c.at(L.st, r);
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";

int hip_fail(hipError_t e, const char *what) {
  snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
  return ASCENT_E_HIP;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(e_, #call); } while (0)

struct DeviceWs {
  double *ws = nullptr;
  size_t bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool launched = false;
};
constexpr int MAX_DEV = 64;
DeviceWs g_ws[MAX_DEV];
std::mutex g_mu[MAX_DEV];

int ensure_ws(int dev, size_t bytes) {
  DeviceWs &w = g_ws[dev];
  if (!w.ev0) {
    HIPCHK(hipEventCreate(&w.ev0));
    HIPCHK(hipEventCreate(&w.ev1));
  }
  if (w.bytes >= bytes) return 0;
  if (w.ws) HIPCHK(hipFree(w.ws));
  w.ws = nullptr;
  w.bytes = 0;
  hipError_t e = hipMalloc(&w.ws, bytes);
  if (e != hipSuccess) {
    snprintf(g_err, sizeof g_err, "workspace hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    return ASCENT_E_NOMEM;
  }
  w.bytes = bytes;
  return 0;
}

int check_common(const ascent_params *p, int64_t batch, const ascent_opts *o, int device_id) {
  if (!p || !o || batch <= 0) { snprintf(g_err, sizeof g_err, "null params/opts or batch <= 0"); return ASCENT_E_ARG; }
  if (o->n_nodes < 3 || o->n_nodes > 100000) { snprintf(g_err, sizeof g_err, "n_nodes out of range"); return ASCENT_E_ARG; }
  if (o->scheme != 0) { snprintf(g_err, sizeof g_err, "scheme %d not supported (0 = backward Euler, the reference's NODES=2)", o->scheme); return ASCENT_E_ARG; }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { snprintf(g_err, sizeof g_err, "no HIP device available"); return ASCENT_E_NODEVICE; }
  if (device_id < 0 || device_id >= n || device_id >= MAX_DEV) { snprintf(g_err, sizeof g_err, "device %d of %d", device_id, n); return ASCENT_E_NODEVICE; }
  return 0;
}

template <typename T>
struct DevBuf {  // device staging buffer for host-pointer calls
  T *d = nullptr;
  ~DevBuf() { if (d) (void)hipFree(d); }
  hipError_t alloc(size_t n) { return hipMalloc(&d, n * sizeof(T)); }
};

}  // namespace

extern "C" {

int ascent_version(void) { return 100; }

int ascent_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *ascent_strerror(int code) {
  switch (code) {
    case ASCENT_OK: return "ok";
    case ASCENT_E_ARG: case ASCENT_E_HIP: case ASCENT_E_NODEVICE: case ASCENT_E_NOMEM:
      return g_err[0] ? g_err : "error";
    default: return "unknown error code";
  }
}

double ascent_last_kernel_ms(int device_id) {
  if (device_id < 0 || device_id >= MAX_DEV) return -1.0;
  std::lock_guard<std::mutex> lock(g_mu[device_id]);
  DeviceWs &w = g_ws[device_id];
  if (!w.launched) return -1.0;
  if (hipSetDevice(device_id) != hipSuccess) return -1.0;
  if (hipEventSynchronize(w.ev1) != hipSuccess) return -1.0;
  float ms = -1.f;
  if (hipEventElapsedTime(&ms, w.ev0, w.ev1) != hipSuccess) return -1.0;
  return ms;
}

int ascent_solve_batch(const ascent_params *p, int64_t batch, const ascent_opts *o,
                       const double *guess, double *traj_out, double *tf_out, int32_t *status_out,
                       int32_t *iters_out, double *sol_blob_out, int device_id, void *stream_,
                       int ptr_is_device) {
  int rc = check_common(p, batch, o, device_id);
  if (rc) return rc;
  if (!tf_out || !status_out || !iters_out) { snprintf(g_err, sizeof g_err, "null output pointer"); return ASCENT_E_ARG; }
  if (o->warm_start < 0 || o->warm_start > 2 || (o->warm_start && !guess)) { snprintf(g_err, sizeof g_err, "warm_start needs a guess blob"); return ASCENT_E_ARG; }
  if (!(o->tol > 0) || o->max_iter < 0) { snprintf(g_err, sizeof g_err, "tol must be > 0, max_iter >= 0"); return ASCENT_E_ARG; }
  std::lock_guard<std::mutex> lock(g_mu[device_id]);
  HIPCHK(hipSetDevice(device_id));
  hipStream_t stream = (hipStream_t)stream_;
  const int K = o->n_nodes - 1, nt = o->n_nodes;
  const long rows = 21L * K + NSC;
  const Layout L = make_layout(K, (long)batch);
  rc = ensure_ws(device_id, (size_t)L.total * (size_t)batch * sizeof(double));
  if (rc) return rc;
  DeviceWs &w = g_ws[device_id];
  const double mu0 = o->mu_init > 0 ? o->mu_init : (o->warm_start ? 1e-4 : 0.1);

  const ascent_params *dp = p;
  const double *dguess = guess;
  double *dtraj = traj_out, *dtf = tf_out, *dblob = sol_blob_out;
  int *dstatus = status_out, *diters = iters_out;
  DevBuf<ascent_params> bp;
  DevBuf<double> bguess, btraj, btf, bblob;
  DevBuf<int> bstatus, biters;
  if (!ptr_is_device) {
    HIPCHK(bp.alloc(batch));
    HIPCHK(hipMemcpyAsync(bp.d, p, batch * sizeof(ascent_params), hipMemcpyHostToDevice, stream));
    dp = bp.d;
    if (o->warm_start) {
      HIPCHK(bguess.alloc(rows * batch));
      HIPCHK(hipMemcpyAsync(bguess.d, guess, rows * batch * sizeof(double), hipMemcpyHostToDevice, stream));
      dguess = bguess.d;
    }
    if (traj_out) { HIPCHK(btraj.alloc((size_t)10 * nt * batch)); dtraj = btraj.d; }
    if (sol_blob_out) { HIPCHK(bblob.alloc(rows * batch)); dblob = bblob.d; }
    HIPCHK(btf.alloc(batch)); dtf = btf.d;
    HIPCHK(bstatus.alloc(batch)); dstatus = bstatus.d;
    HIPCHK(biters.alloc(batch)); diters = biters.d;
  }
  const unsigned grid = (unsigned)((batch + 63) / 64);
  HIPCHK(hipEventRecord(w.ev0, stream));
  hipLaunchKernelGGL(k_solve, dim3(grid), dim3(64), 0, stream, dp, (long)batch, L, w.ws, dguess,
                     (int)o->warm_start, (int)o->max_iter, o->tol, mu0, dtraj, dtf, dstatus, diters, dblob);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(w.ev1, stream));
  w.launched = true;
  if (!ptr_is_device) {
    if (traj_out) HIPCHK(hipMemcpyAsync(traj_out, dtraj, (size_t)10 * nt * batch * sizeof(double), hipMemcpyDeviceToHost, stream));
    if (sol_blob_out) HIPCHK(hipMemcpyAsync(sol_blob_out, dblob, rows * batch * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(tf_out, dtf, batch * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(status_out, dstatus, batch * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(iters_out, diters, batch * sizeof(int), hipMemcpyDeviceToHost, stream));
  }
  if (!ptr_is_device || !stream) HIPCHK(hipStreamSynchronize(stream));
  return ASCENT_OK;
}

int ascent_eval_nodes(const ascent_params *p, int64_t batch, const ascent_opts *o, const double *iterate,
                      double *defects, double *jac_blocks, double *hess_blocks, int device_id) {
  int rc = check_common(p, batch, o, device_id);
  if (rc) return rc;
  if (!iterate || !defects || !jac_blocks || !hess_blocks) { snprintf(g_err, sizeof g_err, "null pointer"); return ASCENT_E_ARG; }
  std::lock_guard<std::mutex> lock(g_mu[device_id]);
  HIPCHK(hipSetDevice(device_id));
  const int K = o->n_nodes - 1;
  const long rows = 21L * K + NSC;
  DevBuf<ascent_params> bp;
  DevBuf<double> bit, bd, bj, bh;
  HIPCHK(bp.alloc(batch)); HIPCHK(bit.alloc(rows * batch));
  HIPCHK(bd.alloc(7L * K * batch)); HIPCHK(bj.alloc(8L * K * batch)); HIPCHK(bh.alloc(10L * K * batch));
  HIPCHK(hipMemcpy(bp.d, p, batch * sizeof(ascent_params), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(bit.d, iterate, rows * batch * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_eval_nodes, dim3((unsigned)((batch + 255) / 256), K), dim3(256), 0, 0, bp.d, (long)batch, K,
                     bit.d, bd.d, bj.d, bh.d);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(defects, bd.d, 7L * K * batch * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(jac_blocks, bj.d, 8L * K * batch * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hess_blocks, bh.d, 10L * K * batch * sizeof(double), hipMemcpyDeviceToHost));
  return ASCENT_OK;
}

int ascent_kkt_step(const ascent_params *p, int64_t batch, const ascent_opts *o, const double *iterate,
                    const double *mu, const double *delta_w, double *step, int32_t *inertia_out, int device_id) {
  int rc = check_common(p, batch, o, device_id);
  if (rc) return rc;
  if (!iterate || !mu || !delta_w || !step || !inertia_out) { snprintf(g_err, sizeof g_err, "null pointer"); return ASCENT_E_ARG; }
  std::lock_guard<std::mutex> lock(g_mu[device_id]);
  HIPCHK(hipSetDevice(device_id));
  const int K = o->n_nodes - 1;
  const long rows = 21L * K + NSC;
  const Layout L = make_layout(K, (long)batch);
  rc = ensure_ws(device_id, (size_t)L.total * (size_t)batch * sizeof(double));
  if (rc) return rc;
  DevBuf<ascent_params> bp;
  DevBuf<double> bit, bmu, bdw, bst;
  DevBuf<int> bin;
  HIPCHK(bp.alloc(batch)); HIPCHK(bit.alloc(rows * batch)); HIPCHK(bmu.alloc(batch)); HIPCHK(bdw.alloc(batch));
  HIPCHK(bst.alloc(rows * batch)); HIPCHK(bin.alloc(batch));
  HIPCHK(hipMemcpy(bp.d, p, batch * sizeof(ascent_params), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(bit.d, iterate, rows * batch * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(bmu.d, mu, batch * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(bdw.d, delta_w, batch * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(bst.d, 0, rows * batch * sizeof(double)));
  hipLaunchKernelGGL(k_kkt_step, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, 0, bp.d, (long)batch, L,
                     g_ws[device_id].ws, bit.d, bmu.d, bdw.d, bst.d, bin.d);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(step, bst.d, rows * batch * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(inertia_out, bin.d, batch * sizeof(int), hipMemcpyDeviceToHost));
  return ASCENT_OK;
}

}  // extern "C"
