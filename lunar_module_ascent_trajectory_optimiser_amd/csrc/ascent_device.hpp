// Device-side building blocks of the batched ascent NLP solver (gfx950, FP64 VALU).
//
// Mapping: one lane = one NLP.  Every per-problem array is structure-of-arrays with the problem
// index fastest, so the 64 lanes of a wavefront touch 64 consecutive doubles (one 512-B
// transaction) for every row they read or write; there is no cross-lane traffic at all.
// The time axis (the 199 collocation steps of /root/reference/Launch_Optimiser.py:20-21) is swept
// serially per lane: backward (evaluate + Riccati factorise), forward (primal step), backward
// (adjoint / multipliers).  The free final time tf (Launch_Optimiser.py:39-40) couples every
// step, and the terminal r.v = 0 constraint (:173) is a second global unknown: both are carried
// as border columns (two extra right-hand sides) of the block-tridiagonal system and closed by a
// 2x2 Schur complement whose entries are accumulated during the backward sweep.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "ascent.h"

namespace ascent {

constexpr int IX = 0, IY = 1, IVX = 2, IVY = 3, IA = 4, IW = 5, IM = 6;
// scalar rows at the end of a blob
constexpr int S_TH = 0, S_ZLT = 1, S_ZUT = 2, S_S1 = 3, S_S2 = 4, S_ZS1 = 5, S_ZS2 = 6, S_NU3 = 7,
              S_NU1 = 8, S_NU2 = 9, NSC = 10;

#define ASC_DEV __device__ __forceinline__
#define ASC_UNROLL _Pragma("unroll")

// index into a packed upper-triangular symmetric 7x7 (28 entries)
__host__ __device__ constexpr int sid(int i, int j) {
  return i <= j ? (i * 7 - (i * (i - 1)) / 2 + (j - i)) : (j * 7 - (j * (j - 1)) / 2 + (i - j));
}

// 1/x: hardware seed (v_rcp_f64) + two Newton steps, ~1 ulp; a full IEEE division costs three times
// the instructions and the solver divides ~40 times per collocation step.
ASC_DEV double rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// sin and cos of a bounded argument (|x| < ~1e5; the solver only passes 3*angle in [0, pi]):
// Cody-Waite reduction by pi/2 and the classic degree-13/14 minimax kernels on [-pi/4, pi/4].
ASC_DEV void sincos_bounded(double x, double &s, double &c) {
  const double kf = rint(x * 0.63661977236758134308);
  const int q = (int)kf;
  double r = fma(-kf, 1.57079632679489655800e+00, x);
  r = fma(-kf, 6.12323399573676603587e-17, r);
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                    2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03),
                    -1.66666666666666324348e-01);
  const double sr = fma(z * r, ps, r);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                    -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03),
                    4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  const double s1 = (q & 1) ? cr : sr, c1 = (q & 1) ? sr : cr;
  s = (q & 2) ? -s1 : s1;
  c = ((q + 1) & 2) ? -c1 : c1;
}

struct Der {  // constants derived from ascent_params (Launch_Optimiser.py:65,72-75,107-109)
  double rho0, rhof, vp2, gam, thr, alpha, mrate, ms, M0, T, aub, tlb, tub;
  // ascent_opts.terminal = 2 (dense-block path): burnout anywhere on the (r_peri, r_apo) ellipse -- its angular momentum and
  // specific energy in scaled units; term = 0 otherwise (the three constraints of Launch_Optimiser.py:158-173)
  int term;
  double ht, Et;
};

ASC_DEV Der derive(const ascent_params &p) {
  Der d;
  const double S = p.r_peri, GM = p.G * p.M;
  const double ravg = 0.5 * (p.r_peri + p.r_apo);
  const double vper2 = GM / (p.R0 + ravg);
  d.rho0 = p.R0 / S;
  d.rhof = (p.R0 + S) / S;
  d.vp2 = vper2 / (S * S);
  d.gam = GM / (S * S * S);
  d.thr = p.Ft / S;
  d.alpha = p.ang_acc_max / 3.0;
  d.mrate = p.mdot / p.fuel_mass;
  d.ms = p.mass_scalar;
  d.M0 = p.M0;
  d.T = p.T_scale;
  d.aub = p.angle_ub;
  d.tlb = p.tf_lb;
  d.tub = p.tf_ub;
  d.term = 0; d.ht = 0.0; d.Et = 0.0;
  return d;
}

// Scaled accelerations xdoubledot / ydoubledot (Launch_Optimiser.py:133-136 / 127-130) in polar
// form: a = th(m) * d(phi + 3*angle) - gam * e(phi) / rho^2, e = radial unit vector.
// LEVEL 0: values; 1: + gradients g[0..3] = d ax/d(x,y,angle,mass), g[4..7] = d ay/d(..);
// 2: + H[10] = upper triangle of the Hessian of px*ax + py*ay.
template <int LEVEL>
ASC_DEV void accel(const Der &d, double x, double y, double a, double m, double px, double py,
                   double &ax, double &ay, double *g, double *H) {
  const double xi = x, et = y + d.rho0;
  const double r2 = xi * xi + et * et;
  const double ir = rsqrt(r2);
  const double ex = xi * ir, ey = et * ir;
  double s, c;
  sincos_bounded(3.0 * a, s, c);
  const double dx = ex * c - ey * s, dy = ey * c + ex * s;
  const double imp = rcp(d.M0 - d.ms * m);
  const double th = d.thr * imp;
  const double g3 = d.gam * ir * ir * ir;
  ax = th * dx - g3 * xi;
  ay = th * dy - g3 * et;
  if constexpr (LEVEL >= 1) {
    const double th1 = th * d.ms * imp;
    const double fx = -ey * ir, fy = ex * ir;
    const double qx = -dy, qy = dx;
    const double g3e = 3.0 * g3 * ex * ey;
    g[0] = th * qx * fx - g3 * (1.0 - 3.0 * ex * ex);
    g[1] = th * qx * fy + g3e;
    g[2] = 3.0 * th * qx;
    g[3] = th1 * dx;
    g[4] = th * qy * fx + g3e;
    g[5] = th * qy * fy - g3 * (1.0 - 3.0 * ey * ey);
    g[6] = 3.0 * th * qy;
    g[7] = th1 * dy;
    if constexpr (LEVEL >= 2) {
      const double th2 = 2.0 * th1 * d.ms * imp;
      const double pd = px * dx + py * dy, pp = px * qx + py * qy, pe = px * ex + py * ey;
      const double ir2 = ir * ir;
      const double fxx = 2.0 * ex * ey * ir2, fxy = (ey * ey - ex * ex) * ir2;
      const double g4 = 3.0 * g3 * ir;
      const double tpd = th * pd, tpp = th * pp;
      H[0] = -tpd * fx * fx + tpp * fxx + g4 * (2.0 * px * ex + pe - 5.0 * pe * ex * ex);
      H[1] = -tpd * fx * fy + tpp * fxy + g4 * (px * ey + py * ex - 5.0 * pe * ex * ey);
      H[2] = -3.0 * tpd * fx;
      H[3] = th1 * pp * fx;
      H[4] = -tpd * fy * fy - tpp * fxx + g4 * (2.0 * py * ey + pe - 5.0 * pe * ey * ey);
      H[5] = -3.0 * tpd * fy;
      H[6] = th1 * pp * fy;
      H[7] = -9.0 * tpd;
      H[8] = 3.0 * th1 * pp;
      H[9] = th2 * pd;
    }
  }
}

// Inertia correction (Waechter & Biegler 2006, Algorithm IC, with retuned constants): the primal regularisation
// delta_w tried after a factorisation with the wrong inertia.  First correction of a solve 1e-2, later ones a
// third of the last successful value; x10 while the inertia stays wrong.  (IPOPT's 1e-4 / x100-then-x8 needs
// 0.54 refactorisations per NLP on the config-3 sweep and 3.4 on config 4; these need 0.28 and 1.7, for the
// same iteration counts -- every refactorisation is one more pass of the serial sweeps for that NLP.)
ASC_DEV double next_delta_w(double dw, double dw_last) {
  return dw == 0.0 ? (dw_last == 0.0 ? 1e-2 : fmax(1e-4, dw_last / 3.0)) : dw * 10.0;
}

// 2x2 inverse E of (I - dt^2 * d(ax,ay)/d(x,y)): the implicit (backward-Euler) position/velocity block
ASC_DEV void implicit_block(const double *G, double dt, double *E) {
  const double d2 = dt * dt;
  const double m11 = 1.0 - d2 * G[0], m12 = -d2 * G[1], m21 = -d2 * G[4], m22 = 1.0 - d2 * G[5];
  const double idet = rcp(m11 * m22 - m12 * m21);
  E[0] = m22 * idet;
  E[1] = -m12 * idet;
  E[2] = -m21 * idet;
  E[3] = m11 * idet;
}

// FORM 0: the current script (angle, angledot are states; u = angledoubledot).  FORM 1: the v1 script (PDF
// p26-28, the angle itself is the MV) embedded in the same 7-slot state: the angle row is algebraic,
// angle_k - (angle_ub/2)(u_k+1) = 0, so d f/d z loses its (angle, angledot) entry and angledot stays 0.

// v = A^-1 r,  A = I - dt * df/dz  (the step Jacobian of the backward-Euler defect w.r.t. z_k)
template <int FORM = 0>
ASC_DEV void solveA(const double *G, const double *E, double dt, const double *r, double *v) {
  const double vw = r[IW], vm = r[IM], va = FORM == 1 ? r[IA] : r[IA] + dt * vw;
  const double t1 = r[IVX] + dt * (G[0] * r[IX] + G[1] * r[IY] + G[2] * va + G[3] * vm);
  const double t2 = r[IVY] + dt * (G[4] * r[IX] + G[5] * r[IY] + G[6] * va + G[7] * vm);
  const double vvx = E[0] * t1 + E[1] * t2, vvy = E[2] * t1 + E[3] * t2;
  v[IX] = r[IX] + dt * vvx;
  v[IY] = r[IY] + dt * vvy;
  v[IVX] = vvx;
  v[IVY] = vvy;
  v[IA] = va;
  v[IW] = vw;
  v[IM] = vm;
}

// v = A^-T r
template <int FORM = 0>
ASC_DEV void solveAT(const double *G, const double *E, double dt, const double *r, double *v) {
  const double t1 = r[IVX] + dt * r[IX], t2 = r[IVY] + dt * r[IY];
  const double vvx = E[0] * t1 + E[2] * t2, vvy = E[1] * t1 + E[3] * t2;
  const double va = r[IA] + dt * (G[2] * vvx + G[6] * vvy);
  v[IX] = r[IX] + dt * (G[0] * vvx + G[4] * vvy);
  v[IY] = r[IY] + dt * (G[1] * vvx + G[5] * vvy);
  v[IVX] = vvx;
  v[IVY] = vvy;
  v[IA] = va;
  v[IM] = r[IM] + dt * (G[3] * vvx + G[7] * vvy);
  v[IW] = FORM == 1 ? r[IW] : r[IW] + dt * va;
}

// N <- T N T',  T = I + c e_I e_J'   (packed symmetric N)
template <int I, int J>
ASC_DEV void cong_add(double *N, double c) {
  const double nij = N[sid(I, J)], njj = N[sid(J, J)];
  ASC_UNROLL
  for (int l = 0; l < 7; l++)
    if (l != I) N[sid(I, l)] += c * N[sid(J, l)];
  N[sid(I, I)] += c * (2.0 * nij + c * njj);
}

// N <- A^-T N A^-1 as four in-place congruences (A^-T = T4 T3 T2 T1, see solveAT):
//   T1: rows xdot,ydot += dt * rows x,y;  T2: 2x2 block E' on (xdot,ydot);
//   T3: rows x,y,angle,mass += dt * G' (rows xdot,ydot);  T4: row angledot += dt * row angle.
template <int FORM = 0>
ASC_DEV void congruence(double *N, const double *G, const double *E, double dt) {
  cong_add<IVX, IX>(N, dt);
  cong_add<IVY, IY>(N, dt);
  {  // T2
    ASC_UNROLL
    for (int l = 0; l < 7; l++) {
      if (l == IVX || l == IVY) continue;
      const double a = N[sid(IVX, l)], b = N[sid(IVY, l)];
      N[sid(IVX, l)] = E[0] * a + E[2] * b;
      N[sid(IVY, l)] = E[1] * a + E[3] * b;
    }
    const double b11 = N[sid(IVX, IVX)], b12 = N[sid(IVX, IVY)], b22 = N[sid(IVY, IVY)];
    const double t11 = E[0] * b11 + E[2] * b12, t12 = E[0] * b12 + E[2] * b22;
    const double t21 = E[1] * b11 + E[3] * b12, t22 = E[1] * b12 + E[3] * b22;
    N[sid(IVX, IVX)] = t11 * E[0] + t12 * E[2];
    N[sid(IVX, IVY)] = t11 * E[1] + t12 * E[3];
    N[sid(IVY, IVY)] = t21 * E[1] + t22 * E[3];
  }
  cong_add<IX, IVX>(N, dt * G[0]);
  cong_add<IX, IVY>(N, dt * G[4]);
  cong_add<IY, IVX>(N, dt * G[1]);
  cong_add<IY, IVY>(N, dt * G[5]);
  cong_add<IA, IVX>(N, dt * G[2]);
  cong_add<IA, IVY>(N, dt * G[6]);
  cong_add<IM, IVX>(N, dt * G[3]);
  cong_add<IM, IVY>(N, dt * G[7]);
  if (FORM == 0) cong_add<IW, IA>(N, dt);
}

// y = N v for packed symmetric N
ASC_DEV void symv(const double *N, const double *v, double *y) {
  ASC_UNROLL
  for (int i = 0; i < 7; i++) {
    double a = 0.0;
    ASC_UNROLL
    for (int l = 0; l < 7; l++) a += N[sid(i, l)] * v[l];
    y[i] = a;
  }
}

// F_z' lambda : (d f/d z)' applied to the defect multipliers of a node
template <int FORM = 0>
ASC_DEV void fzt_lambda(const double *G, const double *l, double *fl) {
  fl[IX] = G[0] * l[IVX] + G[4] * l[IVY];
  fl[IY] = G[1] * l[IVX] + G[5] * l[IVY];
  fl[IVX] = l[IX];
  fl[IVY] = l[IY];
  fl[IA] = G[2] * l[IVX] + G[6] * l[IVY];
  fl[IW] = FORM == 1 ? 0.0 : l[IA];
  fl[IM] = G[3] * l[IVX] + G[7] * l[IVY];
}

// (d f/d z) v
ASC_DEV void fz_mul(const double *G, const double *v, double *o) {
  o[IX] = v[IVX];
  o[IY] = v[IVY];
  o[IVX] = G[0] * v[IX] + G[1] * v[IY] + G[2] * v[IA] + G[3] * v[IM];
  o[IVY] = G[4] * v[IX] + G[5] * v[IY] + G[6] * v[IA] + G[7] * v[IM];
  o[IA] = v[IW];
  o[IW] = 0.0;
  o[IM] = 0.0;
}

// Trapezoid scheme only: N <- Abar' N Abar with Abar = I + c * df/dz (packed symmetric N).  Abar has the
// x<->xdot, y<->ydot two-cycles, so it is not a product of the elementary congruences used for A^-1; the
// product is formed as T = N*Abar (sparse column combinations), then the upper triangle of Abar'*T.
ASC_DEV void congruence_abar(double *N, const double *G, double c) {
  double T[7][7];
  ASC_UNROLL
  for (int i = 0; i < 7; i++) {
    const double nx = N[sid(i, IX)], ny = N[sid(i, IY)], nvx = N[sid(i, IVX)], nvy = N[sid(i, IVY)];
    T[i][IX] = nx + c * (G[0] * nvx + G[4] * nvy);
    T[i][IY] = ny + c * (G[1] * nvx + G[5] * nvy);
    T[i][IVX] = nvx + c * nx;
    T[i][IVY] = nvy + c * ny;
    T[i][IA] = N[sid(i, IA)] + c * (G[2] * nvx + G[6] * nvy);
    T[i][IW] = N[sid(i, IW)] + c * N[sid(i, IA)];
    T[i][IM] = N[sid(i, IM)] + c * (G[3] * nvx + G[7] * nvy);
  }
  ASC_UNROLL
  for (int j = 0; j < 7; j++) {
    if (j >= IX) N[sid(IX, j)] = T[IX][j] + c * (G[0] * T[IVX][j] + G[4] * T[IVY][j]);
    if (j >= IY) N[sid(IY, j)] = T[IY][j] + c * (G[1] * T[IVX][j] + G[5] * T[IVY][j]);
    if (j >= IVX) N[sid(IVX, j)] = T[IVX][j] + c * T[IX][j];
    if (j >= IVY) N[sid(IVY, j)] = T[IVY][j] + c * T[IY][j];
    if (j >= IA) N[sid(IA, j)] = T[IA][j] + c * (G[2] * T[IVX][j] + G[6] * T[IVY][j]);
    if (j >= IW) N[sid(IW, j)] = T[IW][j] + c * T[IA][j];
    if (j >= IM) N[sid(IM, j)] = T[IM][j] + c * (G[3] * T[IVX][j] + G[7] * T[IVY][j]);
  }
}

// right-hand side f(z,u) of the scaled ODEs without the tf*T factor (Launch_Optimiser.py:114-123)
template <int FORM = 0>
ASC_DEV void rhs_f(const Der &d, const double *z, double u, double ax, double ay, double *F) {
  F[IX] = z[IVX];
  F[IY] = z[IVY];
  F[IVX] = ax;
  F[IVY] = ay;
  F[IA] = FORM == 1 ? 0.0 : z[IW];
  F[IW] = FORM == 1 ? 0.0 : d.alpha * u;
  F[IM] = d.mrate;
}

// Terminal constraints at the last node (Launch_Optimiser.py:158-173, divided through by Scalar^2
// where the reference multiplies positions and velocities by Scalar):
//   e3 = (y+rho0)*ydot + x*xdot = 0 ;  g1 = |(x, y+rho0)| - rhof - s1 = 0 ;  g2 = xdot^2+ydot^2 - vp2 - s2 = 0
struct Terminal {
  double e3, g1, g2;        // constraint values (g_i before subtracting the slack)
  double e3g[4];            // d e3 / d(x,y,xdot,ydot)
  double g1g[2];            // d g1 / d(x,y)
  double g2g[2];            // d g2 / d(xdot,ydot)
  double hxx, hxy, hyy;     // second derivatives of g1
  // Der::term == 2 only (terminal_eval_any): both conditions depend on position and velocity
  double g1v[2];            // d g1 / d(xdot,ydot)
  double g2p[2];            // d g2 / d(x,y)
  double qxx, qxy, qyy;     // position block of the Hessian of g2
};

ASC_DEV Terminal terminal_eval(const Der &d, const double *z) {
  Terminal t;
  const double et = z[IY] + d.rho0;
  const double r2 = z[IX] * z[IX] + et * et;
  const double ir = rsqrt(r2), rho = r2 * ir;
  const double ex = z[IX] * ir, ey = et * ir;
  t.e3 = et * z[IVY] + z[IX] * z[IVX];
  t.g1 = rho - d.rhof;
  t.g2 = z[IVX] * z[IVX] + z[IVY] * z[IVY] - d.vp2;
  t.e3g[0] = z[IVX];
  t.e3g[1] = z[IVY];
  t.e3g[2] = z[IX];
  t.e3g[3] = et;
  t.g1g[0] = ex;
  t.g1g[1] = ey;
  t.g2g[0] = 2.0 * z[IVX];
  t.g2g[1] = 2.0 * z[IVY];
  t.hxx = ey * ey * ir;
  t.hxy = -ex * ey * ir;
  t.hyy = ex * ex * ir;
  return t;
}

// Der::term == 2: burnout anywhere on the (r_peri, r_apo) ellipse (README.md:7) instead of Launch_Optimiser.py:158-173's
// three conditions at its periapsis -- two conditions, each an inequality with a slack like the reference's own two, both
// active at the optimum:
//   g1 = h - h_t - s1 = 0,  h = x ydot - (y+rho0) xdot   (angular momentum not below the ellipse's)
//   g2 = E_t - E - s2 = 0,  E = (xdot^2+ydot^2)/2 - gam/rho   (specific energy not above the ellipse's)
// i.e. an orbit nested in the target annulus (periapsis not lower, apoapsis not higher).  There is no r.v = 0: e3 = 0 with
// zero gradient, its multiplier nu3 stays where it is (the border's nu3 row is closed with a unit pivot).
ASC_DEV Terminal terminal_eval_any(const Der &d, const double *z) {
  Terminal t;
  const double et = z[IY] + d.rho0;
  const double r2 = z[IX] * z[IX] + et * et;
  const double ir = rsqrt(r2);
  const double ex = z[IX] * ir, ey = et * ir;
  const double g3 = d.gam * ir * ir * ir;
  t.e3 = 0.0;
  t.e3g[0] = t.e3g[1] = t.e3g[2] = t.e3g[3] = 0.0;
  t.g1 = z[IX] * z[IVY] - et * z[IVX] - d.ht;
  t.g1g[0] = z[IVY]; t.g1g[1] = -z[IVX]; t.g1v[0] = -et; t.g1v[1] = z[IX];
  t.hxx = t.hxy = t.hyy = 0.0;
  t.g2 = d.Et - (0.5 * (z[IVX] * z[IVX] + z[IVY] * z[IVY]) - d.gam * ir);
  t.g2p[0] = -g3 * z[IX]; t.g2p[1] = -g3 * et; t.g2g[0] = -z[IVX]; t.g2g[1] = -z[IVY];
  t.qxx = -g3 * (1.0 - 3.0 * ex * ex); t.qxy = 3.0 * g3 * ex * ey; t.qyy = -g3 * (1.0 - 3.0 * ey * ey);
  return t;
}
// gradient of a1 g1 + a2 g2 with respect to (x, y, xdot, ydot)
ASC_DEV void terminal_grad_any(const Terminal &t, double a1, double a2, double *g) {
  g[0] = a1 * t.g1g[0] + a2 * t.g2p[0];
  g[1] = a1 * t.g1g[1] + a2 * t.g2p[1];
  g[2] = a1 * t.g1v[0] + a2 * t.g2g[0];
  g[3] = a1 * t.g1v[1] + a2 * t.g2g[1];
}
// ... its terminal Lagrangian Hessian and slack-eliminated barrier terms (the counterpart of terminal_hessian below)
ASC_DEV void terminal_hessian_any(double *Q, const Terminal &t, double nu1, double nu2, double sig1, double sig2) {
  const double G1[4] = {t.g1g[0], t.g1g[1], t.g1v[0], t.g1v[1]}, G2[4] = {t.g2p[0], t.g2p[1], t.g2g[0], t.g2g[1]};
  ASC_UNROLL
  for (int i = 0; i < 4; i++) {
    ASC_UNROLL
    for (int j = i; j < 4; j++) Q[sid(i, j)] += sig1 * G1[i] * G1[j] + sig2 * G2[i] * G2[j];
  }
  Q[sid(IX, IX)] += nu2 * t.qxx;
  Q[sid(IX, IY)] += nu2 * t.qxy;
  Q[sid(IY, IY)] += nu2 * t.qyy;
  Q[sid(IVX, IVX)] -= nu2;
  Q[sid(IVY, IVY)] -= nu2;
  Q[sid(IX, IVY)] += nu1;
  Q[sid(IY, IVX)] -= nu1;
}

// terminal constants of ascent_opts.terminal: 0 the reference's (Launch_Optimiser.py:72-78: circular speed of the mean radius
// at r_peri), 1 the ellipse proper (vis-viva speed at the periapsis of the (r_peri, r_apo) ellipse), 2 anywhere on that ellipse
ASC_DEV Der derive_t(const ascent_params &p, int terminal) {
  Der d = derive(p);
  if (terminal == 1 || terminal == 2) {
    const double S = p.r_peri, GM = p.G * p.M, rp = p.R0 + p.r_peri, ra = p.R0 + p.r_apo;
    d.vp2 = GM * (2.0 / rp - 2.0 / (ra + rp)) / (S * S);      // (terminal 2: the cold start still aims at the periapsis)
    if (terminal == 2) {       // burnout anywhere on that ellipse: its angular momentum and specific energy, scaled units
      const double rps = rp / S, ras = ra / S;
      d.term = 2;
      d.ht = sqrt(2.0 * d.gam * rps * ras / (rps + ras));
      d.Et = -d.gam / (rps + ras);
    }
  }
  return d;
}
ASC_DEV Terminal terminal_of(const Der &d, const double *z) { return d.term == 2 ? terminal_eval_any(d, z) : terminal_eval(d, z); }

// adds the terminal Lagrangian Hessian and the slack-eliminated barrier terms to the last node's Q
ASC_DEV void terminal_hessian(double *Q, const Terminal &t, double nu3, double nu1, double nu2,
                              double sig1, double sig2) {
  Q[sid(IX, IX)] += nu1 * t.hxx + sig1 * t.g1g[0] * t.g1g[0];
  Q[sid(IX, IY)] += nu1 * t.hxy + sig1 * t.g1g[0] * t.g1g[1];
  Q[sid(IY, IY)] += nu1 * t.hyy + sig1 * t.g1g[1] * t.g1g[1];
  Q[sid(IVX, IVX)] += 2.0 * nu2 + sig2 * t.g2g[0] * t.g2g[0];
  Q[sid(IVX, IVY)] += sig2 * t.g2g[0] * t.g2g[1];
  Q[sid(IVY, IVY)] += 2.0 * nu2 + sig2 * t.g2g[1] * t.g2g[1];
  Q[sid(IX, IVX)] += nu3;
  Q[sid(IY, IVY)] += nu3;
}

}  // namespace ascent
