/* TEST INFRASTRUCTURE ONLY -- plain-C CPU restatement of the lunar-ascent NLP solve.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (lunar_module_ascent_trajectory_optimiser_amd/) never does.
 *
 * Restates /root/reference/Launch_Optimiser.py ("LO"):
 *   grid + 2-point collocation (backward Euler)  LO:20-21,25     parameters/scales LO:38-75,107-109
 *   variables/bounds LO:83-100                   ODEs LO:114-123  accelerations LO:127-136
 *   initial conditions LO:145-151                terminal constraints LO:158-173 (last node only)
 *   objective LO:176                             outputs LO:187-202
 * The NLP solve (LO:177) happens in GEKKO->APMonitor->IPOPT, which is not in /root/reference and
 * not installed (gekko, version unpinned); its published primal-dual interior-point algorithm
 * (Waechter & Biegler, Math. Prog. 106, 2006) is restated here with an l1-merit line search and a
 * stage-wise (Riccati) factorisation of the bordered block-tridiagonal KKT system.
 * Pinned by tests/golden/golden.json (Numerical_results.png, PDF p30) through
 * tests/test_oracle.py, and cross-checked there against oracle/ascent_numpy.py, which solves
 * the same Newton systems with a generic sparse LU.
 *
 * State order per node: x y xdot ydot angle angledot mass  (all in the reference's scaled units).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define NS 7
enum { IX, IY, IVX, IVY, IA, IW, IM };

typedef struct {            /* 16 doubles, same order as include/ascent.h's ascent_params */
  double G, M, R0, Ft, M0, mdot, fuel_mass, mass_scalar, ang_acc_max, r_peri, r_apo, T_scale,
      angle_ub, tf_lb, tf_ub, dcost;
} oparams;

typedef struct { double S, rho0, rhof, vp2, gam, thr, alpha, mrate, ms, M0, T, aub, tlb, tub, dcost; } oder;

static void derive(const oparams *p, oder *d) {
  double S = p->r_peri, GM = p->G * p->M;            /* LO:73,107 */
  double ravg = 0.5 * (p->r_peri + p->r_apo);        /* LO:72 */
  double vper = sqrt(GM / (p->R0 + ravg));           /* LO:75 */
  d->S = S; d->rho0 = p->R0 / S; d->rhof = (p->R0 + S) / S;
  d->vp2 = (vper / S) * (vper / S);
  d->gam = GM / (S * S * S); d->thr = p->Ft / S;
  d->alpha = p->ang_acc_max / 3.0;                   /* LO:109 */
  d->mrate = p->mdot / p->fuel_mass;                 /* LO:65 */
  d->ms = p->mass_scalar; d->M0 = p->M0; d->T = p->T_scale;
  d->aub = p->angle_ub; d->tlb = p->tf_lb; d->tub = p->tf_ub; d->dcost = p->dcost;
}

/* accelerations LO:127-136; gradients w.r.t. (x,y,angle,mass); Hessian of px*ax+py*ay
 * (order xx xy xa xm yy ya ym aa am mm) when H != NULL */
static void accel(const oder *d, double x, double y, double a, double m, double px, double py,
                  double *ax, double *ay, double gax[4], double gay[4], double H[10]) {
  double xi = x, et = y + d->rho0;
  double rho = sqrt(xi * xi + et * et), ir = 1.0 / rho;
  double ex = xi * ir, ey = et * ir;
  double c = cos(3.0 * a), s = sin(3.0 * a);
  double dx = ex * c - ey * s, dy = ey * c + ex * s;
  double mp = d->M0 - d->ms * m;
  double th = d->thr / mp, th1 = th * d->ms / mp, th2 = 2.0 * th1 * d->ms / mp;
  double g3 = d->gam * ir * ir * ir;
  *ax = th * dx - g3 * xi;
  *ay = th * dy - g3 * et;
  if (!gax) return;
  double fx = -ey * ir, fy = ex * ir, qx = -dy, qy = dx;
  gax[0] = th * qx * fx - g3 * (1 - 3 * ex * ex);
  gax[1] = th * qx * fy + g3 * 3 * ex * ey;
  gax[2] = 3 * th * qx;
  gax[3] = th1 * dx;
  gay[0] = th * qy * fx + g3 * 3 * ex * ey;
  gay[1] = th * qy * fy - g3 * (1 - 3 * ey * ey);
  gay[2] = 3 * th * qy;
  gay[3] = th1 * dy;
  if (!H) return;
  double pd = px * dx + py * dy, pp = px * qx + py * qy, pe = px * ex + py * ey;
  double fxx = 2 * ex * ey * ir * ir, fxy = (ey * ey - ex * ex) * ir * ir, fyy = -fxx;
  double g4 = 3.0 * g3 * ir;
  H[0] = th * (-pd * fx * fx + pp * fxx) + g4 * (2 * px * ex + pe - 5 * pe * ex * ex);
  H[1] = th * (-pd * fx * fy + pp * fxy) + g4 * (px * ey + py * ex - 5 * pe * ex * ey);
  H[2] = -3 * th * pd * fx;
  H[3] = th1 * pp * fx;
  H[4] = th * (-pd * fy * fy + pp * fyy) + g4 * (2 * py * ey + pe - 5 * pe * ey * ey);
  H[5] = -3 * th * pd * fy;
  H[6] = th1 * pp * fy;
  H[7] = -9 * th * pd;
  H[8] = 3 * th1 * pp;
  H[9] = th2 * pd;
}

/* ---- flat iterate blob: z[7K] u[K] lam[7K] zb[6K] sc[10] --------------------------------- */
/* zb per node: zL_angle zU_angle zL_mass zU_mass zL_u zU_u ; sc: th zLth zUth s1 s2 zs1 zs2 nu3 nu1 nu2 */
enum { S_TH, S_ZLT, S_ZUT, S_S1, S_S2, S_ZS1, S_ZS2, S_NU3, S_NU1, S_NU2, NSC };
#define BLOB(K) (21 * (K) + NSC)
/* internal iterates carry five more rows per step behind the external blob -- the move penalty's multiplier and slack pair
 * (lambda_u, p, n, z_p, z_n; see g_mp below); the external layout (include/ascent.h) does not change */
#define BLOBX(K) (26 * (K) + NSC)
typedef struct { int K; double *z, *u, *lam, *zb, *sc, *lu, *pp, *pn, *zp, *zn; } iter_t;
static void view(double *b, int K, iter_t *it) {
  it->K = K; it->z = b; it->u = b + 7 * K; it->lam = b + 8 * K; it->zb = b + 15 * K; it->sc = b + 21 * K;
  it->lu = b + 21 * K + NSC; it->pp = it->lu + K; it->pn = it->pp + K; it->zp = it->pn + K; it->zn = it->zp + K;
}

typedef struct { double G[8], H[10], F[7], E[4]; } stage_t;

/* Collocation scheme (global to one solve; set by the exported entry points):
 *   0 = backward Euler, the reference's NODES=2 (LO:25):   z_k - z_{k-1} - dt f(z_k,u_k)
 *   1 = trapezoid with the control held over the step:      z_k - z_{k-1} - dt/2 [f(z_k,u_k) + f(z_{k-1},u_k)]
 * Scheme 1 is not a reference scheme; it is pinned by SURVEY.md Appendix C's independent probe (435.227 s). */
static __thread int g_scheme = 0;
/* Formulation (global to one solve):
 *   0 = current script: angle and angledot are states, u = angledoubledot is the MV (LO:94-100,120-121)
 *   1 = v1 script (PDF p26-28): the angle itself is the MV.  Embedded in the same 7-slot state vector: the
 *       angle row becomes algebraic,  angle_k - (angle_ub/2)(u_k + 1) = 0  with u in [-1,1] (so angle in
 *       [0, angle_ub], the MV's bounds), it has no coupling to angle_{k-1}, and angledot stays 0. */
static __thread int g_form = 0;
/* Move penalty (LO:99, angledoubledot.DCOST = 1e-5; MV_TYPE = 0, LO:29): 1 = the objective is tf + dcw * sum_k |u_k - u_{k-1}|,
 * u_{-1} = the MV's initial value (0; formulation 1: the angle is the MV, angle_{-1} = 0, i.e. u_{-1} = -1, and the weight on
 * u is dcost * angle_ub/2), as the l1 term APMonitor documents for DCOST: a slack pair per step,
 *   u_k - u_{k-1} - p_k + n_k = 0 (multiplier lambda_u,k),  p, n >= 0,  cost dcw (p_k + n_k).
 * The control then couples to the control of the step before: u_k becomes the eighth state of a stage, the stage's scalar
 * control is delta_k = p_k - n_k, and its two bounded slacks reduce to one pivot (kkt_solve_mp). */
static __thread int g_mp = 0;
static double mp_weight(const oder *d) { return g_form == 1 ? d->dcost * 0.5 * d->aub : d->dcost; }
static double mp_u_init(void) { return g_form == 1 ? -1.0 : 0.0; }

/* rhs of the scaled ODEs without tf*T (LO:114-123) */
static void rhs_f(const oder *d, const double *z, double u, double ax, double ay, double F[7]) {
  F[IX] = z[IVX]; F[IY] = z[IVY]; F[IVX] = ax; F[IVY] = ay;
  F[IA] = z[IW]; F[IW] = d->alpha * u; F[IM] = d->mrate;
  if (g_form == 1) { F[IA] = 0.0; F[IW] = 0.0; }
}

/* step function Fc_k = f(z_k,u_k) (scheme 0) or the trapezoid mean with f(z_{k-1},u_k) (scheme 1) */
static void step_f(const oder *d, const double *z, const double *zp, double u, double ax, double ay, double Fc[7]) {
  rhs_f(d, z, u, ax, ay, Fc);
  if (g_scheme == 1) {
    double axp, ayp, Fb[7];
    accel(d, zp[IX], zp[IY], zp[IA], zp[IM], 0, 0, &axp, &ayp, 0, 0, 0);
    rhs_f(d, zp, u, axp, ayp, Fb);
    for (int i = 0; i < 7; i++) Fc[i] = 0.5 * (Fc[i] + Fb[i]);
  }
}

/* equality constraints: defects (7K), e3, g1-s1, g2-s2 */
static void constraints(const oder *d, int K, double h, const iter_t *it, double *c) {
  double dt = h * d->T * it->sc[S_TH];
  double zero[7] = {0};
  for (int k = 0; k < K; k++) {
    const double *z = it->z + 7 * k, *zp = k ? z - 7 : zero;
    double ax, ay, F[7];
    accel(d, z[IX], z[IY], z[IA], z[IM], 0, 0, &ax, &ay, 0, 0, 0);
    step_f(d, z, zp, it->u[k], ax, ay, F);
    for (int i = 0; i < 7; i++) c[7 * k + i] = z[i] - zp[i] - dt * F[i];
    if (g_form == 1) c[7 * k + IA] = z[IA] - 0.5 * d->aub * (it->u[k] + 1.0);
  }
  const double *z = it->z + 7 * (K - 1);
  double et = z[IY] + d->rho0;
  c[7 * K] = et * z[IVY] + z[IX] * z[IVX];                                /* LO:173 / S^2 */
  c[7 * K + 1] = sqrt(z[IX] * z[IX] + et * et) - d->rhof - it->sc[S_S1];  /* LO:161 */
  c[7 * K + 2] = z[IVX] * z[IVX] + z[IVY] * z[IVY] - d->vp2 - it->sc[S_S2]; /* LO:169 */
}

static void solveA(const stage_t *s, double dt, const double *r, double *v) {
  const double *G = s->G, *E = s->E;
  double vw = r[IW], vm = r[IM], va = r[IA] + (g_form == 1 ? 0.0 : dt * vw);
  double t1 = r[IVX] + dt * (G[0] * r[IX] + G[1] * r[IY] + G[2] * va + G[3] * vm);
  double t2 = r[IVY] + dt * (G[4] * r[IX] + G[5] * r[IY] + G[6] * va + G[7] * vm);
  double vvx = E[0] * t1 + E[1] * t2, vvy = E[2] * t1 + E[3] * t2;
  v[IX] = r[IX] + dt * vvx; v[IY] = r[IY] + dt * vvy; v[IVX] = vvx; v[IVY] = vvy;
  v[IA] = va; v[IW] = vw; v[IM] = vm;
}
static void solveAT(const stage_t *s, double dt, const double *r, double *v) {
  const double *G = s->G, *E = s->E;
  double t1 = r[IVX] + dt * r[IX], t2 = r[IVY] + dt * r[IY];
  double vvx = E[0] * t1 + E[2] * t2, vvy = E[1] * t1 + E[3] * t2;
  double va = r[IA] + dt * (G[2] * vvx + G[6] * vvy);
  v[IX] = r[IX] + dt * (G[0] * vvx + G[4] * vvy);
  v[IY] = r[IY] + dt * (G[1] * vvx + G[5] * vvy);
  v[IVX] = vvx; v[IVY] = vvy; v[IA] = va;
  v[IM] = r[IM] + dt * (G[3] * vvx + G[7] * vvy);
  v[IW] = r[IW] + (g_form == 1 ? 0.0 : dt * va);
}

/* (d f/d z)' v */
static void fzt(const double *G, const double *l, double *fl) {
  fl[IX] = G[0] * l[IVX] + G[4] * l[IVY]; fl[IY] = G[1] * l[IVX] + G[5] * l[IVY];
  fl[IVX] = l[IX]; fl[IVY] = l[IY];
  fl[IA] = G[2] * l[IVX] + G[6] * l[IVY]; fl[IW] = g_form == 1 ? 0.0 : l[IA];
  fl[IM] = G[3] * l[IVX] + G[7] * l[IVY];
}
/* (d f/d z) v */
static void fz(const double *G, const double *v, double *o) {
  o[IX] = v[IVX]; o[IY] = v[IVY];
  o[IVX] = G[0] * v[IX] + G[1] * v[IY] + G[2] * v[IA] + G[3] * v[IM];
  o[IVY] = G[4] * v[IX] + G[5] * v[IY] + G[6] * v[IA] + G[7] * v[IM];
  o[IA] = v[IW]; o[IW] = 0.0; o[IM] = 0.0;
}
/* trapezoid only: Abar = I + c F_z(previous node) maps dz_{k-1} into step k */
static void abar_mul(const double *G, double c, const double *v, double *o) {
  double t[7]; fz(G, v, t);
  for (int i = 0; i < 7; i++) o[i] = v[i] + c * t[i];
}
static void abart_mul(const double *G, double c, const double *v, double *o) {
  double t[7]; fzt(G, v, t);
  for (int i = 0; i < 7; i++) o[i] = v[i] + c * t[i];
}

typedef struct {
  int K; double h;
  stage_t *st;
  double *Q;        /* K*49 : W + Sigma + delta_w (+ slack-eliminated terminal terms at K) */
  double *R;        /* K */
  double *gth, *gu; /* K*7, K : d2L/dtheta dz, d2L/dtheta du */
  double *rz, *ru;  /* K*7, K : barrier-form dual residual  grad phi_mu + J' lam */
  double *c;        /* 7K+3 */
  double *kap, *kap0, *Dp; /* K*7, K*3, K */
  double rth, sth, e3g[7], g1g[7], g2g[7], sig[2], rs[2];
  double *step;     /* blob */
  double *trial, *ctrial;
  /* move penalty: movement equations cu (K), the reduced slack pair's curvature Rd and gradient gdl (K), the 8-state
   * feedback gains kap8 (8K), the stage control's step ddel (K), the regularisation of the last assemble */
  double *cu, *Rd, *gdl, *kap8, *ddel, *cutrial, dw;
} work_t;

static work_t *work_new(int K) {
  work_t *w = calloc(1, sizeof *w);
  w->K = K; w->h = 1.0 / K;
  w->st = calloc(K, sizeof(stage_t));
  w->Q = calloc(49 * K, 8); w->R = calloc(K, 8); w->gth = calloc(7 * K, 8); w->gu = calloc(K, 8);
  w->rz = calloc(7 * K, 8); w->ru = calloc(K, 8); w->c = calloc(7 * K + 3, 8);
  w->kap = calloc(7 * K, 8); w->kap0 = calloc(3 * K, 8); w->Dp = calloc(K, 8);
  w->step = calloc(BLOBX(K), 8); w->trial = calloc(BLOBX(K), 8); w->ctrial = calloc(7 * K + 3, 8);
  w->cu = calloc(K, 8); w->Rd = calloc(K, 8); w->gdl = calloc(K, 8); w->kap8 = calloc(8 * K, 8); w->ddel = calloc(K, 8);
  w->cutrial = calloc(K, 8);
  return w;
}
static void work_free(work_t *w) {
  free(w->st); free(w->Q); free(w->R); free(w->gth); free(w->gu); free(w->rz); free(w->ru);
  free(w->c); free(w->kap); free(w->kap0); free(w->Dp); free(w->step); free(w->trial);
  free(w->ctrial); free(w->cu); free(w->Rd); free(w->gdl); free(w->kap8); free(w->ddel); free(w->cutrial); free(w);
}

/* movement equations of the move penalty: cu_k = u_k - u_{k-1} - p_k + n_k */
static double u_prev_move(const iter_t *it, int k) { return it->u[k] - (k ? it->u[k - 1] : mp_u_init()) - it->pp[k] + it->pn[k]; }
static void move_constraints(int K, const iter_t *it, double *cu) {
  for (int k = 0; k < K; k++) cu[k] = it->u[k] - (k ? it->u[k - 1] : mp_u_init()) - it->pp[k] + it->pn[k];
}

/* evaluate everything the Newton system needs at the iterate (mu enters the barrier gradient) */
static void assemble(const oder *d, work_t *w, const iter_t *it, double mu, double dw) {
  int K = w->K; double hT = w->h * d->T, th = it->sc[S_TH], dt = hT * th;
  /* node k enters step k with weight cs/dt (1 for backward Euler, 1/2 for trapezoid) and, for the
   * trapezoid, step k+1 with the same weight: lt = lambda_k (+ lambda_{k+1}) */
  const double cs = g_scheme == 1 ? 0.5 * dt : dt, hTc = g_scheme == 1 ? 0.5 * hT : hT;
  constraints(d, K, w->h, it, w->c);
  double rth = 1.0, zero7[7] = {0};
  for (int k = 0; k < K; k++) {
    const double *z = it->z + 7 * k, *l = it->lam + 7 * k, *zb = it->zb + 6 * k;
    const double *ln = (k + 1 < K) ? l + 7 : 0;
    stage_t *s = w->st + k;
    double ax, ay, lt[7];
    for (int i = 0; i < 7; i++) lt[i] = l[i] + ((g_scheme == 1 && ln) ? ln[i] : 0.0);
    accel(d, z[IX], z[IY], z[IA], z[IM], -cs * lt[IVX], -cs * lt[IVY], &ax, &ay, s->G, s->G + 4, s->H);
    step_f(d, z, k ? z - 7 : zero7, it->u[k], ax, ay, s->F);
    const double *G = s->G;
    double m11 = 1 - cs * cs * G[0], m12 = -cs * cs * G[1], m21 = -cs * cs * G[4], m22 = 1 - cs * cs * G[5];
    double idet = 1.0 / (m11 * m22 - m12 * m21);
    s->E[0] = m22 * idet; s->E[1] = -m12 * idet; s->E[2] = -m21 * idet; s->E[3] = m11 * idet;
    /* F_z' lt */
    double fl[7];
    fzt(G, lt, fl);
    double *rz = w->rz + 7 * k, *g = w->gth + 7 * k;
    for (int i = 0; i < 7; i++) { rz[i] = l[i] - cs * fl[i] - ((ln && !(g_form == 1 && i == IA)) ? ln[i] : 0.0); g[i] = -hTc * fl[i]; }
    double a = z[IA], m = z[IM], u = it->u[k];
    rz[IA] += -mu / a + mu / (d->aub - a);
    rz[IM] += -mu / m + mu / (1.0 - m);
    w->ru[k] = (g_form == 1 ? -0.5 * d->aub * l[IA] : -dt * d->alpha * l[IW]) - mu / (u + 1.0) + mu / (1.0 - u);
    w->gu[k] = g_form == 1 ? 0.0 : -hT * d->alpha * l[IW];
    w->R[k] = zb[4] / (u + 1.0) + zb[5] / (1.0 - u) + dw;
    for (int i = 0; i < 7; i++) rth -= hT * s->F[i] * l[i];
    double *Q = w->Q + 49 * k;
    memset(Q, 0, 49 * 8);
    static const int q4[4] = {IX, IY, IA, IM};
    int idx = 0;
    for (int i = 0; i < 4; i++) for (int j = i; j < 4; j++) {
      Q[q4[i] * 7 + q4[j]] = s->H[idx]; Q[q4[j] * 7 + q4[i]] = s->H[idx]; idx++;
    }
    Q[IA * 7 + IA] += zb[0] / a + zb[1] / (d->aub - a);
    Q[IM * 7 + IM] += zb[2] / m + zb[3] / (1.0 - m);
    for (int i = 0; i < 7; i++) Q[i * 7 + i] += dw;
  }
  /* terminal node */
  const double *z = it->z + 7 * (K - 1);
  double *rz = w->rz + 7 * (K - 1), *Q = w->Q + 49 * (K - 1);
  double et = z[IY] + d->rho0, rho = sqrt(z[IX] * z[IX] + et * et), ex = z[IX] / rho, ey = et / rho;
  double nu3 = it->sc[S_NU3], nu1 = it->sc[S_NU1], nu2 = it->sc[S_NU2];
  memset(w->e3g, 0, 56); memset(w->g1g, 0, 56); memset(w->g2g, 0, 56);
  w->e3g[IX] = z[IVX]; w->e3g[IY] = z[IVY]; w->e3g[IVX] = z[IX]; w->e3g[IVY] = et;
  w->g1g[IX] = ex; w->g1g[IY] = ey;
  w->g2g[IVX] = 2 * z[IVX]; w->g2g[IVY] = 2 * z[IVY];
  double s1 = it->sc[S_S1], s2 = it->sc[S_S2];
  w->sig[0] = it->sc[S_ZS1] / s1 + dw; w->sig[1] = it->sc[S_ZS2] / s2 + dw;
  w->rs[0] = -mu / s1 - nu1; w->rs[1] = -mu / s2 - nu2;
  double cg1 = w->c[7 * K + 1], cg2 = w->c[7 * K + 2];
  for (int i = 0; i < 7; i++)
    rz[i] += nu3 * w->e3g[i] + nu1 * w->g1g[i] + nu2 * w->g2g[i]
           + w->g1g[i] * (w->sig[0] * cg1 + w->rs[0]) + w->g2g[i] * (w->sig[1] * cg2 + w->rs[1]);
  Q[IX * 7 + IX] += nu1 * ey * ey / rho; Q[IX * 7 + IY] -= nu1 * ex * ey / rho;
  Q[IY * 7 + IX] -= nu1 * ex * ey / rho; Q[IY * 7 + IY] += nu1 * ex * ex / rho;
  Q[IVX * 7 + IVX] += 2 * nu2; Q[IVY * 7 + IVY] += 2 * nu2;
  Q[IX * 7 + IVX] += nu3; Q[IVX * 7 + IX] += nu3; Q[IY * 7 + IVY] += nu3; Q[IVY * 7 + IY] += nu3;
  for (int i = 0; i < 7; i++) for (int j = 0; j < 7; j++)
    Q[i * 7 + j] += w->sig[0] * w->g1g[i] * w->g1g[j] + w->sig[1] * w->g2g[i] * w->g2g[j];
  w->rth = rth - mu / (th - d->tlb) + mu / (d->tub - th);
  w->sth = it->sc[S_ZLT] / (th - d->tlb) + it->sc[S_ZUT] / (d->tub - th) + dw;
  w->dw = dw;
  if (g_mp) {
    /* the control is a state now: its stationarity row gains the multipliers of its two movement equations; the stage's
     * control delta_k = p_k - n_k carries the reduced slack pair: with Sigma_p = z_p/p + dw, Sigma_n = z_n/n + dw and the
     * stationarity residuals r_p = dcw - mu/p - lambda_u, r_n = dcw - mu/n + lambda_u:
     *   curvature Rd = 1/(1/Sigma_p + 1/Sigma_n),  gradient gdl = Rd (r_p/Sigma_p - r_n/Sigma_n),  d lambda_u = Rd d delta + gdl */
    const double dcw = mp_weight(d);
    move_constraints(K, it, w->cu);
    for (int k = 0; k < K; k++) {
      w->ru[k] += it->lu[k] - (k + 1 < K ? it->lu[k + 1] : 0.0);
      const double ip = 1.0 / it->pp[k], in = 1.0 / it->pn[k];
      const double isp = 1.0 / (it->zp[k] * ip + dw), isn = 1.0 / (it->zn[k] * in + dw);
      w->Rd[k] = 1.0 / (isp + isn);
      w->gdl[k] = w->Rd[k] * ((dcw - mu * ip - it->lu[k]) * isp - (dcw - mu * in + it->lu[k]) * isn);
    }
  }
}

/* Newton step by the backward (Riccati) / forward / adjoint sweeps with the (dtheta, dnu3) border.
 * returns 0 ok, 1 wrong inertia (caller raises delta_w). Fills w->step (primal+lam+nu parts). */
static int kkt_solve(const oder *d, work_t *w, const iter_t *it, double mu) {
  int K = w->K; double hT = w->h * d->T, th = it->sc[S_TH], dt = hT * th, be = dt * d->alpha;
  const double cs = g_scheme == 1 ? 0.5 * dt : dt;   /* weight of f(z_k,u_k) in step k, times dt */
  const int IB = g_form == 1 ? IA : IW;              /* the defect row the control enters ... */
  be = g_form == 1 ? 0.5 * d->aub : be;              /* ... and its coefficient */
  double P[49] = {0}, p[3][7] = {{0}};
  double S10 = 0, S11 = 0, S12 = 0, S20 = 0, S22 = 0;
  for (int k = K - 1; k >= 0; k--) {
    const stage_t *s = w->st + k;
    const double *Q = w->Q + 49 * k;
    double N[49], Y[49], M[49], col[7], out[7];
    if (g_scheme == 1 && k + 1 < K) {        /* P <- Abar' P Abar, p <- Abar' p with Abar = I + cs F_z(z_k) */
      double T[49];
      for (int c = 0; c < 7; c++) {          /* T = P Abar : column c of T = P (Abar e_c) */
        double e[7] = {0}, ae[7]; e[c] = 1.0; abar_mul(s->G, cs, e, ae);
        for (int i = 0; i < 7; i++) { double a = 0; for (int l = 0; l < 7; l++) a += P[i * 7 + l] * ae[l]; T[i * 7 + c] = a; }
      }
      for (int c = 0; c < 7; c++) {          /* P = Abar' T : column c */
        for (int i = 0; i < 7; i++) col[i] = T[i * 7 + c];
        abart_mul(s->G, cs, col, out);
        for (int i = 0; i < 7; i++) P[i * 7 + c] = out[i];
      }
      for (int i = 0; i < 7; i++) for (int j = i + 1; j < 7; j++) { double a = 0.5 * (P[i * 7 + j] + P[j * 7 + i]); P[i * 7 + j] = P[j * 7 + i] = a; }
      for (int j = 0; j < 3; j++) { abart_mul(s->G, cs, p[j], out); memcpy(p[j], out, 56); }
    }
    if (g_form == 1 && k + 1 < K) {          /* step k+1 does not see angle_k: drop its row/column */
      for (int i = 0; i < 7; i++) { P[IA * 7 + i] = 0.0; P[i * 7 + IA] = 0.0; }
      for (int j = 0; j < 3; j++) p[j][IA] = 0.0;
    }
    for (int i = 0; i < 49; i++) N[i] = Q[i] + P[i];
    for (int c = 0; c < 7; c++) {           /* Y = A^-T N */
      for (int i = 0; i < 7; i++) col[i] = N[i * 7 + c];
      solveAT(s, cs, col, out);
      for (int i = 0; i < 7; i++) Y[i * 7 + c] = out[i];
    }
    for (int r = 0; r < 7; r++) {           /* M = Y A^-1 : row r of M = A^-T (row r of Y) */
      solveAT(s, cs, Y + 7 * r, out);
      for (int i = 0; i < 7; i++) M[r * 7 + i] = out[i];
    }
    for (int i = 0; i < 7; i++) for (int j = i + 1; j < 7; j++) {
      double a = 0.5 * (M[i * 7 + j] + M[j * 7 + i]); M[i * 7 + j] = M[j * 7 + i] = a;
    }
    double D = w->R[k] + be * be * M[IB * 7 + IB];
    if (!(D > 0.0)) return 1;
    w->Dp[k] = D;
    double *kap = w->kap + 7 * k;
    for (int i = 0; i < 7; i++) kap[i] = be * M[i * 7 + IB] / D;
    for (int i = 0; i < 7; i++) for (int j = 0; j < 7; j++) P[i * 7 + j] = M[i * 7 + j] - D * kap[i] * kap[j];
    /* three right-hand sides: 0 = residual, 1 = -B_theta, 2 = -B_nu3 */
    double rc[3][7], q[3][7], k0[3];
    for (int j = 0; j < 3; j++) {
      double n[7], nt[7], ruj;
      for (int i = 0; i < 7; i++) {
        double rzj = j == 0 ? -w->rz[7 * k + i] : j == 1 ? -w->gth[7 * k + i] : (k == K - 1 ? -w->e3g[i] : 0.0);
        n[i] = rzj + p[j][i];
        rc[j][i] = j == 0 ? -w->c[7 * k + i] : j == 1 ? hT * s->F[i] : 0.0;
      }
      ruj = j == 0 ? -w->ru[k] : j == 1 ? -w->gu[k] : 0.0;
      solveAT(s, cs, n, nt);
      k0[j] = (be * nt[IB] + ruj) / D;
      for (int i = 0; i < 7; i++) q[j][i] = nt[i] - be * M[i * 7 + IB] * k0[j];
      for (int i = 0; i < 7; i++) {
        double a = 0; for (int l = 0; l < 7; l++) a += P[i * 7 + l] * rc[j][l];
        p[j][i] = q[j][i] - a;
      }
      w->kap0[3 * k + j] = k0[j];
    }
#define BIL(i, j) ({ double a_ = D * k0[i] * k0[j]; \
      for (int l = 0; l < 7; l++) { a_ += 0.5 * (rc[i][l] * (q[j][l] + p[j][l]) + rc[j][l] * (q[i][l] + p[i][l])); } \
      a_; })
    S10 += BIL(1, 0); S11 += BIL(1, 1); S12 += BIL(1, 2); S20 += BIL(2, 0); S22 += BIL(2, 2);
#undef BIL
  }
  /* border 2x2 */
  double a11 = w->sth - S11, a12 = -S12, a22 = -S22;
  double b1 = -w->rth + S10, b2 = -w->c[7 * K] + S20;
  double det = a11 * a22 - a12 * a12;
  if (!(det < 0.0)) return 1;
  double dth = (b1 * a22 - a12 * b2) / det, dnu3 = (a11 * b2 - a12 * b1) / det;
  iter_t st; view(w->step, K, &st);
  memset(w->step, 0, BLOBX(K) * 8);
  st.sc[S_TH] = dth; st.sc[S_NU3] = dnu3;
  double zero[7] = {0};
  for (int k = 0; k < K; k++) {             /* forward */
    const stage_t *s = w->st + k;
    const double *zp = k ? st.z + 7 * (k - 1) : zero, *kap = w->kap + 7 * k;
    double xi[7], azp[7], du = w->kap0[3 * k] + w->kap0[3 * k + 1] * dth + w->kap0[3 * k + 2] * dnu3;
    if (g_scheme == 1 && k > 0) abar_mul((s - 1)->G, cs, zp, azp); else memcpy(azp, zp, 56);
    if (g_form == 1) azp[IA] = 0.0;
    for (int i = 0; i < 7; i++) { xi[i] = azp[i] - w->c[7 * k + i] + hT * s->F[i] * dth; du -= kap[i] * xi[i]; }
    st.u[k] = du;
    xi[IB] += be * du;
    solveA(s, cs, xi, st.z + 7 * k);
  }
  for (int k = K - 1; k >= 0; k--) {        /* adjoint */
    const stage_t *s = w->st + k;
    const double *Q = w->Q + 49 * k, *dz = st.z + 7 * k;
    double r[7];
    for (int i = 0; i < 7; i++) {
      double a = -w->rz[7 * k + i] - w->gth[7 * k + i] * dth - (k == K - 1 ? w->e3g[i] * dnu3 : 0.0);
      for (int l = 0; l < 7; l++) a -= Q[i * 7 + l] * dz[l];
      r[i] = a;
    }
    if (k + 1 < K) {
      double t[7];
      if (g_scheme == 1) abart_mul(s->G, cs, st.lam + 7 * (k + 1), t); else memcpy(t, st.lam + 7 * (k + 1), 56);
      if (g_form == 1) t[IA] = 0.0;
      for (int i = 0; i < 7; i++) r[i] += t[i];
    }
    solveAT(s, cs, r, st.lam + 7 * k);
  }
  /* slacks and their multipliers, bound multipliers */
  const double *dzK = st.z + 7 * (K - 1);
  double ds1 = w->c[7 * K + 1], ds2 = w->c[7 * K + 2];
  for (int i = 0; i < 7; i++) { ds1 += w->g1g[i] * dzK[i]; ds2 += w->g2g[i] * dzK[i]; }
  st.sc[S_S1] = ds1; st.sc[S_S2] = ds2;
  st.sc[S_NU1] = w->sig[0] * ds1 + w->rs[0]; st.sc[S_NU2] = w->sig[1] * ds2 + w->rs[1];
  double s1 = it->sc[S_S1], s2 = it->sc[S_S2];
  st.sc[S_ZS1] = mu / s1 - it->sc[S_ZS1] - it->sc[S_ZS1] / s1 * ds1;
  st.sc[S_ZS2] = mu / s2 - it->sc[S_ZS2] - it->sc[S_ZS2] / s2 * ds2;
  double dl = th - d->tlb, dU = d->tub - th;
  st.sc[S_ZLT] = mu / dl - it->sc[S_ZLT] - it->sc[S_ZLT] / dl * dth;
  st.sc[S_ZUT] = mu / dU - it->sc[S_ZUT] + it->sc[S_ZUT] / dU * dth;
  for (int k = 0; k < K; k++) {
    const double *z = it->z + 7 * k, *zb = it->zb + 6 * k, *dz = st.z + 7 * k;
    double *dzb = st.zb + 6 * k, u = it->u[k], du = st.u[k];
    double lo[3] = {z[IA], z[IM], u + 1.0}, up[3] = {d->aub - z[IA], 1.0 - z[IM], 1.0 - u};
    double dx[3] = {dz[IA], dz[IM], du};
    for (int b = 0; b < 3; b++) {
      dzb[2 * b] = mu / lo[b] - zb[2 * b] - zb[2 * b] / lo[b] * dx[b];
      dzb[2 * b + 1] = mu / up[b] - zb[2 * b + 1] + zb[2 * b + 1] / up[b] * dx[b];
    }
  }
  return 0;
}

/* ---- move penalty: the same sweeps with the control as the eighth state --------------------------------------------
 * Extended state xh = (z, u); the step's extended Jacobian is Ah = [[A, -bu e_IB], [0, 1]] (the control enters defect row
 * IB with coefficient -bu; row 7 is the movement equation), the stage's scalar control delta enters row 7 with -1. */
static void solveA8(const stage_t *s, double cs, double bu, int IB, const double *r, double *v) {
  double t[7]; memcpy(t, r, 56);
  t[IB] += bu * r[7];
  solveA(s, cs, t, v);
  v[7] = r[7];
}
static void solveAT8(const stage_t *s, double cs, double bu, int IB, const double *r, double *v) {
  solveAT(s, cs, r, v);
  v[7] = r[7] + bu * v[IB];
}

static int kkt_solve_mp(const oder *d, work_t *w, const iter_t *it, double mu) {
  int K = w->K; double hT = w->h * d->T, th = it->sc[S_TH], dt = hT * th, be = dt * d->alpha;
  const double cs = g_scheme == 1 ? 0.5 * dt : dt;
  const int IB = g_form == 1 ? IA : IW;
  be = g_form == 1 ? 0.5 * d->aub : be;
  const double dcw = mp_weight(d), dw = w->dw;
  double P[64] = {0}, p[3][8] = {{0}};
  double S10 = 0, S11 = 0, S12 = 0, S20 = 0, S22 = 0;
  for (int k = K - 1; k >= 0; k--) {
    const stage_t *s = w->st + k;
    const double *Q = w->Q + 49 * k;
    double N[64], Y[64], M[64], col[8], out[8];
    if (g_scheme == 1 && k + 1 < K) {        /* P <- Abar' P Abar, p <- Abar' p; the control passes through unchanged */
      double T[64];
      for (int c = 0; c < 8; c++) {
        double e[8] = {0}, ae[8]; e[c] = 1.0; abar_mul(s->G, cs, e, ae); ae[7] = e[7];
        for (int i = 0; i < 8; i++) { double a = 0; for (int l = 0; l < 8; l++) a += P[i * 8 + l] * ae[l]; T[i * 8 + c] = a; }
      }
      for (int c = 0; c < 8; c++) {
        for (int i = 0; i < 8; i++) col[i] = T[i * 8 + c];
        abart_mul(s->G, cs, col, out); out[7] = col[7];
        for (int i = 0; i < 8; i++) P[i * 8 + c] = out[i];
      }
      for (int i = 0; i < 8; i++) for (int j = i + 1; j < 8; j++) { double a = 0.5 * (P[i * 8 + j] + P[j * 8 + i]); P[i * 8 + j] = P[j * 8 + i] = a; }
      for (int j = 0; j < 3; j++) { abart_mul(s->G, cs, p[j], out); out[7] = p[j][7]; memcpy(p[j], out, 64); }
    }
    if (g_form == 1 && k + 1 < K) {
      for (int i = 0; i < 8; i++) { P[IA * 8 + i] = 0.0; P[i * 8 + IA] = 0.0; }
      for (int j = 0; j < 3; j++) p[j][IA] = 0.0;
    }
    memcpy(N, P, sizeof N);
    for (int i = 0; i < 7; i++) for (int j = 0; j < 7; j++) N[i * 8 + j] += Q[i * 7 + j];
    N[63] += w->R[k];                        /* the control's bound terms + delta_w */
    for (int c = 0; c < 8; c++) {
      for (int i = 0; i < 8; i++) col[i] = N[i * 8 + c];
      solveAT8(s, cs, be, IB, col, out);
      for (int i = 0; i < 8; i++) Y[i * 8 + c] = out[i];
    }
    for (int r = 0; r < 8; r++) {
      solveAT8(s, cs, be, IB, Y + 8 * r, out);
      for (int i = 0; i < 8; i++) M[r * 8 + i] = out[i];
    }
    for (int i = 0; i < 8; i++) for (int j = i + 1; j < 8; j++) {
      double a = 0.5 * (M[i * 8 + j] + M[j * 8 + i]); M[i * 8 + j] = M[j * 8 + i] = a;
    }
    double D = w->Rd[k] + M[63];
    if (!(D > 0.0)) return 1;
    w->Dp[k] = D;
    double *kap = w->kap8 + 8 * k;
    for (int i = 0; i < 8; i++) kap[i] = M[i * 8 + 7] / D;
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) P[i * 8 + j] = M[i * 8 + j] - D * kap[i] * kap[j];
    double rc[3][8], q[3][8], k0[3];
    for (int j = 0; j < 3; j++) {
      double n[8], nt[8], ruj;
      for (int i = 0; i < 7; i++) {
        double rzj = j == 0 ? -w->rz[7 * k + i] : j == 1 ? -w->gth[7 * k + i] : (k == K - 1 ? -w->e3g[i] : 0.0);
        n[i] = rzj + p[j][i];
        rc[j][i] = j == 0 ? -w->c[7 * k + i] : j == 1 ? hT * s->F[i] : 0.0;
      }
      n[7] = (j == 0 ? -w->ru[k] : j == 1 ? -w->gu[k] : 0.0) + p[j][7];
      rc[j][7] = j == 0 ? -w->cu[k] : 0.0;
      ruj = j == 0 ? -w->gdl[k] : 0.0;
      solveAT8(s, cs, be, IB, n, nt);
      k0[j] = (nt[7] + ruj) / D;
      for (int i = 0; i < 8; i++) q[j][i] = nt[i] - M[i * 8 + 7] * k0[j];
      for (int i = 0; i < 8; i++) {
        double a = 0; for (int l = 0; l < 8; l++) a += P[i * 8 + l] * rc[j][l];
        p[j][i] = q[j][i] - a;
      }
      w->kap0[3 * k + j] = k0[j];
    }
#define BIL8(i, j) ({ double a_ = D * k0[i] * k0[j]; \
      for (int l = 0; l < 8; l++) { a_ += 0.5 * (rc[i][l] * (q[j][l] + p[j][l]) + rc[j][l] * (q[i][l] + p[i][l])); } \
      a_; })
    S10 += BIL8(1, 0); S11 += BIL8(1, 1); S12 += BIL8(1, 2); S20 += BIL8(2, 0); S22 += BIL8(2, 2);
#undef BIL8
  }
  double a11 = w->sth - S11, a12 = -S12, a22 = -S22;
  double b1 = -w->rth + S10, b2 = -w->c[7 * K] + S20;
  double det = a11 * a22 - a12 * a12;
  if (!(det < 0.0)) return 1;
  double dth = (b1 * a22 - a12 * b2) / det, dnu3 = (a11 * b2 - a12 * b1) / det;
  iter_t st; view(w->step, K, &st);
  memset(w->step, 0, BLOBX(K) * 8);
  st.sc[S_TH] = dth; st.sc[S_NU3] = dnu3;
  double zero[7] = {0};
  for (int k = 0; k < K; k++) {             /* forward */
    const stage_t *s = w->st + k;
    const double *zp = k ? st.z + 7 * (k - 1) : zero, *kap = w->kap8 + 8 * k;
    double xi[8], azp[7], xo[8], dd = w->kap0[3 * k] + w->kap0[3 * k + 1] * dth + w->kap0[3 * k + 2] * dnu3;
    if (g_scheme == 1 && k > 0) abar_mul((s - 1)->G, cs, zp, azp); else memcpy(azp, zp, 56);
    if (g_form == 1) azp[IA] = 0.0;
    for (int i = 0; i < 7; i++) xi[i] = azp[i] - w->c[7 * k + i] + hT * s->F[i] * dth;
    xi[7] = (k ? st.u[k - 1] : 0.0) - w->cu[k];
    for (int i = 0; i < 8; i++) dd -= kap[i] * xi[i];
    w->ddel[k] = dd;
    xi[7] += dd;
    solveA8(s, cs, be, IB, xi, xo);
    memcpy(st.z + 7 * k, xo, 56);
    st.u[k] = xo[7];
  }
  for (int k = K - 1; k >= 0; k--) {        /* adjoint: the seven defect multipliers do not see the movement equations */
    const stage_t *s = w->st + k;
    const double *Q = w->Q + 49 * k, *dz = st.z + 7 * k;
    double r[7];
    for (int i = 0; i < 7; i++) {
      double a = -w->rz[7 * k + i] - w->gth[7 * k + i] * dth - (k == K - 1 ? w->e3g[i] * dnu3 : 0.0);
      for (int l = 0; l < 7; l++) a -= Q[i * 7 + l] * dz[l];
      r[i] = a;
    }
    if (k + 1 < K) {
      double t[7];
      if (g_scheme == 1) abart_mul(s->G, cs, st.lam + 7 * (k + 1), t); else memcpy(t, st.lam + 7 * (k + 1), 56);
      if (g_form == 1) t[IA] = 0.0;
      for (int i = 0; i < 7; i++) r[i] += t[i];
    }
    solveAT(s, cs, r, st.lam + 7 * k);
  }
  /* the movement multiplier from the stage control's own stationarity row, then the slack pair: the slack with the larger
   * curvature from its own row (well conditioned), the other one from delta = p - n (its own row divides a difference of two
   * nearly equal numbers by a curvature that vanishes for an inactive slack) */
  for (int k = 0; k < K; k++) {
    const double pp = it->pp[k], pn = it->pn[k], zp = it->zp[k], zn = it->zn[k], lu = it->lu[k], dd = w->ddel[k];
    const double dlu = w->Rd[k] * dd + w->gdl[k];
    const double sgp = zp / pp + dw, sgn = zn / pn + dw;
    double dpp, dpn;
    if (sgp >= sgn) { dpp = (dlu - (dcw - mu / pp - lu)) / sgp; dpn = dpp - dd; }
    else { dpn = (-dlu - (dcw - mu / pn + lu)) / sgn; dpp = dd + dpn; }
    st.lu[k] = dlu; st.pp[k] = dpp; st.pn[k] = dpn;
    st.zp[k] = (mu - zp * dpp) / pp - zp; st.zn[k] = (mu - zn * dpn) / pn - zn;
  }
  const double *dzK = st.z + 7 * (K - 1);
  double ds1 = w->c[7 * K + 1], ds2 = w->c[7 * K + 2];
  for (int i = 0; i < 7; i++) { ds1 += w->g1g[i] * dzK[i]; ds2 += w->g2g[i] * dzK[i]; }
  st.sc[S_S1] = ds1; st.sc[S_S2] = ds2;
  st.sc[S_NU1] = w->sig[0] * ds1 + w->rs[0]; st.sc[S_NU2] = w->sig[1] * ds2 + w->rs[1];
  double s1 = it->sc[S_S1], s2 = it->sc[S_S2];
  st.sc[S_ZS1] = mu / s1 - it->sc[S_ZS1] - it->sc[S_ZS1] / s1 * ds1;
  st.sc[S_ZS2] = mu / s2 - it->sc[S_ZS2] - it->sc[S_ZS2] / s2 * ds2;
  double dl = th - d->tlb, dU = d->tub - th;
  st.sc[S_ZLT] = mu / dl - it->sc[S_ZLT] - it->sc[S_ZLT] / dl * dth;
  st.sc[S_ZUT] = mu / dU - it->sc[S_ZUT] + it->sc[S_ZUT] / dU * dth;
  for (int k = 0; k < K; k++) {
    const double *z = it->z + 7 * k, *zb = it->zb + 6 * k, *dz = st.z + 7 * k;
    double *dzb = st.zb + 6 * k, u = it->u[k], du = st.u[k];
    double lo[3] = {z[IA], z[IM], u + 1.0}, up[3] = {d->aub - z[IA], 1.0 - z[IM], 1.0 - u};
    double dx[3] = {dz[IA], dz[IM], du};
    for (int b = 0; b < 3; b++) {
      dzb[2 * b] = mu / lo[b] - zb[2 * b] - zb[2 * b] / lo[b] * dx[b];
      dzb[2 * b + 1] = mu / up[b] - zb[2 * b + 1] + zb[2 * b + 1] / up[b] * dx[b];
    }
  }
  return 0;
}

/* barrier objective */
static double barrier(const oder *d, const iter_t *it, double mu) {
  int K = it->K; double th = it->sc[S_TH];
  double sl = log(th - d->tlb) + log(d->tub - th) + log(it->sc[S_S1]) + log(it->sc[S_S2]);
  for (int k = 0; k < K; k++) {
    const double *z = it->z + 7 * k; double u = it->u[k];
    sl += log(z[IA]) + log(d->aub - z[IA]) + log(z[IM]) + log(1.0 - z[IM]) + log(u + 1.0) + log(1.0 - u);
  }
  if (g_mp) {
    double mv = 0.0;
    for (int k = 0; k < K; k++) { mv += it->pp[k] + it->pn[k]; sl += log(it->pp[k]) + log(it->pn[k]); }
    return (th + mp_weight(d) * mv) - mu * sl;
  }
  return th - mu * sl;
}

static double g_err_rd, g_err_cc, g_err_comp, g_err_sd;     /* the pieces of the last kkt_error (trace only) */
static double g_err_rt[11];                                  /* ... its dual part by row type: x y xdot ydot angle angledot mass u | theta s1 s2 */
/* optimality error E_mu (Waechter & Biegler eq. 5) */
static double kkt_error(const oder *d, work_t *w, const iter_t *it, double mu) {
  int K = w->K; double hT = w->h * d->T, th = it->sc[S_TH], dt = hT * th;
  const double cs = g_scheme == 1 ? 0.5 * dt : dt;
  double rd = 0, cc = 0, comp = 0, l1 = 0, zsum = 0, zero7[7] = {0};
  constraints(d, K, w->h, it, w->c);
  for (int i = 0; i < 7 * K + 3; i++) cc = fmax(cc, fabs(w->c[i]));
  double rth = 1.0;
  for (int k = 0; k < K; k++) {
    const double *z = it->z + 7 * k, *l = it->lam + 7 * k, *zb = it->zb + 6 * k;
    const double *ln = (k + 1 < K) ? l + 7 : 0;
    double ax, ay, G[8], F[7], fl[7], r[7], lt[7];
    accel(d, z[IX], z[IY], z[IA], z[IM], 0, 0, &ax, &ay, G, G + 4, 0);
    step_f(d, z, k ? z - 7 : zero7, it->u[k], ax, ay, F);
    for (int i = 0; i < 7; i++) lt[i] = l[i] + ((g_scheme == 1 && ln) ? ln[i] : 0.0);
    fzt(G, lt, fl);
    for (int i = 0; i < 7; i++) { r[i] = l[i] - cs * fl[i] - ((ln && !(g_form == 1 && i == IA)) ? ln[i] : 0.0); rth -= hT * F[i] * l[i]; l1 += fabs(l[i]); }
    r[IA] += -zb[0] + zb[1]; r[IM] += -zb[2] + zb[3];
    if (k == K - 1) {
      double et = z[IY] + d->rho0, rho = sqrt(z[IX] * z[IX] + et * et);
      double nu3 = it->sc[S_NU3], nu1 = it->sc[S_NU1], nu2 = it->sc[S_NU2];
      r[IX] += nu3 * z[IVX] + nu1 * z[IX] / rho; r[IY] += nu3 * z[IVY] + nu1 * et / rho;
      r[IVX] += nu3 * z[IX] + 2 * nu2 * z[IVX]; r[IVY] += nu3 * et + 2 * nu2 * z[IVY];
    }
    for (int i = 0; i < 7; i++) rd = fmax(rd, fabs(r[i]));
    double u = it->u[k];
    double ru = (g_form == 1 ? -0.5 * d->aub * l[IA] : -dt * d->alpha * l[IW]) - zb[4] + zb[5];
    if (g_mp) {
      const double dcw = mp_weight(d);
      ru += it->lu[k] - (k + 1 < K ? it->lu[k + 1] : 0.0);
      rd = fmax(rd, fmax(fabs(dcw - it->lu[k] - it->zp[k]), fabs(dcw + it->lu[k] - it->zn[k])));
      comp = fmax(comp, fmax(fabs(it->pp[k] * it->zp[k] - mu), fabs(it->pn[k] * it->zn[k] - mu)));
      zsum += it->zp[k] + it->zn[k]; l1 += fabs(it->lu[k]);
      cc = fmax(cc, fabs(u_prev_move(it, k)));
    }
    rd = fmax(rd, fabs(ru));
    if (k == 0) memset(g_err_rt, 0, sizeof g_err_rt);
    for (int i = 0; i < 7; i++) g_err_rt[i] = fmax(g_err_rt[i], fabs(r[i]));
    g_err_rt[7] = fmax(g_err_rt[7], fabs(ru));
    double lo[3] = {z[IA], z[IM], u + 1.0}, up[3] = {d->aub - z[IA], 1.0 - z[IM], 1.0 - u};
    for (int b = 0; b < 3; b++) {
      comp = fmax(comp, fmax(fabs(lo[b] * zb[2 * b] - mu), fabs(up[b] * zb[2 * b + 1] - mu)));
      zsum += zb[2 * b] + zb[2 * b + 1];
    }
  }
  rd = fmax(rd, fabs(rth - it->sc[S_ZLT] + it->sc[S_ZUT]));
  rd = fmax(rd, fmax(fabs(-it->sc[S_NU1] - it->sc[S_ZS1]), fabs(-it->sc[S_NU2] - it->sc[S_ZS2])));
  g_err_rt[8] = fabs(rth - it->sc[S_ZLT] + it->sc[S_ZUT]); g_err_rt[9] = fabs(-it->sc[S_NU1] - it->sc[S_ZS1]); g_err_rt[10] = fabs(-it->sc[S_NU2] - it->sc[S_ZS2]);
  comp = fmax(comp, fmax(fabs((th - d->tlb) * it->sc[S_ZLT] - mu), fabs((d->tub - th) * it->sc[S_ZUT] - mu)));
  comp = fmax(comp, fmax(fabs(it->sc[S_S1] * it->sc[S_ZS1] - mu), fabs(it->sc[S_S2] * it->sc[S_ZS2] - mu)));
  l1 += fabs(it->sc[S_NU3]) + fabs(it->sc[S_NU1]) + fabs(it->sc[S_NU2]);
  zsum += it->sc[S_ZLT] + it->sc[S_ZUT] + it->sc[S_ZS1] + it->sc[S_ZS2];
  double sd = fmax(100.0, (l1 + zsum) / (double)(7 * K + 3 + 6 * K + 4 + (g_mp ? 3 * K : 0))) / 100.0;
  g_err_rd = rd; g_err_cc = cc; g_err_comp = comp; g_err_sd = sd;
  return fmax(fmax(rd / sd, cc), comp / sd);
}

/* straight-line initial guess toward a tangential insertion point (same as ascent_numpy.initial_guess) */
static void initial_guess(const oder *d, int K, double h, iter_t *it) {
  double tf0 = 0.9, dr = 0.166, aend = 0.5, vp = sqrt(d->vp2), dt = h * d->T * tf0;
  double xf = -d->rhof * sin(dr), yf = d->rhof * cos(dr) - d->rho0;
  for (int k = 0; k < K; k++) {
    double fr = (double)(k + 1) / K, *z = it->z + 7 * k;
    z[IX] = fr * xf; z[IY] = fr * yf; z[IVX] = -fr * vp * cos(dr); z[IVY] = -fr * vp * sin(dr);
    z[IA] = fr * aend; z[IW] = aend / (K * dt); z[IM] = d->mrate * dt * (k + 1);
    it->u[k] = 0.0;
    if (g_form == 1) { z[IW] = 0.0; it->u[k] = z[IA] / (0.5 * d->aub) - 1.0; }
  }
  it->sc[S_TH] = tf0;
}

static double push(double v, double lb, double ub) {
  double k1 = 1e-2, pl = fmin(k1 * fmax(1.0, fabs(lb)), k1 * (ub - lb)), pu = fmin(k1 * fmax(1.0, fabs(ub)), k1 * (ub - lb));
  return fmin(fmax(v, lb + pl), ub - pu);
}

enum { ST_CONVERGED = 0, ST_MAXITER = 1, ST_LINESEARCH = 2, ST_REGULARISATION = 3 };

/* barrier schedule (Waechter & Biegler 2006 defaults: mu0 0.1, kappa_eps 10, kappa_mu 0.2, theta_mu 1.5);
 * oracle_set_barrier_schedule exists for experiments only -- the product uses the defaults */
static double g_mu0 = 0.1, g_keps = 10.0, g_kmu = 0.2, g_thmu = 1.5;
void oracle_set_barrier_schedule(double mu0, double keps, double kmu, double thmu) { g_mu0 = mu0; g_keps = keps; g_kmu = kmu; g_thmu = thmu; }

/* one NLP.  blob: in = initial guess (primal part used) if use_guess, out = solution iterate. */
/* warm: 0 = built-in cold-start guess, 1 = primal part of the blob, 2 = the full primal-dual blob (multipliers and
 * slacks kept, floored away from zero).  A blob whose theta is not positive is treated as "no guess" (that is how the
 * nested iteration marks problems whose coarse solve failed).  mu0 <= 0: 0.1 cold, 1e-4 warm. */
static int g_trace = 0;       /* oracle_set_trace(1): one line per interior-point iteration on stderr (diagnostics) */
void oracle_set_trace(int on) { g_trace = on; }

static int solve_one(const oparams *prm, int nt, int max_iter, double tol, int warm, double mu0, double *blob,
                     int *iters_out, int *nreg_out) {
  int K = nt - 1; oder d; derive(prm, &d);
  work_t *w = work_new(K);
  iter_t it, tr, st; view(blob, K, &it); view(w->trial, K, &tr); view(w->step, K, &st);
  const int asked_warm = warm;
  if (warm && !(it.sc[S_TH] > 0.0)) warm = 0;
  if (!warm) { memset(blob, 0, BLOBX(K) * 8); initial_guess(&d, K, w->h, &it); }
  const double s1g = it.sc[S_S1], s2g = it.sc[S_S2];
  it.sc[S_S1] = it.sc[S_S2] = 0.0;
  constraints(&d, K, w->h, &it, w->c);
  it.sc[S_S1] = fmax(w->c[7 * K + 1], 1e-2); it.sc[S_S2] = fmax(w->c[7 * K + 2], 1e-4);
  for (int k = 0; k < K; k++) {
    double *z = it.z + 7 * k;
    z[IA] = push(z[IA], 0.0, d.aub); z[IM] = push(z[IM], 0.0, 1.0); it.u[k] = push(it.u[k], -1.0, 1.0);
    for (int b = 0; b < 6; b++) it.zb[6 * k + b] = warm == 2 ? fmax(it.zb[6 * k + b], 1e-12) : 1.0;
  }
  it.sc[S_TH] = push(it.sc[S_TH], d.tlb, d.tub);
  if (warm == 2) {
    it.sc[S_S1] = fmax(s1g, 1e-10); it.sc[S_S2] = fmax(s2g, 1e-10);
    it.sc[S_ZLT] = fmax(it.sc[S_ZLT], 1e-12); it.sc[S_ZUT] = fmax(it.sc[S_ZUT], 1e-12);
    it.sc[S_ZS1] = fmax(it.sc[S_ZS1], 1e-12); it.sc[S_ZS2] = fmax(it.sc[S_ZS2], 1e-12);
  } else {
    it.sc[S_S1] = fmax(it.sc[S_S1], 1e-2); it.sc[S_S2] = fmax(it.sc[S_S2], 1e-2);
    it.sc[S_ZLT] = it.sc[S_ZUT] = it.sc[S_ZS1] = it.sc[S_ZS2] = 1.0;
    memset(it.lam, 0, 7 * K * 8); it.sc[S_NU3] = it.sc[S_NU1] = it.sc[S_NU2] = 0.0;
  }
  const double dcw = g_mp ? mp_weight(&d) : 0.0;
  double mu = (asked_warm && !warm) ? g_mu0 : mu0 > 0.0 ? mu0 : (warm ? 1e-4 : g_mu0), nu_pen = 1.0, dw_last = 0.0;
  if (g_mp) {      /* slacks of the movement equations around the guess's own movement; multipliers that zero their stationarity rows */
    const double eps = warm ? 1e-4 : 1e-2;
    for (int k = 0; k < K; k++) {
      const double dl = it.u[k] - (k ? it.u[k - 1] : mp_u_init());
      it.lu[k] = 0.0; it.pp[k] = fmax(dl, 0.0) + eps; it.pn[k] = fmax(-dl, 0.0) + eps; it.zp[k] = it.zn[k] = dcw;
    }
  }
  int status = ST_MAXITER, iters = 0, nreg = 0;
  for (int iter = 0; iter < max_iter; iter++) {
    double e0 = kkt_error(&d, w, &it, 0.0);
    const double t_rd = g_err_rd, t_cc = g_err_cc, t_comp = g_err_comp, t_sd = g_err_sd;
    double t_rt[11]; memcpy(t_rt, g_err_rt, sizeof t_rt);
    if (e0 <= tol) { status = ST_CONVERGED; break; }
    while (mu > tol / 10.0 && kkt_error(&d, w, &it, mu) <= g_keps * mu) {
      mu = fmax(tol / 10.0, fmin(g_kmu * mu, pow(mu, g_thmu)));
      nu_pen = 1.0;
    }
    double dw = 0.0; int fail = 0;
    for (;;) {
      assemble(&d, w, &it, mu, dw);
      if ((g_mp ? kkt_solve_mp(&d, w, &it, mu) : kkt_solve(&d, w, &it, mu)) == 0) break;
      /* inertia correction: first one of a solve 1e-2, later ones a third of the last successful value, x10 while wrong */
      dw = dw == 0.0 ? (dw_last == 0.0 ? 1e-2 : fmax(1e-4, dw_last / 3.0)) : dw * 10.0;
      nreg++;
      if (dw > 1e10) { fail = 1; break; }
    }
    if (fail) { status = ST_REGULARISATION; break; }
    dw_last = dw;
    /* fraction to the boundary */
    double tau = fmax(0.99, 1.0 - mu), apr = 1.0, adu = 1.0;
#define FTB(a, val, dv) do { if ((dv) < 0) a = fmin(a, -tau * (val) / (dv)); } while (0)
    double th = it.sc[S_TH];
    FTB(apr, th - d.tlb, st.sc[S_TH]); FTB(apr, d.tub - th, -st.sc[S_TH]);
    FTB(apr, it.sc[S_S1], st.sc[S_S1]); FTB(apr, it.sc[S_S2], st.sc[S_S2]);
    FTB(adu, it.sc[S_ZLT], st.sc[S_ZLT]); FTB(adu, it.sc[S_ZUT], st.sc[S_ZUT]);
    FTB(adu, it.sc[S_ZS1], st.sc[S_ZS1]); FTB(adu, it.sc[S_ZS2], st.sc[S_ZS2]);
    double gd = st.sc[S_TH] * (1.0 - mu / (th - d.tlb) + mu / (d.tub - th))
              - mu * st.sc[S_S1] / it.sc[S_S1] - mu * st.sc[S_S2] / it.sc[S_S2];
    for (int k = 0; k < K; k++) {
      const double *z = it.z + 7 * k, *dz = st.z + 7 * k; double u = it.u[k], du = st.u[k];
      FTB(apr, z[IA], dz[IA]); FTB(apr, d.aub - z[IA], -dz[IA]);
      FTB(apr, z[IM], dz[IM]); FTB(apr, 1.0 - z[IM], -dz[IM]);
      FTB(apr, u + 1.0, du); FTB(apr, 1.0 - u, -du);
      for (int b = 0; b < 6; b++) FTB(adu, it.zb[6 * k + b], st.zb[6 * k + b]);
      gd += dz[IA] * (-mu / z[IA] + mu / (d.aub - z[IA])) + dz[IM] * (-mu / z[IM] + mu / (1.0 - z[IM]))
          + du * (-mu / (u + 1.0) + mu / (1.0 - u));
    }
    if (g_mp) for (int k = 0; k < K; k++) {
      FTB(apr, it.pp[k], st.pp[k]); FTB(apr, it.pn[k], st.pn[k]);
      FTB(adu, it.zp[k], st.zp[k]); FTB(adu, it.zn[k], st.zn[k]);
      gd += dcw * (st.pp[k] + st.pn[k]) - mu * (st.pp[k] / it.pp[k] + st.pn[k] / it.pn[k]);
    }
    /* l1 merit (Nocedal & Wright eq. 18.36); curvature from the Newton identity dx'H dx = -gd + c'(lam+dlam) */
    double c1 = 0, cl = 0;
    if (g_mp) for (int k = 0; k < K; k++) { c1 += fabs(w->cu[k]); cl += w->cu[k] * (it.lu[k] + st.lu[k]); }
    for (int i = 0; i < 7 * K; i++) { c1 += fabs(w->c[i]); cl += w->c[i] * (it.lam[i] + st.lam[i]); }
    for (int j = 0; j < 3; j++) {
      static const int nu_ix[3] = {S_NU3, S_NU1, S_NU2};
      c1 += fabs(w->c[7 * K + j]); cl += w->c[7 * K + j] * (it.sc[nu_ix[j]] + st.sc[nu_ix[j]]);
    }
    double curv = -gd + cl;
    if (c1 > 0) { double need = (gd + 0.5 * fmax(curv, 0.0)) / (0.9 * c1); if (nu_pen < need) nu_pen = need + 1.0; }
    double Dm = gd - nu_pen * c1, phi0 = barrier(&d, &it, mu) + nu_pen * c1, alpha = apr;
    int ok = 0;
    for (int ls = 0; ls < 40; ls++) {
      memcpy(w->trial, blob, BLOBX(K) * 8);
      for (int i = 0; i < 8 * K; i++) w->trial[i] += alpha * w->step[i];
      if (g_mp) for (int k = 0; k < K; k++) { tr.pp[k] += alpha * st.pp[k]; tr.pn[k] += alpha * st.pn[k]; }
      tr.sc[S_TH] += alpha * st.sc[S_TH]; tr.sc[S_S1] += alpha * st.sc[S_S1]; tr.sc[S_S2] += alpha * st.sc[S_S2];
      constraints(&d, K, w->h, &tr, w->ctrial);
      double ct = 0; for (int i = 0; i < 7 * K + 3; i++) ct += fabs(w->ctrial[i]);
      if (g_mp) { move_constraints(K, &tr, w->cutrial); for (int k = 0; k < K; k++) ct += fabs(w->cutrial[k]); }
      double phit = barrier(&d, &tr, mu) + nu_pen * ct;
      if (isfinite(phit) && phit <= phi0 + 1e-8 * alpha * Dm + 10 * 2.220446049250313e-16 * fabs(phi0)) { ok = 1; break; }
      alpha *= 0.5;
    }
    if (!ok) { status = ST_LINESEARCH; break; }
    if (g_trace) fprintf(stderr, "[oracle] K=%d iter %2d mu %.1e E0 %.2e (dual %.1e primal %.1e compl %.1e s_d %.2g) Emu %.2e alpha %.3g (apr %.3g) adu %.3g dw %.1e nu_pen %.2g c1 %.2e\n", K, iter, mu, e0,
                         t_rd, t_cc, t_comp, t_sd, kkt_error(&d, w, &it, mu), alpha, apr, adu, dw, nu_pen, c1);
    if (g_trace) fprintf(stderr, "[oracle]       dual rows: x %.2e y %.2e vx %.2e vy %.2e a %.2e w %.2e m %.2e u %.2e | th %.2e s1 %.2e s2 %.2e\n", t_rt[0], t_rt[1], t_rt[2],
                         t_rt[3], t_rt[4], t_rt[5], t_rt[6], t_rt[7], t_rt[8], t_rt[9], t_rt[10]);
    for (int i = 0; i < 8 * K; i++) blob[i] += alpha * w->step[i];
    for (int i = 0; i < 7 * K; i++) it.lam[i] += alpha * st.lam[i];
    for (int i = 0; i < 6 * K; i++) it.zb[i] += adu * st.zb[i];
    if (g_mp) for (int k = 0; k < K; k++) {
      it.lu[k] += alpha * st.lu[k]; it.pp[k] += alpha * st.pp[k]; it.pn[k] += alpha * st.pn[k];
      it.zp[k] += adu * st.zp[k]; it.zn[k] += adu * st.zn[k];
    }
    it.sc[S_TH] += alpha * st.sc[S_TH]; it.sc[S_S1] += alpha * st.sc[S_S1]; it.sc[S_S2] += alpha * st.sc[S_S2];
    it.sc[S_NU3] += alpha * st.sc[S_NU3]; it.sc[S_NU1] += alpha * st.sc[S_NU1]; it.sc[S_NU2] += alpha * st.sc[S_NU2];
    it.sc[S_ZLT] += adu * st.sc[S_ZLT]; it.sc[S_ZUT] += adu * st.sc[S_ZUT];
    it.sc[S_ZS1] += adu * st.sc[S_ZS1]; it.sc[S_ZS2] += adu * st.sc[S_ZS2];
    /* keep z within [mu/(k d), k mu/d], k = 1e10 (Waechter & Biegler eq. 16) */
#define CLIP(zv, dist) zv = fmin(fmax(zv, mu / (1e10 * (dist))), 1e10 * mu / (dist))
    th = it.sc[S_TH];
    CLIP(it.sc[S_ZLT], th - d.tlb); CLIP(it.sc[S_ZUT], d.tub - th);
    CLIP(it.sc[S_ZS1], it.sc[S_S1]); CLIP(it.sc[S_ZS2], it.sc[S_S2]);
    for (int k = 0; k < K; k++) {
      double *z = it.z + 7 * k, *zb = it.zb + 6 * k, u = it.u[k];
      CLIP(zb[0], z[IA]); CLIP(zb[1], d.aub - z[IA]); CLIP(zb[2], z[IM]); CLIP(zb[3], 1.0 - z[IM]);
      CLIP(zb[4], u + 1.0); CLIP(zb[5], 1.0 - u);
      if (g_mp) { CLIP(it.zp[k], it.pp[k]); CLIP(it.zn[k], it.pn[k]); }
    }
    iters = iter + 1;
  }
  *iters_out = iters; if (nreg_out) *nreg_out = nreg;
  work_free(w);
  return status;
}

/* ================================ exported (ctypes) =========================================== */
int oracle_blob_size(int nt) { return BLOB(nt - 1); }

/* accelerations + gradients + weighted Hessian for n points (parity of the eval kernel) */
void oracle_accel(const double *params, int n, const double *x, const double *y, const double *a,
                  const double *m, const double *px, const double *py, double *ax, double *ay,
                  double *gax, double *gay, double *H) {
  oder d; derive((const oparams *)params, &d);
  for (int i = 0; i < n; i++)
    accel(&d, x[i], y[i], a[i], m[i], px[i], py[i], ax + i, ay + i, gax + 4 * i, gay + 4 * i, H + 10 * i);
}

/* equality-constraint values at an iterate blob */
void oracle_set_scheme(int scheme) { g_scheme = scheme == 1 ? 1 : 0; }
void oracle_set_formulation(int form) { g_form = form == 1 ? 1 : 0; }
void oracle_set_move_penalty(int on) { g_mp = on ? 1 : 0; }
int oracle_get_scheme(void) { return g_scheme; }

void oracle_constraints(const double *params, int nt, const double *blob, double *c) {
  oder d; derive((const oparams *)params, &d);
  iter_t it; view((double *)blob, nt - 1, &it);
  constraints(&d, nt - 1, 1.0 / (nt - 1), &it, c);
}

/* one Newton step of the barrier problem at an arbitrary interior iterate; returns inertia flag */
int oracle_newton_step(const double *params, int nt, const double *blob, double mu, double delta_w,
                       double *step) {
  int K = nt - 1; oder d; derive((const oparams *)params, &d);
  work_t *w = work_new(K); iter_t it; view((double *)blob, K, &it);
  double *xb = 0;
  if (g_mp) {      /* the external blob has no slack rows: they are set around the iterate's own movement, as a warm start does */
    xb = malloc(BLOBX(K) * 8); memcpy(xb, blob, BLOB(K) * 8); view(xb, K, &it);
    const double dcw = mp_weight(&d);
    for (int k = 0; k < K; k++) {
      const double dl = it.u[k] - (k ? it.u[k - 1] : mp_u_init());
      it.lu[k] = 0.0; it.pp[k] = fmax(dl, 0.0) + 1e-4; it.pn[k] = fmax(-dl, 0.0) + 1e-4; it.zp[k] = it.zn[k] = dcw;
    }
  }
  assemble(&d, w, &it, mu, delta_w);
  int rc = g_mp ? kkt_solve_mp(&d, w, &it, mu) : kkt_solve(&d, w, &it, mu);
  memcpy(step, w->step, BLOB(K) * 8);
  free(xb);
  work_free(w);
  return rc;
}

double oracle_kkt_error(const double *params, int nt, const double *blob, double mu) {
  int K = nt - 1; oder d; derive((const oparams *)params, &d);
  work_t *w = work_new(K); iter_t it; view((double *)blob, K, &it);
  double e = kkt_error(&d, w, &it, mu);
  work_free(w);
  return e;
}

/* ---- nested iteration (mesh continuation) ------------------------------------------------------------------
 * A cold start on a grid of >= 40 nodes first solves the same NLP on a grid of three tenths of the nodes (recursively in
 * the automatic mode: 201 -> 60 -> 17), prolongs that primal-dual solution to the next grid (linear in tau; node 0 is the
 * fixed initial state; bound multipliers scale with the step) and warm-starts the solve there: mu0 = 1e-6 when the guess
 * comes from the cold-started coarsest grid, mu0 = max(1e-9, tol/100) (tol of the finest grid) when it comes from a grid that
 * was warm-started itself.  The
 * coarse levels are solved to max(tol, 1e-3) only: their discretisation error is 1e-2.  A problem whose coarse solve does
 * not converge starts cold on the fine grid. */
static int coarse_of(int nt) {      /* one to three intervals beyond a multiple of 16 are given up (18 nodes -> 17): the GPU kernels work in 16-interval chunks */
  int c = (3 * nt + 5) / 10;
  if (c < 14) c = 14;
  const int over = (c - 1) % 16;
  if (c > 17 && over >= 1 && over <= 3) c -= over;
  return c;
}
#define NESTED_MIN_NODES 40
#define NESTED_MU_FIRST 1e-6
#define NESTED_MU_NEXT(tol_finest) fmax(1e-9, 1e-2 * (tol_finest))
/* ... with the move penalty: 1e-5 and max(1e-8, 10 tol).  The slack pairs of the movement equations are re-centred on every grid
 * (solve_one), and wherever the control's movement changes sign between the prolonged guess and the grid's own solution a pair
 * has to swap roles through the kink of |.|: at a barrier parameter of 1e-9 that costs the affected problems up to seven
 * fraction-to-boundary-limited iterations (config-3 sweep, 200-node grid: 9-16 iterations, mean 10.9; with these starts 9-13,
 * mean 10.5, and 3-4 instead of 4-5 on the 60-node grid; the unpenalised problem is best left at 1e-6 / 1e-9). */
#define NESTED_MU_FIRST_MP 1e-5
#define NESTED_MU_NEXT_MP(tol_finest) fmax(1e-8, 10.0 * (tol_finest))
#define NESTED_COARSE_TOL 1e-3    /* the coarse levels are solved to the reference's own OTOL/RTOL, not to `tol` */

static void prolong(const double *bc, int Kc, double *bf, int Kf) {
  iter_t c, f; view((double *)bc, Kc, &c); view(bf, Kf, &f);
  const double zsc = (double)Kc / (double)Kf;
  for (int k = 0; k < Kf; k++) {
    const double x = (double)(k + 1) / (double)Kf * (double)Kc;
    int j = (int)x; if (j > Kc - 1) j = Kc - 1;
    const double wt = x - (double)j;
    const int ja = j ? j - 1 : 0;                 /* coarse record of the left node (node 0 has none) */
    for (int i = 0; i < 7; i++) {
      const double a = j ? c.z[7 * ja + i] : ((g_form == 1 && i == IA) ? c.z[i] : 0.0), b = c.z[7 * j + i];
      f.z[7 * k + i] = fma(wt, b - a, a);
      const double la = c.lam[7 * ja + i], lb = c.lam[7 * j + i];
      f.lam[7 * k + i] = fma(wt, lb - la, la);
    }
    { const double a = c.u[ja], b = c.u[j]; f.u[k] = fma(wt, b - a, a); }
    for (int b6 = 0; b6 < 6; b6++) {
      const double a = c.zb[6 * ja + b6], b = c.zb[6 * j + b6];
      f.zb[6 * k + b6] = fma(wt, b - a, a) * zsc;
    }
  }
  memcpy(f.sc, c.sc, 10 * 8);
}

/* blob: out = solution on the nt-grid.  coarse: 0 automatic, -1 single grid, > 0 that many nodes (two levels) */
static int solve_nested(const oparams *prm, int nt, int max_iter, double tol, double tol_finest, int coarse, double *blob, int *iters_out, int *depth_out) {
  int nc = coarse > 0 ? coarse : coarse_of(nt);
  if (coarse == -1 || (coarse == 0 && nt < NESTED_MIN_NODES) || nc >= nt || nc < 3) {
    if (depth_out) *depth_out = 0;
    return solve_one(prm, nt, max_iter, tol, 0, 0.0, blob, iters_out, 0);
  }
  double *bc = malloc(BLOBX(nc - 1) * 8);
  int itc = 0, itf = 0, depth_c = 0;
  const int stc = solve_nested(prm, nc, max_iter, fmax(tol, NESTED_COARSE_TOL), tol_finest, coarse > 0 ? -1 : 0, bc, &itc, &depth_c);
  int warm = 0;
  if (stc == ST_CONVERGED) { prolong(bc, nc - 1, blob, nt - 1); warm = 2; }
  free(bc);
  const double mu_first = g_mp ? NESTED_MU_FIRST_MP : NESTED_MU_FIRST, mu_next = g_mp ? NESTED_MU_NEXT_MP(tol_finest) : NESTED_MU_NEXT(tol_finest);
  const int st = solve_one(prm, nt, max_iter, tol, warm, depth_c == 0 ? mu_first : mu_next, blob, &itf, 0);
  *iters_out = itc + itf;
  if (depth_out) *depth_out = depth_c + 1;
  return st;
}

static int g_coarse = 0;     /* 0 automatic nested iteration, -1 single grid, > 0 explicit coarse grid */
void oracle_set_coarse_nodes(int c) { g_coarse = c; }
static double g_warm_mu0 = 0.0; static int g_warm_mode = 1;
void oracle_set_warm_start(int mode, double mu0) { g_warm_mode = mode; g_warm_mu0 = mu0; }
void oracle_prolong(const double *blob_c, int nt_c, double *blob_f, int nt_f) { prolong(blob_c, nt_c - 1, blob_f, nt_f - 1); }

/* batch solve. params [batch][16]; traj_out [batch][10][nt] with fields
 * x y xdot ydot xdoubledot ydoubledot angle angledot angledoubledot mass (reference .value lists);
 * blob_out (optional) [batch][blob] full primal-dual solution.  With a guess: warm start (oracle_set_warm_start:
 * mode 1 primal / 2 primal-dual, mu0); without: cold start, nested iteration per oracle_set_coarse_nodes. */
int oracle_solve_batch(const double *params, int batch, int nt, int max_iter, double tol,
                       const double *guess_blob_or_null, double *traj_out, double *tf_out,
                       int *status_out, int *iters_out, double *blob_out_or_null) {
  int K = nt - 1;
  for (int b = 0; b < batch; b++) {
    const oparams *prm = (const oparams *)(params + 16 * b);
    double *blob = malloc(BLOBX(K) * 8);
    int iters = 0, st;
    if (guess_blob_or_null) {
      memcpy(blob, guess_blob_or_null + (size_t)b * BLOB(K), BLOB(K) * 8);
      st = solve_one(prm, nt, max_iter, tol, g_warm_mode, g_warm_mu0, blob, &iters, 0);
    } else {
      st = solve_nested(prm, nt, max_iter, tol, tol, g_coarse, blob, &iters, 0);
    }
    status_out[b] = st; iters_out[b] = iters;
    iter_t it; view(blob, K, &it);
    tf_out[b] = it.sc[S_TH];
    if (traj_out) {
      oder d; derive(prm, &d);
      double *t = traj_out + (size_t)b * 10 * nt;
      for (int k = 0; k < nt; k++) {
        double zz[7] = {0}, u = 0, ax, ay;
        if (k) { memcpy(zz, it.z + 7 * (k - 1), 56); u = it.u[k - 1]; }
        accel(&d, zz[IX], zz[IY], zz[IA], zz[IM], 0, 0, &ax, &ay, 0, 0, 0);
        t[0 * nt + k] = zz[IX]; t[1 * nt + k] = zz[IY]; t[2 * nt + k] = zz[IVX]; t[3 * nt + k] = zz[IVY];
        t[4 * nt + k] = ax; t[5 * nt + k] = ay; t[6 * nt + k] = zz[IA]; t[7 * nt + k] = zz[IW];
        t[8 * nt + k] = u; t[9 * nt + k] = zz[IM];
      }
    }
    if (blob_out_or_null) memcpy(blob_out_or_null + (size_t)b * BLOB(K), blob, BLOB(K) * 8);
    free(blob);
  }
  return 0;
}
