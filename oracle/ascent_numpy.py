"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the lunar-ascent NLP (the oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (lunar_module_ascent_trajectory_optimiser_amd/) never does.

What is restated (reference = /root/reference/Launch_Optimiser.py, "LO" below):
  * time grid tau_k = k/(nt-1), NODES=2 two-point collocation = backward Euler   LO:20-21,25
  * parameters, scale factors                                                   LO:38-40,50-75,107-109
  * variables and bounds (mass, y, ydot, x, xdot, angle, angledot, MV u)         LO:83-100
  * the 7 scaled ODEs                                                           LO:114-123
  * the algebraic accelerations ydoubledot / xdoubledot                          LO:127-136
  * initial conditions                                                          LO:145-151
  * the three terminal constraints (applied at the last node only; the           LO:158-173
    reference masks them to the last node with 0/1 parameter arrays)
  * objective min tf                                                            LO:176
  * output re-dimensionalisation                                                LO:187-202
The "v1" formulation (angle itself is the manipulated variable, circular target,
mass_scalar=2576 quirk) follows the PDF appendix p26-28 as transcribed in SURVEY.md B.2.

The NLP solve itself (GEKKO -> APMonitor -> IPOPT, LO:177) lives in third-party code
that is not in /root/reference and is not installed (gekko, unpinned); its published
algorithm (primal-dual interior point, Waechter & Biegler 2006) is restated here with a
generic sparse LU on the full KKT matrix -- deliberately NOT the stage-structured
recursion the HIP product uses, so that the two are independent.  Parity is pinned on
the reference's own artefacts: Numerical_results.png (current script) and PDF p30 (v1),
committed as tests/golden/golden.json.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

# field order of the per-node state vector used throughout this repo
X, Y, VX, VY, A, W, MS = range(7)
NS = 7          # states per node
NW = 8          # states + control per node (u last)


@dataclasses.dataclass
class Params:
    """One NLP's physical parameters, SI units (LO:38-75, 107-109)."""
    G: float = 6.674e-11            # LO:50
    M: float = 7.346e22             # LO:51
    R0: float = 1738100.0           # LO:52
    Ft: float = 15346.0             # LO:61
    M0: float = 4821.0              # LO:62
    mdot: float = 5.053             # LO:63,65
    fuel_mass: float = 2376.0       # LO:64  (mflow = mdot/fuel_mass)
    mass_scalar: float = 2376.0     # LO:108 (2576 in v1: PDF p26)
    ang_acc_max: float = 5e-4       # LO:66
    r_peri: float = 17703.0         # LO:70  (= Rfmin = Scalar, LO:73,107)
    r_apo: float = 88615.0          # LO:71
    T_scale: float = 470.0          # LO:38
    angle_ub: float = math.pi / 3   # LO:94
    tf_lb: float = 0.0              # LO:39
    tf_ub: float = 1.0              # LO:39
    dcost: float = 0.0              # LO:99 (movement penalty)

    def derived(self):
        S = self.r_peri
        GM = self.G * self.M
        r_avg = 0.5 * (self.r_peri + self.r_apo)                  # LO:72
        vper = math.sqrt(GM / (self.R0 + r_avg))                  # LO:75
        return dict(
            S=S, GM=GM, rho0=self.R0 / S, rhof=(self.R0 + S) / S,
            vp2=(vper / S) ** 2, vper=vper,
            gam=GM / S ** 3, thr=self.Ft / S,
            alpha=self.ang_acc_max / 3.0,                         # LO:109
            beta=self.mdot / self.fuel_mass,                      # LO:65
        )


def v1_params() -> Params:
    """The v1 script of the PDF appendix."""
    return Params(r_peri=53108.4, r_apo=53108.4, mass_scalar=2576.0)


# ----------------------------------------------------------------------------------
# dynamics: scaled accelerations (LO:127-136) with first and second derivatives
# ----------------------------------------------------------------------------------
def accel(x, y, a, m, P: Params, px=None, py=None):
    """ax, ay (LO:133-136, 127-130), their gradients w.r.t. (x, y, a, m) and, when
    weights (px, py) are given, the Hessian of px*ax + py*ay (10 upper-triangle
    entries in the order xx xy xa xm yy ya ym aa am mm).  Vectorised over nodes."""
    d = P.derived()
    xi = np.asarray(x, float)
    et = np.asarray(y, float) + d["rho0"]
    rho2 = xi * xi + et * et
    rho = np.sqrt(rho2)
    ir = 1.0 / rho
    ex, ey = xi * ir, et * ir
    c, s = np.cos(3.0 * a), np.sin(3.0 * a)
    dx, dy = ex * c - ey * s, ey * c + ex * s          # thrust direction
    mp = P.M0 - P.mass_scalar * m
    th = d["thr"] / mp                                   # Ft/(S*mp)
    th1 = th * P.mass_scalar / mp                        # d th/dm
    th2 = 2.0 * th1 * P.mass_scalar / mp                 # d2 th/dm2
    g3 = d["gam"] * ir * ir * ir                         # gamma/rho^3
    ax = th * dx - g3 * xi
    ay = th * dy - g3 * et
    # gradient
    fx, fy = -ey * ir, ex * ir                           # d(polar angle)/d(x,y)
    pxd, pyd = -dy, dx                                   # d-perp
    gax = np.stack([th * pxd * fx - g3 * (1 - 3 * ex * ex),
                    th * pxd * fy - g3 * (-3 * ex * ey),
                    3 * th * pxd,
                    th1 * dx], -1)
    gay = np.stack([th * pyd * fx - g3 * (-3 * ex * ey),
                    th * pyd * fy - g3 * (1 - 3 * ey * ey),
                    3 * th * pyd,
                    th1 * dy], -1)
    if px is None:
        return ax, ay, gax, gay
    pd = px * dx + py * dy
    pp = px * pxd + py * pyd
    pe = px * ex + py * ey
    fxx, fxy, fyy = 2 * ex * ey * ir * ir, (ey * ey - ex * ex) * ir * ir, -2 * ex * ey * ir * ir
    g4 = 3.0 * g3 * ir                                   # 3*gamma/rho^4
    hxx = th * (-pd * fx * fx + pp * fxx) + g4 * (2 * px * ex + pe - 5 * pe * ex * ex)
    hxy = th * (-pd * fx * fy + pp * fxy) + g4 * (px * ey + py * ex - 5 * pe * ex * ey)
    hyy = th * (-pd * fy * fy + pp * fyy) + g4 * (2 * py * ey + pe - 5 * pe * ey * ey)
    hxa = -3 * th * pd * fx
    hya = -3 * th * pd * fy
    haa = -9 * th * pd
    hxm = th1 * pp * fx
    hym = th1 * pp * fy
    ham = 3 * th1 * pp
    hmm = th2 * pd
    H = np.stack([hxx, hxy, hxa, hxm, hyy, hya, hym, haa, ham, hmm], -1)
    return ax, ay, gax, gay, H


# ----------------------------------------------------------------------------------
# the NLP in flat-vector form (generic, for the sparse-LU interior point below)
# ----------------------------------------------------------------------------------
class AscentNLP:
    """min tf  s.t. backward-Euler defects, terminal constraints, bounds.

    formulation 0 = current script  (7 states + control u = angledoubledot)
    formulation 1 = v1 script       (5 states x,y,vx,vy,mass + control = angle)
    Unknown vector v = [w_1 .. w_K, tf, s1, s2], w_k = node k's states then control,
    K = nt-1 (node 0 is fixed by the initial conditions, LO:145-151).
    Equalities: K*ns defects, then e3 (LO:173), then g1 - s1 (LO:161), g2 - s2 (LO:169).
    """

    def __init__(self, P: Params, nt: int = 200, formulation: int = 0, scheme: int = 0):
        """scheme 0 = backward Euler (the reference's NODES=2); scheme 1 = trapezoid with the control held over
        the step (zero-order hold, MV_TYPE=0): z_k - z_{k-1} - dt/2 [f(z_k,u_k) + f(z_{k-1},u_k)] = 0 -- not a
        reference scheme; SURVEY.md Appendix C's independent probe gives t_f = 435.227 s for it at nt=200."""
        if scheme == 1 and formulation != 0:
            raise NotImplementedError("trapezoid is restated for the current formulation only")
        self.scheme = scheme
        self.P, self.nt, self.K, self.form = P, nt, nt - 1, formulation
        self.d = P.derived()
        self.h = 1.0 / (nt - 1)
        if formulation == 0:
            self.ns, self.fields = 7, (X, Y, VX, VY, A, W, MS)
            self.ix, self.iy, self.ivx, self.ivy, self.ia, self.iw, self.im, self.iu = 0, 1, 2, 3, 4, 5, 6, 7
        else:
            self.ns = 5
            self.ix, self.iy, self.ivx, self.ivy, self.im, self.ia = 0, 1, 2, 3, 4, 5
            self.iw = self.iu = None
        self.nw = self.ns + 1
        K, nw = self.K, self.nw
        self.n = nw * K + 3
        self.m = self.ns * K + 3
        self.itf, self.is1, self.is2 = nw * K, nw * K + 1, nw * K + 2
        lb = np.full(self.n, -np.inf)
        ub = np.full(self.n, np.inf)
        base = np.arange(K) * nw
        lb[base + self.im], ub[base + self.im] = 0.0, 1.0               # LO:83
        lb[base + self.ia], ub[base + self.ia] = 0.0, P.angle_ub        # LO:94
        if formulation == 0:
            lb[base + self.iu], ub[base + self.iu] = -1.0, 1.0          # LO:96
        lb[self.itf], ub[self.itf] = P.tf_lb, P.tf_ub                   # LO:39
        lb[self.is1] = lb[self.is2] = 0.0
        self.lb, self.ub = lb, ub

    # -- helpers -------------------------------------------------------------------
    def split(self, v):
        Wk = v[: self.nw * self.K].reshape(self.K, self.nw)
        return Wk, v[self.itf], v[self.is1], v[self.is2]

    def objective(self, v):
        return v[self.itf]

    def grad_objective(self, v):
        g = np.zeros(self.n)
        g[self.itf] = 1.0
        return g

    def _rhs(self, Wk):
        """f(z_k,u_k) of the scaled ODEs without the tf*T factor (LO:114-123) and df/dw."""
        P, d = self.P, self.d
        ax, ay, gax, gay = accel(Wk[:, self.ix], Wk[:, self.iy], Wk[:, self.ia], Wk[:, self.im], P)
        F = np.zeros((self.K, self.ns))
        F[:, self.ix], F[:, self.iy] = Wk[:, self.ivx], Wk[:, self.ivy]
        F[:, self.ivx], F[:, self.ivy] = ax, ay
        F[:, self.im] = d["beta"]
        if self.form == 0:
            F[:, self.ia] = Wk[:, self.iw]
            F[:, self.iw] = d["alpha"] * Wk[:, self.iu]
        return F, gax, gay

    def _prev_nodes(self, Wk):
        """Rows (z_{k-1}, u_k), k = 1..K, with z_0 = 0: where the trapezoid rule evaluates f a second time."""
        Wp = np.vstack([np.zeros((1, self.nw)), Wk[:-1]]).copy()
        Wp[:, self.iu] = Wk[:, self.iu]
        return Wp

    def constraints(self, v):
        Wk, tf, s1, s2 = self.split(v)
        d = self.d
        F, _, _ = self._rhs(Wk)
        Z = Wk[:, : self.ns]
        Zprev = np.vstack([np.zeros((1, self.ns)), Z[:-1]])
        dt = self.h * self.P.T_scale * tf
        c = np.empty(self.m)
        if self.scheme == 0:
            c[: self.ns * self.K] = (Z - Zprev - dt * F).ravel()
        else:
            Fb, _, _ = self._rhs(self._prev_nodes(Wk))           # f(z_{k-1}, u_k)
            c[: self.ns * self.K] = (Z - Zprev - 0.5 * dt * (F + Fb)).ravel()
        xK, yK, vxK, vyK = Wk[-1, self.ix], Wk[-1, self.iy], Wk[-1, self.ivx], Wk[-1, self.ivy]
        eta = yK + d["rho0"]
        c[-3] = eta * vyK + xK * vxK                                  # LO:173 divided by S^2
        c[-2] = math.hypot(xK, eta) - d["rhof"] - s1                  # LO:161
        c[-1] = vxK * vxK + vyK * vyK - d["vp2"] - s2                 # LO:169
        return c

    def jacobian(self, v):
        Wk, tf, s1, s2 = self.split(v)
        K, ns, nw, d, P = self.K, self.ns, self.nw, self.d, self.P
        F, gax, gay = self._rhs(Wk)
        hT = self.h * P.T_scale
        dt = hT * tf
        rows, cols, vals = [], [], []

        def add(r, c_, val):
            rows.append(np.broadcast_to(r, np.shape(val)).ravel() if np.ndim(val) else np.atleast_1d(r))
            cols.append(np.broadcast_to(c_, np.shape(val)).ravel() if np.ndim(val) else np.atleast_1d(c_))
            vals.append(np.atleast_1d(val).ravel())

        k = np.arange(K)
        rb, cb = k * ns, k * nw
        one = np.ones(K)
        trap = self.scheme == 1
        wc = 0.5 if trap else 1.0                # weight of f(z_k, u_k) in the step
        for i in range(ns):                      # d c_k / d z_k (identity) and d c_k / d z_{k-1}
            add(rb + i, cb + i, one)
            add(rb[1:] + i, cb[:-1] + i, -one[1:])
        add(rb + self.ix, cb + self.ivx, -wc * dt * one)
        add(rb + self.iy, cb + self.ivy, -wc * dt * one)
        for j, col in enumerate((self.ix, self.iy, self.ia, self.im)):
            add(rb + self.ivx, cb + col, -wc * dt * gax[:, j])
            add(rb + self.ivy, cb + col, -wc * dt * gay[:, j])
        if self.form == 0:
            add(rb + self.ia, cb + self.iw, -wc * dt * one)
            add(rb + self.iw, cb + self.iu, -dt * d["alpha"] * one)      # u_k enters both halves of the step
        if trap:                                 # d c_k / d z_{k-1} through f(z_{k-1}, u_k), k >= 2
            r1_, c0_ = rb[1:], cb[:-1]
            add(r1_ + self.ix, c0_ + self.ivx, -0.5 * dt * one[1:])
            add(r1_ + self.iy, c0_ + self.ivy, -0.5 * dt * one[1:])
            for j, col in enumerate((self.ix, self.iy, self.ia, self.im)):
                add(r1_ + self.ivx, c0_ + col, -0.5 * dt * gax[:-1, j])
                add(r1_ + self.ivy, c0_ + col, -0.5 * dt * gay[:-1, j])
            add(r1_ + self.ia, c0_ + self.iw, -0.5 * dt * one[1:])
            Fb, _, _ = self._rhs(self._prev_nodes(Wk))
            F = 0.5 * (F + Fb)
        for i in range(ns):                      # tf column
            add(rb + i, np.full(K, self.itf), -hT * F[:, i])
        last = (K - 1) * nw
        xK, yK, vxK, vyK = Wk[-1, self.ix], Wk[-1, self.iy], Wk[-1, self.ivx], Wk[-1, self.ivy]
        eta = yK + d["rho0"]
        rho = math.hypot(xK, eta)
        r3, r1, r2 = ns * K, ns * K + 1, ns * K + 2
        for c_, val in ((self.ix, vxK), (self.iy, vyK), (self.ivx, xK), (self.ivy, eta)):
            add(r3, last + c_, val)
        add(r1, last + self.ix, xK / rho)
        add(r1, last + self.iy, eta / rho)
        add(r1, self.is1, -1.0)
        add(r2, last + self.ivx, 2 * vxK)
        add(r2, last + self.ivy, 2 * vyK)
        add(r2, self.is2, -1.0)
        return sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                             shape=(self.m, self.n))

    def hessian(self, v, lam):
        """Hessian of  tf + lam . c  (full symmetric, csc)."""
        Wk, tf, s1, s2 = self.split(v)
        K, ns, nw, d, P = self.K, self.ns, self.nw, self.d, self.P
        hT = self.h * P.T_scale
        dt = hT * tf
        L = lam[: ns * K].reshape(K, ns)
        Lu = L                                   # multipliers seen by u_k (its own step only)
        if self.scheme == 1:                     # node k enters step k and step k+1, each with weight 1/2
            L = 0.5 * (L + np.vstack([L[1:], np.zeros((1, ns))]))
        lvx, lvy = L[:, self.ivx], L[:, self.ivy]
        ax, ay, gax, gay, H = accel(Wk[:, self.ix], Wk[:, self.iy], Wk[:, self.ia], Wk[:, self.im], P,
                                    -dt * lvx, -dt * lvy)
        rows, cols, vals = [], [], []

        def addsym(r, c_, val):
            r = np.atleast_1d(r); c_ = np.atleast_1d(c_); val = np.atleast_1d(val)
            r, c_, val = np.broadcast_arrays(r, c_, val)
            rows.append(r); cols.append(c_); vals.append(val)
            off = r != c_
            rows.append(c_[off]); cols.append(r[off]); vals.append(val[off])

        cb = np.arange(K) * nw
        q = (self.ix, self.iy, self.ia, self.im)
        idx = 0
        for i in range(4):
            for j in range(i, 4):
                addsym(cb + q[i], cb + q[j], H[:, idx]); idx += 1
        # tf coupling column: d2L/dtf dw_k = -hT * lam_k . dF/dw_k
        tfc = np.full(K, self.itf)
        addsym(tfc, cb + self.ivx, -hT * L[:, self.ix])
        addsym(tfc, cb + self.ivy, -hT * L[:, self.iy])
        for j, col in enumerate(q):
            addsym(tfc, cb + col, -hT * (lvx * gax[:, j] + lvy * gay[:, j]))
        if self.form == 0:
            addsym(tfc, cb + self.iw, -hT * L[:, self.ia])
            addsym(tfc, cb + self.iu, -hT * d["alpha"] * Lu[:, self.iw])
        # terminal constraints
        last = (K - 1) * nw
        nu3, nu1, nu2 = lam[-3], lam[-2], lam[-1]
        xK, yK = Wk[-1, self.ix], Wk[-1, self.iy]
        eta = yK + d["rho0"]
        rho = math.hypot(xK, eta)
        ex, ey = xK / rho, eta / rho
        addsym(last + self.ix, last + self.ix, nu1 * ey * ey / rho)
        addsym(last + self.ix, last + self.iy, -nu1 * ex * ey / rho)
        addsym(last + self.iy, last + self.iy, nu1 * ex * ex / rho)
        addsym(last + self.ivx, last + self.ivx, 2 * nu2)
        addsym(last + self.ivy, last + self.ivy, 2 * nu2)
        addsym(last + self.ix, last + self.ivx, nu3)
        addsym(last + self.iy, last + self.ivy, nu3)
        return sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                             shape=(self.n, self.n))

    # -- initial guess -------------------------------------------------------------
    def initial_guess(self, tf0=0.9, downrange=0.166, angle_end=0.5):
        """Straight-line states toward a tangential orbit-insertion point; u = 0."""
        K, d = self.K, self.d
        v = np.zeros(self.n)
        Wk = v[: self.nw * K].reshape(K, self.nw)
        fr = np.arange(1, K + 1) / K
        vp = math.sqrt(d["vp2"])
        xf, yf = -d["rhof"] * math.sin(downrange), d["rhof"] * math.cos(downrange) - d["rho0"]
        Wk[:, self.ix], Wk[:, self.iy] = fr * xf, fr * yf
        Wk[:, self.ivx], Wk[:, self.ivy] = -fr * vp * math.cos(downrange), -fr * vp * math.sin(downrange)
        Wk[:, self.ia] = fr * angle_end
        dt = self.h * self.P.T_scale * tf0
        Wk[:, self.im] = d["beta"] * dt * np.arange(1, K + 1)
        if self.form == 0:
            Wk[:, self.iw] = angle_end / (K * dt)
        v[self.itf] = tf0
        return v

    # -- reference-style outputs (LO:187-202) --------------------------------------
    def outputs(self, v):
        Wk, tf, _, _ = self.split(v)
        P, d = self.P, self.d
        Z = np.vstack([np.zeros((1, self.nw)), Wk])
        ax, ay, _, _ = accel(Z[:, self.ix], Z[:, self.iy], Z[:, self.ia], Z[:, self.im], P)
        S = d["S"]
        return dict(
            tf=tf, final_time=tf * P.T_scale,
            t=np.linspace(0, 1, self.nt) * tf * P.T_scale,
            x=Z[:, self.ix], y=Z[:, self.iy], xdot=Z[:, self.ivx], ydot=Z[:, self.ivy],
            xdoubledot=ax, ydoubledot=ay, angle=Z[:, self.ia], mass=Z[:, self.im],
            angledot=Z[:, self.iw] if self.form == 0 else None,
            angledoubledot=Z[:, self.iu] if self.form == 0 else None,
            final_y=Z[-1, self.iy] * S, final_x=Z[-1, self.ix] * S,
            final_ydot=Z[-1, self.ivy] * S, final_xdot=Z[-1, self.ivx] * S,
            final_ydoubledot=ay[-1] * S, final_xdoubledot=ax[-1] * S,
        )


# ----------------------------------------------------------------------------------
# primal-dual interior point (restating Waechter & Biegler 2006 with an l1 merit line search)
# ----------------------------------------------------------------------------------
def _factor_with_inertia(Kmat, n, m):
    """Sparse LU without row pivoting on a symmetric ordering (= an LDL' factorisation with a diagonal D): the signs of
    U's diagonal are the inertia (Sylvester).  Returns (lu, ok) with ok = exactly n positive and m negative pivots."""
    Kmat.eliminate_zeros()
    lu = spla.splu(Kmat, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    dU = lu.U.diagonal()
    sym = np.array_equal(lu.perm_r, lu.perm_c)
    return lu, bool(sym and np.all(np.isfinite(dU)) and (dU > 0).sum() == n and (dU < 0).sum() == m)


class MovePenaltyNLP:
    """Any of this module's NLPs plus the MV's DCOST (LO:99; v1 script: `angle.DCOST = 1e-5`, PDF p26) as APMonitor documents it:
    + dcost * sum_k |c_k - c_{k-1}| over the control c (c_0 = c_init), with a slack pair per step -- c_k - c_{k-1} = p_k - n_k,
    p, n >= 0, cost dcost (p_k + n_k) -- as explicit unknowns and rows; nothing is reduced.  Unknowns [base | p | n]; equalities
    [base rows but the last three | movement equations | the base's last three (terminal) rows], so that `solve_ip` finds the two
    terminal inequality rows where it expects them.  The independent anchor of the stage-structured implementations (C
    restatement, HIP kernels) for the formulations oracle/ascent_general.py does not restate (the v1 script)."""

    def __init__(self, base, ctrl_cols, dcost: float, c_init: float = 0.0):
        self.base, self.cc, self.dcost, self.c_init = base, np.asarray(ctrl_cols), float(dcost), float(c_init)
        K = len(self.cc)
        self.Kc = K
        self.n, self.m = base.n + 2 * K, base.m + K
        self.ip, self.in_ = base.n + np.arange(K), base.n + K + np.arange(K)
        self.rmove = base.m - 3 + np.arange(K)
        self.lb = np.concatenate([base.lb, np.zeros(2 * K)])
        self.ub = np.concatenate([base.ub, np.full(2 * K, np.inf)])
        self.itf, self.is1, self.is2 = base.itf, base.is1, base.is2
        self.P, self.nt, self.K = base.P, base.nt, base.K

    def objective(self, v):
        return self.base.objective(v[: self.base.n]) + self.dcost * (v[self.ip].sum() + v[self.in_].sum())

    def grad_objective(self, v):
        g = np.concatenate([self.base.grad_objective(v[: self.base.n]), np.full(2 * self.Kc, self.dcost)])
        return g

    def constraints(self, v):
        cb = self.base.constraints(v[: self.base.n])
        c = v[self.cc]
        mv = c - np.concatenate([[self.c_init], c[:-1]]) - v[self.ip] + v[self.in_]
        return np.concatenate([cb[:-3], mv, cb[-3:]])

    def jacobian(self, v):
        import scipy.sparse as sp_
        Jb = self.base.jacobian(v[: self.base.n]).tocsr()
        K = self.Kc
        pad = sp_.csr_matrix((Jb.shape[0], 2 * K))
        Jb = sp_.hstack([Jb, pad]).tocsr()
        rows = np.concatenate([np.arange(K), np.arange(1, K), np.arange(K), np.arange(K)])
        cols = np.concatenate([self.cc, self.cc[:-1], self.ip, self.in_])
        vals = np.concatenate([np.ones(K), -np.ones(K - 1), -np.ones(K), np.ones(K)])
        Jm = sp_.csr_matrix((vals, (rows, cols)), shape=(K, self.n))
        return sp_.vstack([Jb[:-3], Jm, Jb[-3:]]).tocsc()

    def hessian(self, v, lam):
        import scipy.sparse as sp_
        lb_ = np.concatenate([lam[: self.base.m - 3], lam[-3:]])
        Hb = self.base.hessian(v[: self.base.n], lb_)
        return sp_.block_diag([Hb, sp_.csr_matrix((2 * self.Kc, 2 * self.Kc))]).tocsc()

    def initial_guess(self, **kw):
        return np.concatenate([self.base.initial_guess(**kw), np.full(2 * self.Kc, 0.1)])

    def outputs(self, v):
        return self.base.outputs(v[: self.base.n])

    def split(self, v):
        return self.base.split(v[: self.base.n])


def solve_ip(nlp: AscentNLP, v0=None, tol=1e-9, max_iter=300, mu0=0.1, verbose=False, inertia="curvature"):
    """inertia: "curvature" = accept a step when dx'(W+Sigma+dw)dx > 0 along it (the test this oracle has always used: cheap,
    sufficient on the single-phase problems); "exact" = IPOPT's rule, the KKT matrix must have exactly n positive and m
    negative eigenvalues, read off an LDL' factorisation (needed on the burn-coast problems, whose Newton systems are
    indefinite in directions a single curvature sample does not see)."""
    n, m = nlp.n, nlp.m
    lb, ub = nlp.lb, nlp.ub
    hasL, hasU = np.isfinite(lb), np.isfinite(ub)
    v = nlp.initial_guess() if v0 is None else v0.copy()
    # slacks from the constraint values, then push everything strictly inside the bounds
    v[nlp.is1] = v[nlp.is2] = 0.0
    c0 = nlp.constraints(v)
    v[nlp.is1], v[nlp.is2] = max(c0[-2], 1e-2), max(c0[-1], 1e-4)
    k1 = 1e-2
    both = hasL & hasU
    pl = np.where(hasL, np.minimum(k1 * np.maximum(1, np.abs(np.where(hasL, lb, 0))),
                                   np.where(both, k1 * (ub - lb), np.inf)), 0)
    pu = np.where(hasU, np.minimum(k1 * np.maximum(1, np.abs(np.where(hasU, ub, 0))),
                                   np.where(both, k1 * (ub - lb), np.inf)), 0)
    v = np.where(hasL, np.maximum(v, lb + pl), v)
    v = np.where(hasU, np.minimum(v, ub - pu), v)
    zL = np.where(hasL, 1.0, 0.0)
    zU = np.where(hasU, 1.0, 0.0)
    lam = np.zeros(m)
    mu = mu0
    nu_pen = 1.0
    dw_last = 0.0
    info = dict(iters=0, status="max_iter", reg=0)

    def barrier(vv):
        return (nlp.objective(vv) - mu * np.sum(np.log(vv[hasL] - lb[hasL]))
                - mu * np.sum(np.log(ub[hasU] - vv[hasU])))

    for it in range(max_iter):
        c = nlp.constraints(v)
        J = nlp.jacobian(v)
        gf = nlp.grad_objective(v)
        dL = np.where(hasL, v - lb, 1.0)
        dU = np.where(hasU, ub - v, 1.0)
        rd = gf + J.T @ lam - zL + zU
        sd = max(100.0, (np.abs(lam).sum() + zL.sum() + zU.sum()) / (m + hasL.sum() + hasU.sum())) / 100.0

        def err(mu_):
            comp = max(np.abs(dL * zL - mu_)[hasL].max(), np.abs(dU * zU - mu_)[hasU].max())
            return max(np.abs(rd).max() / sd, np.abs(c).max(), comp / sd)

        e0 = err(0.0)
        if verbose:
            print(f"it {it:3d} tf={v[nlp.itf]:.9f} mu={mu:.1e} E0={e0:.2e} |c|={np.abs(c).max():.2e} "
                  f"|rd|={np.abs(rd).max():.2e} reg={dw_last:.1e}")
        if e0 <= tol:
            info.update(status="converged")
            break
        while err(mu) <= 10.0 * mu and mu > tol / 10:
            mu = max(tol / 10.0, min(0.2 * mu, mu ** 1.5))
            nu_pen = 1.0
        Wm = nlp.hessian(v, lam)
        Sig = np.where(hasL, zL / dL, 0) + np.where(hasU, zU / dU, 0)
        gphi = gf - np.where(hasL, mu / dL, 0) + np.where(hasU, mu / dU, 0)
        rhs = -np.concatenate([gphi + J.T @ lam, c])
        dw = 0.0
        while True:
            Hm = Wm + sp.diags(Sig + dw)
            # (exact inertia: a 1e-10 dual regularisation makes the matrix quasi-definite, so that the unpivoted LDL' exists)
            Kmat = sp.bmat([[Hm, J.T], [J, -1e-10 * sp.eye(m) if inertia == "exact" else None]], format="csc")
            try:
                if inertia == "exact":
                    lu, ok = _factor_with_inertia(Kmat, n, m)
                    sol = lu.solve(rhs)
                    ok = ok and bool(np.all(np.isfinite(sol)))
                    dx, dlam = sol[:n], sol[n:]
                    curv = dx @ (Hm @ dx)
                else:
                    sol = spla.splu(Kmat).solve(rhs)
                    dx, dlam = sol[:n], sol[n:]
                    curv = dx @ (Hm @ dx)
                    ok = np.all(np.isfinite(sol)) and curv >= 1e-11 * (dx @ dx)
            except RuntimeError:
                ok = False
            if ok:
                break
            if inertia == "exact":       # IPOPT's Algorithm IC: restart from a third of the last successful value, floor 1e-20
                dw = (max(1e-20, dw_last / 3.0) if dw_last > 0 else 1e-4) if dw == 0 else dw * (8.0 if dw_last > 0 else 100.0)
            else:
                dw = 1e-4 if dw == 0 else (max(1e-4, dw_last / 3) if dw < 0 else dw * 8)
            info["reg"] += 1
            if dw > 1e10:
                info.update(status="reg_failed")
                return v, lam, info
        if inertia != "exact" or dw > 0:
            dw_last = dw
        dzL = np.where(hasL, mu / dL - zL - zL / dL * dx, 0)
        dzU = np.where(hasU, mu / dU - zU + zU / dU * dx, 0)
        tau = max(0.99, 1 - mu)

        def amax(val, dval, mask):
            neg = mask & (dval < 0)
            return min(1.0, (-tau * val[neg] / dval[neg]).min()) if neg.any() else 1.0

        a_pr = min(amax(dL, dx, hasL), amax(dU, -dx, hasU))
        a_du = min(amax(zL, dzL, hasL), amax(zU, dzU, hasU))
        # l1 merit with penalty update (Nocedal & Wright, eq. 18.36 / 19.2)
        c1 = np.abs(c).sum()
        gd = gphi @ dx
        if c1 > 0:
            need = (gd + 0.5 * max(curv, 0.0)) / (0.9 * c1)
            if nu_pen < need:
                nu_pen = need + 1.0
        D = gd - nu_pen * c1
        phi0 = barrier(v) + nu_pen * c1
        alpha = a_pr
        accepted = False
        for _ in range(40):
            vt = v + alpha * dx
            ct = nlp.constraints(vt)
            phit = barrier(vt) + nu_pen * np.abs(ct).sum()
            if np.isfinite(phit) and phit <= phi0 + 1e-8 * alpha * D + 10 * np.finfo(float).eps * abs(phi0):
                accepted = True
                break
            alpha *= 0.5
        if not accepted:
            info.update(status="linesearch_failed")
            break
        v = vt
        lam = lam + alpha * dlam
        zL = zL + a_du * dzL
        zU = zU + a_du * dzU
        # IPOPT's safeguard keeping z within [mu/(k d), k mu/d], k = 1e10
        dL = np.where(hasL, v - lb, 1.0)
        dU = np.where(hasU, ub - v, 1.0)
        zL = np.where(hasL, np.clip(zL, mu / (1e10 * dL), 1e10 * mu / dL), 0)
        zU = np.where(hasU, np.clip(zU, mu / (1e10 * dU), 1e10 * mu / dU), 0)
        info["iters"] = it + 1
    info["mu"] = mu
    return v, lam, info
