"""TEST INFRASTRUCTURE ONLY -- generalised numpy restatement of the ascent NLP (oracle for the widened rows).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(lunar_module_ascent_trajectory_optimiser_amd/) never does.

What it adds to oracle/ascent_numpy.py (same problem data, same state order, same interior point `solve_ip`):
  * scheme 2, Hermite-Simpson in compressed form with the control held over the step (the reference's
    zero-order-hold MV, MV_TYPE=0, /root/reference/Launch_Optimiser.py:29):
        z_m = (z_{k-1}+z_k)/2 + dt/8 [f(z_{k-1},u_k) - f(z_k,u_k)]
        c_k = z_k - z_{k-1} - dt/6 [f(z_{k-1},u_k) + 4 f(z_m,u_k) + f(z_k,u_k)]
    (Kelly, SIAM Review 59(4) 2017, the method source the reference's report cites, PDF p3/p25);
  * phases: a burn arc (the reference's dynamics, LO:114-136) followed by an optional coast arc (thrust off, mass and
    attitude frozen, two-body gravity), each with its own free duration theta_p * T; with a coast arc the terminal
    constraints are those of the 87 x 17 km ellipse proper (README.md:7): arrival at its apoapsis, r.v = 0,
    radius >= R0 + r_apo, speed^2 >= vis-viva speed^2 at the apoapsis of the (r_peri, r_apo) ellipse -- instead of
    LO:72-78's circular speed of the mean radius -- and the objective is the burn time (the fuel);
  * DCOST, the l1 movement penalty on the MV (LO:99):  + dcost * sum_k |u_k - u_{k-1}|, u_0 = 0, as APMonitor documents
    it, with a slack pair per step (u_k - u_{k-1} = p_k - n_k, p, n >= 0, cost dcost*(p_k + n_k)).

The derivatives are NOT hand-written here: the step defect is written once symbolically (sympy) and its Jacobian and
the Hessian of lambda'c are generated from that expression and compiled with lambdify.  The HIP product derives the same
quantities by hand (chain rule through the three evaluation points); agreement of the two is therefore evidence, not
code equivalence.  Schemes 0/1 of this module are checked against the hand-written oracle/ascent_numpy.py in
tests/test_general_oracle.py.

parity unpinned by the reference for everything that is new here: the reference has neither Hermite-Simpson nor a
coast arc (LO is single-phase, NODES=2), and GEKKO is not installed, so DCOST's effect cannot be compared with a
reference run either; what pins these rows is stated with each test.
"""
from __future__ import annotations

import functools
import math

import numpy as np
import scipy.sparse as sp

from .ascent_numpy import Params, accel

NS, NW = 7, 8
X, Y, VX, VY, A, W, MS = range(7)


# ----------------------------------------------------------------------------------------------------------------
# symbolic step defect and its derivatives
# ----------------------------------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=None)
def _step_functions(scheme: int, coast: bool):
    """Returns (c, J, H) as numpy-callable functions of (za[7], zb[7], u, th, lam[7], consts[8]) -> arrays.
    c: 7 defects; J: 7 x 16 Jacobian w.r.t. (za, zb, u, th); H: 16 x 16 Hessian of lam'c.  Vectorised over steps."""
    import sympy as sy
    za = sy.symbols("za0:7", real=True)
    zb = sy.symbols("zb0:7", real=True)
    lam = sy.symbols("l0:7", real=True)
    u, th = sy.symbols("u th", real=True)
    rho0, gam, thr, M0, ms, alpha, mrate, hT = sy.symbols("rho0 gam thr M0 ms alpha mrate hT", positive=True)

    def f(z):
        x, y, vx, vy, a, w, m = z
        et = y + rho0
        rho = sy.sqrt(x * x + et * et)
        g3 = gam / rho ** 3
        if coast:                       # thrust off; mass and attitude frozen
            return [vx, vy, -g3 * x, -g3 * et, sy.Integer(0), sy.Integer(0), sy.Integer(0)]
        ex, ey = x / rho, et / rho
        c3, s3 = sy.cos(3 * a), sy.sin(3 * a)
        t = thr / (M0 - ms * m)
        return [vx, vy, t * (ex * c3 - ey * s3) - g3 * x, t * (ey * c3 + ex * s3) - g3 * et, w, alpha * u, mrate]

    dt = hT * th
    fa, fb = f(za), f(zb)
    if scheme == 0:
        c = [zb[i] - za[i] - dt * fb[i] for i in range(7)]
    elif scheme == 1:
        c = [zb[i] - za[i] - dt / 2 * (fa[i] + fb[i]) for i in range(7)]
    elif scheme == 2:
        zm = [(za[i] + zb[i]) / 2 + dt / 8 * (fa[i] - fb[i]) for i in range(7)]
        fm = f(zm)
        c = [zb[i] - za[i] - dt / 6 * (fa[i] + 4 * fm[i] + fb[i]) for i in range(7)]
    else:
        raise ValueError("scheme must be 0 (backward Euler), 1 (trapezoid) or 2 (Hermite-Simpson)")
    var = list(za) + list(zb) + [u, th]
    J = [[sy.diff(ci, v) for v in var] for ci in c]
    L = sum(lam[i] * c[i] for i in range(7))
    g = [sy.diff(L, v) for v in var]
    H = [[sy.diff(g[i], var[j]) if j >= i else sy.Integer(0) for j in range(16)] for i in range(16)]
    args = list(za) + list(zb) + [u, th] + list(lam) + [rho0, gam, thr, M0, ms, alpha, mrate, hT]
    flat = c + [e for row in J for e in row] + [H[i][j] for i in range(16) for j in range(i, 16)]
    fn = sy.lambdify(args, flat, modules="numpy", cse=True)

    def evaluate(za_, zb_, u_, th_, lam_, consts):
        n = za_.shape[0]
        out = fn(*za_.T, *zb_.T, u_, th_, *lam_.T, *consts)
        out = [np.broadcast_to(np.asarray(o, float), (n,)) for o in out]
        cc = np.stack(out[:7], 1)
        JJ = np.stack(out[7:7 + 112], 1).reshape(n, 7, 16)
        HH = np.zeros((n, 16, 16))
        iu = np.triu_indices(16)
        HH[:, iu[0], iu[1]] = np.stack(out[119:], 1)
        HH[:, iu[1], iu[0]] = HH[:, iu[0], iu[1]]
        return cc, JJ, HH

    return evaluate


class GeneralNLP:
    """min sum_p w_p theta_p (+ dcost * sum (p+n))  s.t.  step defects, terminal constraints, bounds.

    phases: sequence of (n_steps, "burn" | "coast").  Unknowns v = [per step: z_k (7) and, on burn steps, u_k |
    theta_1..theta_P | s1 s2 | p_1..p_Kb n_1..n_Kb (with dcost)].  Equalities: 7K defects | K_b movement equations (with
    dcost) | e3 | g1 - s1 | g2 - s2.  The interface is the one solve_ip (ascent_numpy.py) expects.
    """

    def __init__(self, P: Params, phases=((199, "burn"),), scheme: int = 0, dcost: float | None = None,
                 coast_ub: float = 12.0, terminal: str | None = None):
        self.P, self.scheme = P, scheme
        self.d = P.derived()
        self.phases = tuple((int(n), str(kind)) for n, kind in phases)
        assert all(kind in ("burn", "coast") for _, kind in self.phases) and self.phases[0][1] == "burn"
        self.K = sum(n for n, _ in self.phases)
        self.nt = self.K + 1
        self.NP = len(self.phases)
        self.dcost = P.dcost if dcost is None else dcost
        self.terminal = terminal or ("apoapsis" if any(kind == "coast" for _, kind in self.phases) else "reference")
        self.phase_of = np.concatenate([np.full(n, p) for p, (n, _) in enumerate(self.phases)])
        self.coast = np.array([self.phases[p][1] == "coast" for p in self.phase_of])
        self.h = np.concatenate([np.full(n, 1.0 / n) for n, _ in self.phases])       # tau-step of each phase's own [0,1] grid
        self.has_u = ~self.coast
        width = np.where(self.has_u, 8, 7)
        self.col = np.concatenate([[0], np.cumsum(width)])[:-1]                       # first column of step k
        nwk = int(width.sum())
        self.ith = nwk + np.arange(self.NP)
        self.is1, self.is2 = nwk + self.NP, nwk + self.NP + 1
        self.itf = int(self.ith[0])
        self.Kb = int(self.has_u.sum())
        self.ucol = (self.col + 7)[self.has_u]
        n = nwk + self.NP + 2
        if self.dcost > 0:
            self.ip = n + np.arange(self.Kb)
            self.in_ = n + self.Kb + np.arange(self.Kb)
            n += 2 * self.Kb
        self.n = n
        # terminal "ellipse": burnout ANYWHERE on the (r_peri, r_apo) ellipse -- two conditions, angular momentum h >= h_t and
        # specific energy E <= E_t of that ellipse (an orbit nested inside the target annulus: periapsis not lower, apoapsis not
        # higher; both active at the optimum, like the reference's own radius and speed conditions), and no r.v = 0: the coast
        # arc that follows is exact two-body motion from whatever true anomaly the burn ends at
        self.nterm = 2 if self.terminal == "ellipse" else 3
        self.m = 7 * self.K + (self.Kb if self.dcost > 0 else 0) + self.nterm
        self.rmove = 7 * self.K + np.arange(self.Kb) if self.dcost > 0 else None
        lb = np.full(n, -np.inf)
        ub = np.full(n, np.inf)
        lb[self.col + MS], ub[self.col + MS] = 0.0, 1.0                   # LO:83
        lb[self.col + A], ub[self.col + A] = 0.0, P.angle_ub              # LO:94
        lb[self.ucol], ub[self.ucol] = -1.0, 1.0                          # LO:96
        lb[self.ith[0]], ub[self.ith[0]] = P.tf_lb, P.tf_ub               # LO:39
        for p in range(1, self.NP):
            lb[self.ith[p]], ub[self.ith[p]] = 0.0, coast_ub
        lb[self.is1] = lb[self.is2] = 0.0
        if self.dcost > 0:
            lb[self.ip] = lb[self.in_] = 0.0
        self.lb, self.ub = lb, ub
        d = self.d
        if self.terminal == "reference":                                  # LO:158-173
            self.rho_t, self.v2_t = d["rhof"], d["vp2"]
        elif self.terminal == "periapsis":                                # insertion at the periapsis of the (r_peri, r_apo)
            ra, rp = P.R0 + P.r_apo, P.R0 + P.r_peri                      # ellipse: LO:158-173 with the vis-viva speed there
            self.rho_t = rp / d["S"]                                      # instead of LO:72-78's circular speed of the mean radius
            self.v2_t = d["GM"] * (2.0 / rp - 2.0 / (ra + rp)) / d["S"] ** 2
        elif self.terminal == "ellipse":                                  # angular momentum and energy of that ellipse, scaled units
            ra, rp = (P.R0 + P.r_apo) / d["S"], (P.R0 + P.r_peri) / d["S"]
            self.h_t = math.sqrt(2.0 * d["gam"] * rp * ra / (rp + ra))
            self.E_t = -d["gam"] / (rp + ra)
            self.rho_t, self.v2_t = rp, d["gam"] * (2.0 / rp - 2.0 / (ra + rp))      # (initial guess only)
        else:                                                             # apoapsis of the (r_peri, r_apo) ellipse
            ra, rp = P.R0 + P.r_apo, P.R0 + P.r_peri
            self.rho_t = ra / d["S"]
            self.v2_t = d["GM"] * (2.0 / ra - 2.0 / (ra + rp)) / d["S"] ** 2
        self.consts = [d["rho0"], d["gam"], d["thr"], P.M0, P.mass_scalar, d["alpha"], d["beta"], 0.0]

    # -- helpers ---------------------------------------------------------------------------------------------
    def states(self, v):
        return v[self.col[:, None] + np.arange(7)]

    def controls(self, v):
        u = np.zeros(self.K)
        u[self.has_u] = v[self.ucol]
        return u

    def split(self, v):
        """(K,8) states+control, theta_1 (interface of ascent_numpy.AscentNLP.split for single-phase callers)"""
        return np.hstack([self.states(v), self.controls(v)[:, None]]), v[self.ith[0]], v[self.is1], v[self.is2]

    def objective(self, v):
        f = v[self.ith[0]]
        if self.dcost > 0:
            f = f + self.dcost * (v[self.ip].sum() + v[self.in_].sum())
        return f

    def grad_objective(self, v):
        g = np.zeros(self.n)
        g[self.ith[0]] = 1.0
        if self.dcost > 0:
            g[self.ip] = g[self.in_] = self.dcost
        return g

    def _steps(self, v, lam=None):
        """Evaluates every step: returns c (K,7), J (K,7,16), H (K,16,16), grouped by (coast?) because the two arcs have
        different symbolic defects."""
        Z = self.states(v)
        Za = np.vstack([np.zeros((1, 7)), Z[:-1]])
        U = self.controls(v)
        TH = v[self.ith][self.phase_of]
        Lm = np.zeros((self.K, 7)) if lam is None else lam[: 7 * self.K].reshape(self.K, 7)
        c = np.zeros((self.K, 7)); J = np.zeros((self.K, 7, 16)); H = np.zeros((self.K, 16, 16))
        for coast in (False, True):
            sel = self.coast == coast
            if not sel.any():
                continue
            consts = list(self.consts)
            consts[7] = self.h[sel] * self.P.T_scale
            cc, JJ, HH = _step_functions(self.scheme, coast)(Za[sel], Z[sel], U[sel], TH[sel], Lm[sel], consts)
            c[sel], J[sel], H[sel] = cc, JJ, HH
        return c, J, H

    def _step_columns(self):
        """(K,16) global column of every local variable (za, zb, u, th); -1 where there is none (za of step 0, u on a
        coast step)."""
        cols = np.full((self.K, 16), -1)
        cols[1:, 0:7] = self.col[:-1, None] + np.arange(7)
        cols[:, 7:14] = self.col[:, None] + np.arange(7)
        cols[self.has_u, 14] = self.ucol
        cols[:, 15] = self.ith[self.phase_of]
        return cols

    def terminal_values(self, v):
        zK = self.states(v)[-1]
        eta = zK[Y] + self.d["rho0"]
        if self.terminal == "ellipse":       # (e3 does not exist): h = X vy - Y vx,  E = |v|^2/2 - gam/rho
            return (0.0, zK[X] * zK[VY] - eta * zK[VX] - self.h_t,
                    self.E_t - (0.5 * (zK[VX] ** 2 + zK[VY] ** 2) - self.d["gam"] / math.hypot(zK[X], eta)))
        return (eta * zK[VY] + zK[X] * zK[VX], math.hypot(zK[X], eta) - self.rho_t,
                zK[VX] ** 2 + zK[VY] ** 2 - self.v2_t)

    def constraints(self, v):
        c, _, _ = self._steps(v)
        out = np.empty(self.m)
        out[: 7 * self.K] = c.ravel()
        if self.dcost > 0:
            U = v[self.ucol]
            out[self.rmove] = U - np.concatenate([[0.0], U[:-1]]) - v[self.ip] + v[self.in_]
        e3, g1, g2 = self.terminal_values(v)
        if self.nterm == 3:
            out[-3] = e3
        out[-2], out[-1] = g1 - v[self.is1], g2 - v[self.is2]
        return out

    def jacobian(self, v):
        _, J, _ = self._steps(v)
        cols = self._step_columns()
        rows = (7 * np.arange(self.K)[:, None, None] + np.arange(7)[None, :, None]) + np.zeros((1, 1, 16), int)
        colsb = np.broadcast_to(cols[:, None, :], J.shape)
        ok = (colsb >= 0) & (J != 0.0)
        r, c_, val = [rows[ok]], [colsb[ok]], [J[ok]]
        if self.dcost > 0:
            kb = np.arange(self.Kb)
            r += [self.rmove, self.rmove[1:], self.rmove, self.rmove]
            c_ += [self.ucol, self.ucol[:-1], self.ip, self.in_]
            val += [np.ones(self.Kb), -np.ones(self.Kb - 1), -np.ones(self.Kb), np.ones(self.Kb)]
            del kb
        last = self.col[-1]
        zK = self.states(v)[-1]
        eta = zK[Y] + self.d["rho0"]
        rho = math.hypot(zK[X], eta)
        r3, r1, r2 = self.m - 3, self.m - 2, self.m - 1
        if self.terminal == "ellipse":
            g3 = self.d["gam"] / rho ** 3
            r += [np.array([r1] * 5 + [r2] * 5)]
            c_ += [np.array([last + X, last + Y, last + VX, last + VY, self.is1, last + X, last + Y, last + VX, last + VY, self.is2])]
            val += [np.array([zK[VY], -zK[VX], -eta, zK[X], -1.0, -g3 * zK[X], -g3 * eta, -zK[VX], -zK[VY], -1.0])]
        else:
            r += [np.array([r3, r3, r3, r3, r1, r1, r1, r2, r2, r2])]
            c_ += [np.array([last + X, last + Y, last + VX, last + VY, last + X, last + Y, self.is1, last + VX, last + VY, self.is2])]
            val += [np.array([zK[VX], zK[VY], zK[X], eta, zK[X] / rho, eta / rho, -1.0, 2 * zK[VX], 2 * zK[VY], -1.0])]
        return sp.csc_matrix((np.concatenate(val), (np.concatenate(r), np.concatenate(c_))), shape=(self.m, self.n))

    def hessian(self, v, lam):
        _, _, H = self._steps(v, lam)
        cols = self._step_columns()
        ri = np.broadcast_to(cols[:, :, None], H.shape)
        ci = np.broadcast_to(cols[:, None, :], H.shape)
        ok = (ri >= 0) & (ci >= 0) & (H != 0.0)
        r, c_, val = [ri[ok]], [ci[ok]], [H[ok]]
        last = self.col[-1]
        nu3, nu1, nu2 = (lam[-3] if self.nterm == 3 else 0.0), lam[-2], lam[-1]
        zK = self.states(v)[-1]
        eta = zK[Y] + self.d["rho0"]
        rho = math.hypot(zK[X], eta)
        ex, ey = zK[X] / rho, eta / rho
        if self.terminal == "ellipse":      # nu1 * Hessian of h (the antisymmetric position-velocity pattern) + nu2 * Hessian of E
            g3 = self.d["gam"] / rho ** 3
            tr = [(X, VY, nu1), (VY, X, nu1), (Y, VX, -nu1), (VX, Y, -nu1),
                  (X, X, -nu2 * g3 * (1 - 3 * ex * ex)), (X, Y, 3 * nu2 * g3 * ex * ey), (Y, X, 3 * nu2 * g3 * ex * ey),
                  (Y, Y, -nu2 * g3 * (1 - 3 * ey * ey)), (VX, VX, -nu2), (VY, VY, -nu2)]
        else:
            tr = [(X, X, nu1 * ey * ey / rho), (X, Y, -nu1 * ex * ey / rho), (Y, X, -nu1 * ex * ey / rho), (Y, Y, nu1 * ex * ex / rho),
                  (VX, VX, 2 * nu2), (VY, VY, 2 * nu2), (X, VX, nu3), (VX, X, nu3), (Y, VY, nu3), (VY, Y, nu3)]
        r.append(np.array([last + a for a, _, _ in tr])); c_.append(np.array([last + b for _, b, _ in tr]))
        val.append(np.array([w for _, _, w in tr]))
        return sp.csc_matrix((np.concatenate(val), (np.concatenate(r), np.concatenate(c_))), shape=(self.n, self.n))

    # -- initial guess ---------------------------------------------------------------------------------------------
    def initial_guess(self, tf0=0.9, downrange=0.166, angle_end=0.5):
        """Burn arc: the straight-line guess of ascent_numpy.AscentNLP.initial_guess (towards the periapsis of the target
        ellipse when a coast arc follows).  Coast arc: the Kepler arc of the target ellipse from its periapsis to its
        apoapsis, sampled uniformly in time (Kepler's equation by Newton), duration half a period."""
        d, P = self.d, self.P
        v = np.zeros(self.n)
        n1 = self.phases[0][0]
        fr = np.arange(1, n1 + 1) / n1
        two = self.NP > 1
        rp, ra = d["rhof"], self.rho_t
        sma, ecc = 0.5 * (rp + ra), (ra - rp) / (ra + rp)
        vp = math.sqrt(d["gam"] * (2.0 / rp - 1.0 / sma)) if two else math.sqrt(d["vp2"])
        xf, yf = -d["rhof"] * math.sin(downrange), d["rhof"] * math.cos(downrange) - d["rho0"]
        Z = np.zeros((self.K, 7))
        Z[:n1, X], Z[:n1, Y] = fr * xf, fr * yf
        Z[:n1, VX], Z[:n1, VY] = -fr * vp * math.cos(downrange), -fr * vp * math.sin(downrange)
        Z[:n1, A] = fr * angle_end
        dt = P.T_scale * tf0 / n1
        Z[:n1, MS] = d["beta"] * dt * np.arange(1, n1 + 1)
        Z[:n1, W] = angle_end / (n1 * dt)
        v[self.ith[0]] = tf0
        k0 = n1
        for p in range(1, self.NP):
            n2 = self.phases[p][0]
            Mean = math.pi * np.arange(1, n2 + 1) / n2                 # mean anomaly 0 .. pi
            E = Mean.copy()
            for _ in range(30):
                E -= (E - ecc * np.sin(E) - Mean) / (1.0 - ecc * np.cos(E))
            nu = 2.0 * np.arctan2(np.sqrt(1 + ecc) * np.sin(E / 2), np.sqrt(1 - ecc) * np.cos(E / 2))
            plat = sma * (1 - ecc * ecc)
            rr = plat / (1 + ecc * np.cos(nu))
            vr, vt = math.sqrt(d["gam"] / plat) * ecc * np.sin(nu), math.sqrt(d["gam"] / plat) * (1 + ecc * np.cos(nu))
            ph = downrange + nu
            Z[k0:k0 + n2, X], Z[k0:k0 + n2, Y] = -rr * np.sin(ph), rr * np.cos(ph) - d["rho0"]
            Z[k0:k0 + n2, VX] = -vr * np.sin(ph) - vt * np.cos(ph)
            Z[k0:k0 + n2, VY] = vr * np.cos(ph) - vt * np.sin(ph)
            Z[k0:k0 + n2, A], Z[k0:k0 + n2, W], Z[k0:k0 + n2, MS] = Z[n1 - 1, A], Z[n1 - 1, W], Z[n1 - 1, MS]
            v[self.ith[p]] = math.pi * math.sqrt(sma ** 3 / d["gam"]) / P.T_scale
            k0 += n2
        v[self.col[:, None] + np.arange(7)] = Z
        if self.dcost > 0:
            v[self.ip] = v[self.in_] = 0.1
        return v

    # -- reference-style outputs (LO:187-202) --------------------------------------------------------------------
    def outputs(self, v):
        P, d = self.P, self.d
        Z = np.vstack([np.zeros((1, 7)), self.states(v)])
        U = np.concatenate([[0.0], self.controls(v)])
        ax, ay, _, _ = accel(Z[:, X], Z[:, Y], Z[:, A], Z[:, MS], P)
        S = d["S"]
        th = v[self.ith]
        t = np.concatenate([[0.0], np.cumsum(self.h * P.T_scale * th[self.phase_of])])
        return dict(tf=th[0], theta=th.copy(), final_time=th[0] * P.T_scale, t=t,
                    x=Z[:, X], y=Z[:, Y], xdot=Z[:, VX], ydot=Z[:, VY], xdoubledot=ax, ydoubledot=ay,
                    angle=Z[:, A], angledot=Z[:, W], angledoubledot=U, mass=Z[:, MS],
                    final_y=Z[-1, Y] * S, final_x=Z[-1, X] * S, final_ydot=Z[-1, VY] * S, final_xdot=Z[-1, VX] * S,
                    final_ydoubledot=ay[-1] * S, final_xdoubledot=ax[-1] * S)


def kepler_elements(P: Params, x, y, vx, vy):
    """Two-body orbit through a scaled state: periapsis / apoapsis altitude above R0 (m)."""
    d = P.derived()
    S, GM = d["S"], d["GM"]
    Xp, Yp, VXp, VYp = x * S, y * S + P.R0, vx * S, vy * S
    r, v2 = math.hypot(Xp, Yp), VXp * VXp + VYp * VYp
    a = 1.0 / (2.0 / r - v2 / GM)
    hh = Xp * VYp - Yp * VXp
    e = math.sqrt(max(0.0, 1.0 - hh * hh / (GM * a)))
    return a * (1 - e) - P.R0, a * (1 + e) - P.R0
