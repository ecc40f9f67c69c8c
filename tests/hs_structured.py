"""TEST INFRASTRUCTURE -- numpy restatement of the STRUCTURED Hermite-Simpson Newton step of csrc/ascent_hs.hip.

The persistent Hermite-Simpson kernel (h_solve) does not work on dense 7x7 step blocks: it applies the step Jacobians
    Ja = -I - (h/6) Fa - (2h/3) Fm Ma,   Jb = I - (h/6) Fb - (2h/3) Fm Mb,   Ma = I/2 + (h/8) Fa,  Mb = I/2 - (h/8) Fb
as operators built from the sparse F = df/dz of the three evaluation points (a solve with Jb' is two 2x2 inverses and a
back substitution), and carries the cross Hessian of a step as a rank-4 term through the midpoint's (x, y, angle, mass).
This module restates that algorithm operation by operation (same order as the kernel's lanes, plain loops) so that the
derivation is checked on the CPU against the generic sparse-LU Newton step of the generalised oracle
(tests/conftest.py::generic_lu_newton_step, sympy-generated derivatives): tests/test_hs_structured.py.
"""
from __future__ import annotations

import numpy as np

from oracle.ascent_numpy import Params, accel

IX, IY, IVX, IVY, IA, IW, IM = range(7)
Q4 = (IX, IY, IA, IM)            # the states the accelerations depend on


def h4(H10):
    H = np.zeros((4, 4))
    iu = np.triu_indices(4)
    H[iu] = H10
    return H + np.triu(H, 1).T


def f_rhs(P, d, z, u, ax, ay):
    return np.array([z[IVX], z[IVY], ax, ay, z[IW], d["alpha"] * u, d["beta"]])


def F_dense(G):
    """df/dz from the 2x4 block G = d(ax, ay)/d(x, y, angle, mass)"""
    F = np.zeros((7, 7))
    F[IX, IVX] = 1.0; F[IY, IVY] = 1.0; F[IA, IW] = 1.0
    for j, q in enumerate(Q4):
        F[IVX, q] = G[j]; F[IVY, q] = G[4 + j]
    return F


def evalpt(P, z, px=None, py=None):
    if px is None:
        ax, ay, gx, gy = accel(z[IX], z[IY], z[IA], z[IM], P)
        return float(ax), float(ay), np.concatenate([gx, gy]), None
    ax, ay, gx, gy, H = accel(z[IX], z[IY], z[IA], z[IM], P, px, py)
    return float(ax), float(ay), np.concatenate([gx, gy]), np.asarray(H, float)


class StepData:
    """Everything node-local of step k (a = node k-1, b = node k): the kernel's node-parallel evaluation."""

    def __init__(self, P, d, hT, th, za, zb, u, lam):
        h = hT * th
        self.h, self.hT = h, hT
        e8, h8 = h / 8.0, hT / 8.0
        sa = sb = h / 6.0
        sm = 4.0 * h / 6.0
        axa, aya, Ga, _ = evalpt(P, za)
        axb, ayb, Gb, _ = evalpt(P, zb)
        fa, fb = f_rhs(P, d, za, u, axa, aya), f_rhs(P, d, zb, u, axb, ayb)
        zm = 0.5 * (za + zb) + e8 * (fa - fb)
        axm, aym, Gm, Hm10 = evalpt(P, zm, lam[IVX], lam[IVY])
        fm = f_rhs(P, d, zm, u, axm, aym)
        Fa, Fb, Fm = F_dense(Ga), F_dense(Gb), F_dense(Gm)
        self.Ga, self.Gb, self.Gm = Ga, Gb, Gm
        self.Fa, self.Fb, self.Fm = Fa, Fb, Fm
        I = np.eye(7)
        Ma, Mb = 0.5 * I + e8 * Fa, 0.5 * I - e8 * Fb
        mth = h8 * (fa - fb)
        wsum = (fa + 4.0 * fm + fb) / 6.0
        self.c = zb - za - h * wsum
        self.Ja = -I - sa * Fa - sm * Fm @ Ma
        self.Jb = I - sb * Fb - sm * Fm @ Mb
        self.bu = h * d["alpha"]                       # Ju = -bu e_w
        self.Jth = -hT * wsum - sm * Fm @ mth
        gm = Fm.T @ lam
        self.gm = gm
        self.W = -sm * h4(Hm10)                        # midpoint curvature on (x, y, angle, mass)
        self.La, self.Lb = Ma[list(Q4)], Mb[list(Q4)]  # d xi / d za, d xi / d zb   (4 x 7)
        self.wa = sa * lam[[IVX, IVY]] + sm * e8 * gm[[IVX, IVY]]     # weights of the end-point Hessians
        self.wb = sb * lam[[IVX, IVY]] - sm * e8 * gm[[IVX, IVY]]
        Mag, Mbg = Ma.T @ gm, Mb.T @ gm
        self.ga = -lam - sa * Fa.T @ lam - sm * Mag     # Ja' lam
        self.gb = lam - sb * Fb.T @ lam - sm * Mbg      # Jb' lam
        Hmm = h4(Hm10) @ mth[list(Q4)]
        self.Hath = -hT * (Fa.T @ lam / 6.0 + (4.0 / 6.0) * Mag) - sm * (h8 * Fa.T @ gm + self.La.T @ Hmm)
        self.Hbth = -hT * (Fb.T @ lam / 6.0 + (4.0 / 6.0) * Mbg) - sm * (-h8 * Fb.T @ gm + self.Lb.T @ Hmm)
        self.Hthth = -2.0 * hT * (4.0 / 6.0) * gm @ mth - sm * mth[list(Q4)] @ Hmm
        self.Huth = -hT * d["alpha"] * lam[IW]
        # the two 2x2 inverses of the structured solve with Jb (B11 = I + eps A_b; S = B22 + (h/2) B21 B11^-1)
        eps = h * h / 12.0
        Ab = np.array([[Gb[0], Gb[1]], [Gb[4], Gb[5]]]); Am = np.array([[Gm[0], Gm[1]], [Gm[4], Gm[5]]])
        self.eps = eps
        self.B11 = np.eye(2) + eps * Ab
        self.B21 = -(h / 6.0) * Ab - (h / 3.0) * Am
        self.B22 = np.eye(2) + eps * Am
        self.Eb = np.linalg.inv(self.B11)
        self.Es = np.linalg.inv(self.B22 + 0.5 * h * self.B21 @ self.Eb)

    # ---- structured operators (what the lanes do) -------------------------------------------------------------------
    def solve_JbT(self, a):
        """x with Jb' x = a.  Jb = [[B, C], [0, T]] over (x y xdot ydot | angle angledot mass): B 4x4 in 2x2 blocks with
        B12 = -(h/2) I, T unit upper triangular with the single entry T[angle][angledot] = -h/2."""
        h, eps, Gb, Gm = self.h, self.eps, self.Gb, self.Gm
        a1, a2 = a[[IX, IY]], a[[IVX, IVY]]
        t1 = self.Eb.T @ a1
        x2 = self.Es.T @ (a2 + 0.5 * h * t1)
        x1 = t1 - self.Eb.T @ (self.B21.T @ x2)
        x = np.zeros(7)
        x[[IX, IY]], x[[IVX, IVY]] = x1, x2
        # columns angle, angledot, mass of Jb on the rows q = (x, y) and v = (xdot, ydot)
        ca_q = eps * np.array([Gb[2], Gb[6]]); ca_v = -(h / 6.0) * np.array([Gb[2], Gb[6]]) - (h / 3.0) * np.array([Gm[2], Gm[6]])
        cw_v = eps * np.array([Gm[2], Gm[6]])
        cm_q = eps * np.array([Gb[3], Gb[7]]); cm_v = -(h / 6.0) * np.array([Gb[3], Gb[7]]) - (h / 3.0) * np.array([Gm[3], Gm[7]])
        ra = a[IA] - ca_q @ x1 - ca_v @ x2
        rw = a[IW] - cw_v @ x2
        rm = a[IM] - cm_q @ x1 - cm_v @ x2
        x[IA] = ra; x[IW] = rw + 0.5 * h * ra; x[IM] = rm
        return x

    def apply_J8T(self, x):
        """-[Ja Ju]' x  (8-vector: the states of node k-1 and the control of the step), from the block form of Ja:
        rows q = (x, y): [-I - eps A_a, -(h/2) I, -eps a_angle^a, 0, -eps a_mass^a];  rows v = (xdot, ydot): [-(h/6) A_a - (h/3) A_m,
        -I - eps A_m, -(h/6) a_angle^a - (h/3) a_angle^m, -eps a_angle^m, -(h/6) a_mass^a - (h/3) a_mass^m];  row angle: -1, -h/2 (angledot);
        rows angledot and mass: -1."""
        h, eps, Ga, Gm = self.h, self.eps, self.Ga, self.Gm
        Aa = np.array([[Ga[0], Ga[1]], [Ga[4], Ga[5]]]); Am = np.array([[Gm[0], Gm[1]], [Gm[4], Gm[5]]])
        b21a = -(h / 6.0) * Aa - (h / 3.0) * Am
        eaa = eps * np.array([Ga[2], Ga[6]]); ema = eps * np.array([Ga[3], Ga[7]])
        cava = -(h / 6.0) * np.array([Ga[2], Ga[6]]) - (h / 3.0) * np.array([Gm[2], Gm[6]])
        cmva = -(h / 6.0) * np.array([Ga[3], Ga[7]]) - (h / 3.0) * np.array([Gm[3], Gm[7]])
        cwv = eps * np.array([Gm[2], Gm[6]])
        xq, xv = x[[IX, IY]], x[[IVX, IVY]]
        out = np.zeros(8)
        out[[IX, IY]] = xq + eps * Aa.T @ xq - b21a.T @ xv
        out[[IVX, IVY]] = 0.5 * h * xq + xv + eps * Am.T @ xv
        out[IA] = eaa @ xq - cava @ xv + x[IA]
        out[IW] = cwv @ xv + 0.5 * h * x[IA] + x[IW]
        out[IM] = ema @ xq - cmva @ xv + x[IM]
        out[7] = self.bu * x[IW]
        assert np.allclose(out[:7], -(self.Ja.T @ x), rtol=1e-12, atol=1e-14 * (1.0 + np.abs(x).max()))
        return out


class HSProblem:
    def __init__(self, P: Params, nt, blob):
        self.P, self.nt = P, nt
        K = self.K = nt - 1
        self.d = d = P.derived()
        self.hT = P.T_scale / K
        self.Z = blob[:7 * K].reshape(K, 7).copy()
        self.U = blob[7 * K:8 * K].copy()
        self.L = blob[8 * K:15 * K].reshape(K, 7).copy()
        self.ZB = blob[15 * K:21 * K].reshape(K, 6).copy()
        self.sc = blob[21 * K:].copy()      # th zlt zut s1 s2 zs1 zs2 nu3 nu1 nu2
        th = self.sc[0]
        self.steps = []
        for k in range(K):
            za = self.Z[k - 1] if k else np.zeros(7)
            self.steps.append(StepData(P, d, self.hT, th, za, self.Z[k], self.U[k], self.L[k]))

    def terminal(self):
        P, d = self.P, self.d
        z = self.Z[-1]
        eta = z[IY] + d["rho0"]
        rho = np.hypot(z[IX], eta)
        ex, ey = z[IX] / rho, eta / rho
        e3 = eta * z[IVY] + z[IX] * z[IVX]
        g1 = rho - d["rhof"]; g2 = z[IVX] ** 2 + z[IVY] ** 2 - d["vp2"]
        e3g = np.zeros(7); e3g[[IX, IY, IVX, IVY]] = [z[IVX], z[IVY], z[IX], eta]
        g1g = np.zeros(7); g1g[[IX, IY]] = [ex, ey]
        g2g = np.zeros(7); g2g[[IVX, IVY]] = [2 * z[IVX], 2 * z[IVY]]
        H1 = np.zeros((7, 7)); H1[np.ix_([IX, IY], [IX, IY])] = np.array([[ey * ey, -ex * ey], [-ex * ey, ex * ex]]) / rho
        H2 = np.zeros((7, 7)); H2[IVX, IVX] = H2[IVY, IVY] = 2.0
        H3 = np.zeros((7, 7)); H3[IX, IVX] = H3[IVX, IX] = H3[IY, IVY] = H3[IVY, IY] = 1.0
        return e3, g1, g2, e3g, g1g, g2g, H1, H2, H3


def newton_step(P: Params, nt, blob, mu, dw):
    """The Newton step of the barrier problem in blob layout, and the inertia flag (0 = correct)."""
    pr = HSProblem(P, nt, blob)
    K, d, st = pr.K, pr.d, pr.steps
    th, zlt, zut, s1, s2, zs1, zs2, nu3, nu1, nu2 = pr.sc
    aub = P.angle_ub
    # ---- node quantities ------------------------------------------------------------------------------------------
    Q = np.zeros((K, 7, 7)); rz = np.zeros((K, 7)); gth = np.zeros((K, 7))
    R0 = np.zeros(K); ru = np.zeros(K)
    idn = np.zeros((K, 6))
    for k in range(K):
        z, u, zb = pr.Z[k], pr.U[k], pr.ZB[k]
        dist = np.array([z[IA], aub - z[IA], z[IM], 1.0 - z[IM], u + 1.0, 1.0 - u])
        idn[k] = 1.0 / dist
        w2 = st[k].wb + (st[k + 1].wa if k + 1 < K else 0.0)
        _, _, _, H10 = evalpt(P, z, w2[0], w2[1])
        Q[k][np.ix_(Q4, Q4)] -= h4(H10)
        Q[k][IA, IA] += zb[0] * idn[k, 0] + zb[1] * idn[k, 1]
        Q[k][IM, IM] += zb[2] * idn[k, 2] + zb[3] * idn[k, 3]
        Q[k] += dw * np.eye(7)
        rz[k] = st[k].gb + (st[k + 1].ga if k + 1 < K else 0.0)
        rz[k][IA] += mu * (idn[k, 1] - idn[k, 0])
        rz[k][IM] += mu * (idn[k, 3] - idn[k, 2])
        gth[k] = st[k].Hbth + (st[k + 1].Hath if k + 1 < K else 0.0)
        R0[k] = zb[4] * idn[k, 4] + zb[5] * idn[k, 5] + dw
        ru[k] = -st[k].bu * pr.L[k][IW] + mu * (idn[k, 5] - idn[k, 4])
    e3, g1, g2, e3g, g1g, g2g, H1, H2, H3 = pr.terminal()
    sig1, sig2 = zs1 / s1 + dw, zs2 / s2 + dw
    rs1, rs2 = -mu / s1 - nu1, -mu / s2 - nu2
    cg1, cg2 = g1 - s1, g2 - s2
    w1, w2_ = nu1 + sig1 * cg1 + rs1, nu2 + sig2 * cg2 + rs2
    Q[K - 1] += nu3 * H3 + nu1 * H1 + nu2 * H2 + sig1 * np.outer(g1g, g1g) + sig2 * np.outer(g2g, g2g)
    # (rz of the last node: the stationarity residual with the terminal multipliers; the eliminated slacks add sig (g - s) + rs)
    rz[K - 1] += nu3 * e3g + w1 * g1g + w2_ * g2g
    # ---- backward sweep: value function 1/2 dz'P dz + p_alpha'dz for the right-hand sides alpha = residual, theta, nu3 -----------
    Pm = np.zeros((7, 7)); p = np.zeros((7, 3))
    v = np.zeros((3, 3))                      # bilinear constants r_alpha . x_beta
    kap = np.zeros((K, 7)); k0 = np.zeros((K, 3)); D = np.zeros(K)
    bad = False
    for k in range(K - 1, -1, -1):
        s = st[k]
        Pm = Pm + Q[k]
        p = p + np.stack([rz[k], gth[k], e3g if k == K - 1 else np.zeros(7)], 1)
        rc = np.stack([s.c, s.Jth, np.zeros(7)], 1)
        # (ii) all lanes: Jb^-T; the columns a second time after the transpose; four spare lanes carry the rows of Lb
        B1 = np.stack([s.solve_JbT(Pm[:, j]) for j in range(7)], 1)             # Jb^-T P
        N = np.stack([s.solve_JbT(B1[j, :]) for j in range(7)], 1)              # Jb^-T P Jb^-1 (columns; symmetric)
        n = np.stack([s.solve_JbT(p[:, a]) for a in range(3)], 1)
        X = np.stack([s.solve_JbT(s.Lb[r]) for r in range(4)], 1)               # (Lb Jb^-1)' : 7 x 4
        # (iii) shift by the defect: n' = n - N rc; constants
        n1 = n - N @ rc
        for a in range(3):
            for b in range(3):
                v[a, b] += -0.5 * ((n[:, a] + n1[:, a]) @ rc[:, b] + (n[:, b] + n1[:, b]) @ rc[:, a])
        xi0 = -X.T @ rc                                                           # 4 x 3
        # (iv) pull back through -[Ja Ju]: columns (twice, transposed in between), right-hand sides and spare lanes once
        M1 = np.stack([s.apply_J8T(N[:, j]) for j in range(7)], 1)              # 8 x 7
        T8 = np.stack([s.apply_J8T(M1[i, :]) for i in range(8)], 1)             # 8 x 8
        t8 = np.stack([s.apply_J8T(n1[:, a]) for a in range(3)], 1)             # 8 x 3
        Lam = np.stack([s.apply_J8T(X[:, r]) for r in range(4)], 0)             # 4 x 8: -Lb Jb^-1 [Ja Ju]
        Lam[:, :7] += s.La
        # (v) the midpoint's curvature: rank 4
        T8 += Lam.T @ s.W @ Lam
        t8 += Lam.T @ s.W @ xi0
        v += xi0.T @ s.W @ xi0
        # (vi) the control: pivot, gains
        T8[7, 7] += R0[k]
        t8[7] += np.array([ru[k], s.Huth, 0.0])
        D[k] = T8[7, 7]
        if not D[k] > 0.0:
            bad = True
        kap[k] = T8[:7, 7] / D[k]
        k0[k] = t8[7] / D[k]
        v -= D[k] * np.outer(k0[k], k0[k])
        Pm = T8[:7, :7] - D[k] * np.outer(kap[k], kap[k])
        p = t8[:7] - np.outer(kap[k], t8[7])
    # ---- border -----------------------------------------------------------------------------------------------------
    itl, itu = 1.0 / (th - P.tf_lb), 1.0 / (P.tf_ub - th)
    rth = 1.0 + sum(s.Jth @ pr.L[k] for k, s in enumerate(st))
    rthp = rth + mu * (itu - itl)
    sth = zlt * itl + zut * itu + dw + sum(s.Hthth for s in st)
    a11, a12, a22 = sth + v[1, 1], v[1, 2], v[2, 2]
    b1, b2 = -rthp - v[1, 0], -e3 - v[2, 0]
    det = a11 * a22 - a12 * a12
    if bad or not det < 0.0:
        return None, 1
    dth = (b1 * a22 - a12 * b2) / det
    dnu3 = (a11 * b2 - a12 * b1) / det
    beta = np.array([1.0, dth, dnu3])
    # ---- forward ------------------------------------------------------------------------------------------------------
    dz = np.zeros((K, 7)); du = np.zeros(K)
    zprev = np.zeros(7)
    for k in range(K):
        s = st[k]
        du[k] = -(k0[k] @ beta) - kap[k] @ zprev
        rhs = -(s.Ja @ zprev) + s.bu * du[k] * np.eye(7)[IW] - s.c - s.Jth * dth
        dz[k] = np.linalg.solve(s.Jb, rhs)
        zprev = dz[k]
    # ---- adjoint: psi_{k-1} = -Ja_k' Jb_k^-T (psi_k - rhs_k),  Jb_k' dlam_k = psi_k - rhs_k -------------------------------------
    dl = np.zeros((K, 7))
    psi = np.zeros(7)
    om_next = np.zeros(4); La_next = np.zeros((4, 7))
    for k in range(K - 1, -1, -1):
        s = st[k]
        za_d = dz[k - 1] if k else np.zeros(7)
        om = s.W @ (s.La @ za_d + s.Lb @ dz[k])
        rhs = rz[k] + Q[k] @ dz[k] + s.Lb.T @ om + La_next.T @ om_next + gth[k] * dth
        if k == K - 1:
            rhs = rhs + e3g * dnu3
        phi = psi - rhs
        dl[k] = s.solve_JbT(phi)
        psi = -(s.Ja.T @ dl[k])
        om_next, La_next = om, s.La
    # ---- bound multipliers, scalars -------------------------------------------------------------------------------------
    dzb = np.zeros((K, 6))
    for k in range(K):
        dx3 = (dz[k, IA], dz[k, IM], du[k])
        for b in range(3):
            zl, zu = pr.ZB[k, 2 * b], pr.ZB[k, 2 * b + 1]
            dzb[k, 2 * b] = idn[k, 2 * b] * (mu - zl * dx3[b]) - zl
            dzb[k, 2 * b + 1] = idn[k, 2 * b + 1] * (mu + zu * dx3[b]) - zu
    ds1 = cg1 + g1g @ dz[-1]; ds2 = cg2 + g2g @ dz[-1]
    dnu1, dnu2 = sig1 * ds1 + rs1, sig2 * ds2 + rs2
    dzs1 = mu / s1 - zs1 - zs1 / s1 * ds1; dzs2 = mu / s2 - zs2 - zs2 / s2 * ds2
    dzlt = mu * itl - zlt - zlt * itl * dth; dzut = mu * itu - zut + zut * itu * dth
    step = np.zeros_like(blob)
    step[:7 * K] = dz.ravel(); step[7 * K:8 * K] = du; step[8 * K:15 * K] = dl.ravel(); step[15 * K:21 * K] = dzb.ravel()
    step[21 * K:] = [dth, dzlt, dzut, ds1, ds2, dzs1, dzs2, dnu3, dnu1, dnu2]
    return step, 0
