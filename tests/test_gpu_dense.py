"""GPU parity tests of the dense-block solver path (csrc/ascent_dense.hip: d_eval -> d_newton, one wavefront per NLP):
Hermite-Simpson (scheme 2), the ellipse-proper terminal constraints, the coast arc -- BASELINE.json configs[4] -- and, as
a cross-check of two independent HIP implementations, schemes 0/1 through the dense path against the hand-tuned
kernels.  The checker for scheme 2 is oracle/ascent_general.py (sympy-generated derivatives, generic sparse LU); the
HIP side derives the same blocks by hand."""
import json
import os

import numpy as np
import pytest

import lunar_module_ascent_trajectory_optimiser_amd as A

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)
X, Y, VX, VY, AN, W, MS = range(7)


def _interior_blobs(coracle, S, nt, scheme):
    """Strictly interior primal-dual iterates (a few oracle iterations of a scheme the C oracle has, multipliers perturbed)."""
    blobs = []
    for b, row in enumerate(S):
        rng = np.random.default_rng(300 + b)
        r = coracle.solve_batch(row[None], nt, 3 + b % 4, 1e-9, want_blob=True, coarse_nodes=-1, scheme=min(scheme, 1))
        blob = r["blob"][0].copy()
        K = nt - 1
        blob[8 * K:15 * K] += 0.05 * rng.standard_normal(7 * K)
        blob[15 * K:21 * K] *= rng.uniform(0.7, 1.3, 6 * K)
        blobs.append(blob)
    coracle.set_scheme(0)
    return np.stack(blobs, axis=1)


@pytest.mark.parametrize("scheme", [0, 1, 2])
def test_dense_stage_records_match_symbolic_derivatives(coracle, scheme):
    """d_eval's dense blocks -- d c_k/d z_{k-1}, d c_k/d z_k, the three Hessian blocks of lambda_k'c_k, the theta and
    control columns -- against the Jacobian and Hessian sympy generates from the one-line symbolic step defect."""
    from conftest import params_of_row
    from oracle.ascent_general import GeneralNLP
    nt = 24
    K = nt - 1
    S = A.sweep_isp_drymass(2, 1)
    blobs = _interior_blobs(coracle, S, nt, scheme)
    rec = A.dense_records(S, blobs, nt, scheme=scheme)
    for b in range(len(S)):
        blob = blobs[:, b]
        nlp = GeneralNLP(params_of_row(S[b]), ((K, "burn"),), scheme)
        v = np.zeros(nlp.n); lam = np.zeros(nlp.m)
        v[:8 * K].reshape(K, 8)[:, :7] = blob[:7 * K].reshape(K, 7)
        v[:8 * K].reshape(K, 8)[:, 7] = blob[7 * K:8 * K]
        lam[:7 * K] = blob[8 * K:15 * K]
        v[nlp.itf] = blob[21 * K]
        c, J, H = nlp._steps(v, lam)
        for k in range(K):
            g = rec[b, k]
            scale = max(1.0, np.abs(H[k]).max())
            assert np.abs(g[5, 0, :7] - c[k]).max() < 1e-13
            assert np.abs(g[1, :7, :7] - J[k][:, 7:14]).max() < 1e-12
            if k:
                assert np.abs(g[0, :7, :7] - J[k][:, 0:7]).max() < 1e-12
                assert np.abs(g[2, :7, :7] - H[k][0:7, 0:7]).max() < 1e-11 * scale
                assert np.abs(g[3, :7, :7] - H[k][0:7, 7:14]).max() < 1e-11 * scale
                assert np.abs(g[5, 3, :7] - H[k][0:7, 15]).max() < 1e-11 * scale
            assert np.abs(g[4, :7, :7] - H[k][7:14, 7:14]).max() < 1e-11 * scale
            assert np.abs(g[5, 1, :7] - J[k][:, 14]).max() < 1e-13
            assert np.abs(g[5, 2, :7] - J[k][:, 15]).max() < 1e-12 * max(1.0, np.abs(J[k][:, 15]).max())
            assert np.abs(g[5, 4, :7] - H[k][7:14, 15]).max() < 1e-11 * scale
            assert g[1, 7, 7] == 1.0 and np.abs(g[0, 7]).max() == 0.0          # the padding slot


@pytest.mark.parametrize("scheme", [0, 1, 2])
def test_dense_newton_step_matches_generic_lu(coracle, scheme):
    """One Newton step of the barrier problem through d_eval -> d_newton (dense Riccati recursion with two border
    columns, Gauss-Jordan step-Jacobian inverse, multiplier step from the forward pass) against the generic sparse-LU
    solve of the full KKT matrix -- for scheme 2 assembled from the sympy-generated derivatives -- and, for schemes 0/1,
    against the C oracle's stage-wise step as well; with and without primal regularisation."""
    from conftest import generic_lu_newton_step, params_of_row
    nt = 40
    K = nt - 1
    S = A.sweep_isp_drymass(2, 2)
    blobs = _interior_blobs(coracle, S, nt, scheme)
    mu = np.array([0.1, 0.02, 1e-3, 0.05]); dw = np.array([0.0, 1e-4, 1e-2, 1.0])
    step, inertia = A.kkt_step(S, blobs, mu, dw, nt, path="dense", scheme=scheme)
    n_ok = 0
    for b in range(len(S)):
        blob = np.ascontiguousarray(blobs[:, b])
        if scheme < 2:
            rc, ref = coracle.newton_step(S[b], nt, blob, mu[b], dw[b], scheme=scheme)
            coracle.set_scheme(0)
            assert rc == inertia[b]
            if rc == 0:
                assert np.abs(step[:, b] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
        if inertia[b]:
            continue
        n_ok += 1
        lu, _, _, _ = generic_lu_newton_step(params_of_row(S[b]), nt, blob, mu[b], dw[b], scheme)
        for lo, hi in ((0, 8 * K), (8 * K, 15 * K), (15 * K, 21 * K), (21 * K, 21 * K + 10)):
            assert np.abs(step[lo:hi, b] - lu[lo:hi]).max() <= 1e-8 * max(1.0, np.abs(lu[lo:hi]).max())
    assert n_ok >= 3


def test_dense_path_equals_hand_tuned_path_on_schemes_0_and_1(coracle):
    """Two independent HIP implementations of the same interior point: the hand-tuned sparse kernels (closed-form step
    Jacobian inverse, packed congruences) and the dense-block path.  Same iteration counts, t_f to rounding, both the oracle's."""
    S = A.sweep_isp_drymass(5, 4)
    for scheme in (0, 1):
        d = A.solve_batch(S, 200, tol=1e-9, scheme=scheme, path="dense")
        h = A.solve_batch(S, 200, tol=1e-9, scheme=scheme)
        ref = coracle.solve_batch(S, 200, 300, 1e-9, scheme=scheme)
        coracle.set_scheme(0)
        assert np.all(d.status == 0) and np.all(h.status == 0)
        assert np.array_equal(d.iters, h.iters) and np.array_equal(d.iters, ref["iters"])
        assert np.abs(d.tf - h.tf).max() <= 1e-11 and np.abs(d.tf - ref["tf"]).max() <= 1e-9
        for f in (0, 1, 2, 3, 6, 9):
            assert np.abs(d.traj[f] - h.traj[f]).max() <= 1e-6 * max(1.0, np.abs(h.traj[f]).max())


def _fixtures():
    fx = json.load(open(os.path.join(HERE, "golden", "hs_fixtures.json")))
    return {(c["nt"], c["scheme"], c["terminal"]): c for c in fx["cases"]}


def test_hermite_simpson_matches_independent_fixtures():
    """scheme = 2 against tests/golden/hs_fixtures.json (generalised numpy oracle; made by scripts/make_hs_fixtures.py):
    t_f to 1e-9, final state to 1e-6, sampled trajectory to 1e-6 (angle 1e-5) -- at N = 50, 200, 400 with the reference's terminal
    constraints, at N = 200 with the ellipse-proper ones (and at N = 2000 when that fixture has been generated)."""
    by = _fixtures()
    for (nt, scheme, terminal), c in sorted(by.items()):
        r = A.solve_batch(A.AscentParams(), nt, tol=1e-10, scheme=scheme, terminal={"reference": "reference", "periapsis": "ellipse", "ellipse": "ellipse_free"}[terminal],
                          max_iter=500)
        assert r.status[0] == 0
        assert abs(r.tf[0] - c["tf"]) <= 1e-9, (nt, scheme, terminal, r.tf[0], c["tf"])
        fs = np.array([r.traj[f][-1, 0] for f in (0, 1, 2, 3, 6, 7, 9)])
        assert np.abs(fs - np.array(c["final_state"])).max() <= 1e-6
        stride = max(1, (nt - 1) // 20)
        # (the angle around the junction of the saturated and the singular arc is only weakly determined by a KKT point)
        for name, key, tol in (("x", "x", 1e-6), ("y", "y", 1e-6), ("angle", "angle", 1e-5)):
            assert np.abs(r.field(name)[::stride, 0] - np.array(c[key])).max() <= tol
        o = r.orbit()
        assert abs(o["periapsis_alt"][0] - c["orbit_periapsis_alt_m"]) < 0.01 and abs(o["apoapsis_alt"][0] - c["orbit_apoapsis_alt_m"]) < 0.01


def test_hermite_simpson_sweep_and_mesh_refinement():
    """A sweep through scheme 2: every NLP converges; HS at N = 200 is within 0.01 s of HS at N = 800 on every problem (the
    backward-Euler answers of the same problems are 1.2 s away from their own limit), and the limit agrees with the
    trapezoid answers at N = 800 (the two second-order-or-better schemes of this repository, independent code paths)."""
    S = A.sweep_isp_drymass(4, 4)
    h2 = A.solve_batch(S, 200, tol=1e-9, scheme=2)
    h8 = A.solve_batch(S, 800, tol=1e-9, scheme=2, max_iter=500)
    t8 = A.solve_batch(S, 800, tol=1e-9, scheme=1, max_iter=500)
    assert np.all(h2.status == 0) and np.all(h8.status == 0) and np.all(t8.status == 0)
    assert np.abs(h2.final_time() - h8.final_time()).max() < 0.01
    assert np.abs(h8.final_time() - t8.final_time()).max() < 0.01
    be = A.solve_batch(S, 200, tol=1e-9)
    assert np.all(h8.final_time() - be.final_time() > 1.0)


def test_config5_high_resolution_burn_then_coast():
    """BASELINE.json configs[4]: N = 2000 Hermite-Simpson, angular-acceleration bound active, burn to the 17.7 x 88.6 km
    ellipse proper (README.md:7) followed by the coast arc to its apoapsis, propagated on the device.  Size-independent
    properties on a 64-NLP sweep: every problem converges; the burnout orbit is the target ellipse to 1 m for every
    problem (BatchResult.orbit(): host-side closed form; coast(): the device's own elements); the coast ends at r_apo with
    r.v = 0; its duration is half the orbital period; the bang-bang part of the control sits on its bound."""
    S = np.vstack([A.AscentParams().as_row()[None], A.sweep_isp_drymass(9, 7)])
    r = A.solve_batch(S, 2000, tol=1e-9, scheme="hermite_simpson", terminal="ellipse", max_iter=500)
    assert np.all(r.status == 0)
    o = r.orbit()
    assert np.abs(o["periapsis_alt"] - 17703.0).max() < 1.0 and np.abs(o["apoapsis_alt"] - 88615.0).max() < 1.0
    assert np.abs(o["flight_path_angle"]).max() < 1e-8
    c = r.coast(coast_nodes=2000)
    assert np.abs(c["periapsis_alt"] - 17703.0).max() < 1.0 and np.abs(c["apoapsis_alt"] - 88615.0).max() < 1.0
    Sx, R0, GM = 17703.0, 1738100.0, 6.674e-11 * 7.346e22
    Xe, Ye = c["traj"][0, -1] * Sx, c["traj"][1, -1] * Sx + R0
    VXe, VYe = c["traj"][2, -1] * Sx, c["traj"][3, -1] * Sx
    assert np.abs(np.hypot(Xe, Ye) - R0 - 88615.0).max() < 1.0
    assert np.abs(Xe * VXe + Ye * VYe).max() / (np.hypot(Xe, Ye) * np.hypot(VXe, VYe)).min() < 1e-9
    a = R0 + 0.5 * (17703.0 + 88615.0)
    assert np.abs(c["tf"] * 470.0 - np.pi * np.sqrt(a ** 3 / GM)).max() < 0.05
    # first node of the coast = last node of the burn; energy and angular momentum constant along the arc
    assert np.abs(c["traj"][:, 0, :] - r.traj[:4, -1, :]).max() < 1e-10
    X_, Y_ = c["traj"][0] * Sx, c["traj"][1] * Sx + R0
    VX_, VY_ = c["traj"][2] * Sx, c["traj"][3] * Sx
    en = 0.5 * (VX_ ** 2 + VY_ ** 2) - GM / np.hypot(X_, Y_)
    hm = X_ * VY_ - Y_ * VX_
    assert np.abs(en / en[0] - 1).max() < 1e-10 and np.abs(hm / hm[0] - 1).max() < 1e-10
    u = r.field("angledoubledot")
    assert np.all(np.abs(u) <= 1 + 1e-9) and (np.abs(u[:, 0]) > 0.999).sum() >= 400     # 47 of 199 steps at N = 200
    # mesh refinement: the same problems at N = 200 land within 0.01 s
    r2 = A.solve_batch(S[:8], 200, tol=1e-9, scheme=2, terminal="ellipse")
    assert np.abs(r2.final_time() - r.final_time()[:8]).max() < 0.01
    # the generalised numpy oracle's solution of the nominal problem at this size (17 minutes of sympy-generated numpy; the
    # fixture was made at tol 1e-10: with 12 000 bound terms the barrier's pull on t_f is 5e-8 at tol 1e-9)
    by = _fixtures()
    if (2000, 2, "periapsis") in by:
        r10 = A.solve_batch(A.AscentParams(), 2000, tol=1e-10, scheme=2, terminal="ellipse", max_iter=500)
        assert r10.status[0] == 0 and abs(r10.tf[0] - by[(2000, 2, "periapsis")]["tf"]) <= 1e-9
        assert abs(r.tf[0] - r10.tf[0]) < 1e-7


def test_ellipse_terminal_through_the_hand_tuned_kernels(coracle):
    """ascent_opts.terminal = 1 is available on every path (the solvers run on a parameter copy whose mean radius
    reproduces the vis-viva speed): backward Euler through the 16-lane split kernels against the C oracle run on
    parameters transformed the same way on the host, and against the generalised numpy oracle's fixture."""
    S = A.sweep_isp_drymass(3, 3)
    r = A.solve_batch(S, 200, tol=1e-9, terminal="ellipse")
    assert np.all(r.status == 0)
    f = A.PARAM_FIELDS
    T = S.copy()
    rp, ra = T[:, f.index("R0")] + T[:, f.index("r_peri")], T[:, f.index("R0")] + T[:, f.index("r_apo")]
    T[:, f.index("r_apo")] = 2.0 * (1.0 / (2.0 / rp - 2.0 / (rp + ra)) - T[:, f.index("R0")]) - T[:, f.index("r_peri")]
    ref = coracle.solve_batch(T, 200, 300, 1e-9)
    assert np.abs(r.tf - ref["tf"]).max() <= 1e-9 and np.array_equal(r.iters, ref["iters"])
    o = r.orbit()
    assert np.abs(o["periapsis_alt"] - 17703.0).max() < 0.01 and np.abs(o["apoapsis_alt"] - 88615.0).max() < 0.01
    n = A.solve_batch(A.AscentParams(), 200, tol=1e-10, terminal="ellipse")
    assert abs(n.tf[0] - _fixtures()[(200, 0, "periapsis")]["tf"]) <= 1e-9


def test_terminal2_burnout_anywhere_on_the_ellipse_then_coast():
    """BASELINE config 5's burn--coast problem with the coast arc eliminated exactly (ascent_opts.terminal = 2): burnout ANYWHERE on
    the (r_peri, r_apo) ellipse -- angular momentum and specific energy of the ellipse, no r.v = 0 -- then the Kepler coast from
    that anomaly (ascent_coast_batch).  Against the generalised numpy oracle's fixtures (terminal "ellipse": t_f to 1e-9);
    size-independent properties on a sweep and at N = 2000 Hermite-Simpson: the burnout orbit is the target ellipse to 1 m on
    every problem, the burn is never longer than the periapsis insertion's (terminal 1), burnout is within 100 m of the periapsis
    radius but not at it (r.v != 0), the coast from there ends at the apoapsis."""
    by = _fixtures()
    n = 0
    for (nt, scheme, terminal), c in sorted(by.items()):
        if terminal != "ellipse" or nt > 400:
            continue
        r = A.solve_batch(A.AscentParams(), nt, tol=1e-10, scheme=scheme, terminal="ellipse_free", max_iter=500)
        assert r.status[0] == 0 and abs(r.tf[0] - c["tf"]) <= 1e-9, (nt, scheme, r.tf[0], c["tf"])
        fs = np.array([r.traj[f][-1, 0] for f in (0, 1, 2, 3, 6, 7, 9)])
        assert np.abs(fs - np.array(c["final_state"])).max() <= 1e-6
        n += 1
    assert n >= 3
    assert A.default_path(16, 200, scheme=0) == "persist"
    S = A.sweep_isp_drymass(4, 4)
    for scheme in (0, 1, 2):      # schemes 0 / 1: the persistent kernel (p_solve<.,0,.,2>), scheme 2: the dense-block path
        r1 = A.solve_batch(S, 200, tol=1e-9, scheme=scheme, terminal="ellipse", max_iter=500)
        r2 = A.solve_batch(S, 200, tol=1e-9, scheme=scheme, terminal="ellipse_free", max_iter=500)
        assert np.all(r1.status == 0) and np.all(r2.status == 0)
        if scheme < 2:            # ... against the dense-block path: two independent HIP implementations of the terminal block
            rd = A.solve_batch(S, 200, tol=1e-9, scheme=scheme, terminal="ellipse_free", max_iter=500, path="dense")
            assert np.all(rd.status == 0) and np.abs(rd.tf - r2.tf).max() <= 1e-9
        o = r2.orbit()
        assert np.abs(o["periapsis_alt"] - 17703.0).max() < 1.0 and np.abs(o["apoapsis_alt"] - 88615.0).max() < 1.0
        assert np.all(r2.tf <= r1.tf + 1e-12) and np.all(r1.tf - r2.tf < 1e-4)
        assert np.abs(o["flight_path_angle"]).min() > 1e-6
        c2 = r2.coast(coast_nodes=256)
        Sx, R0 = 17703.0, 1738100.0
        Xe, Ye = c2["traj"][0, -1] * Sx, c2["traj"][1, -1] * Sx + R0
        assert np.abs(np.hypot(Xe, Ye) - R0 - 88615.0).max() < 1.0
    # config 5's grid: N = 2000 Hermite-Simpson
    n1 = A.solve_batch(A.AscentParams(), 2000, tol=1e-9, scheme=2, terminal="ellipse", max_iter=500)
    n2 = A.solve_batch(A.AscentParams(), 2000, tol=1e-9, scheme=2, terminal="ellipse_free", max_iter=500)
    assert n1.status[0] == 0 and n2.status[0] == 0 and n2.tf[0] <= n1.tf[0] and n1.tf[0] - n2.tf[0] < 1e-4
    o = n2.orbit()
    assert abs(o["periapsis_alt"][0] - 17703.0) < 1.0 and abs(o["apoapsis_alt"][0] - 88615.0) < 1.0
    Xb, Yb = n2.traj[0][-1, 0] * 17703.0, n2.traj[1][-1, 0] * 17703.0 + 1738100.0
    assert 0.0 < np.hypot(Xb, Yb) - 1738100.0 - 17703.0 < 100.0
    if (2000, 2, "ellipse") in by:
        r10 = A.solve_batch(A.AscentParams(), 2000, tol=1e-10, scheme=2, terminal="ellipse_free", max_iter=500)
        assert r10.status[0] == 0 and abs(r10.tf[0] - by[(2000, 2, "ellipse")]["tf"]) <= 1e-9


def test_coast_arc_against_the_closed_form():
    """ascent_coast_batch on synthetic states: an orbit given by its elements, a start at an arbitrary true anomaly."""
    P = A.AscentParams()
    GM, R0, Sx = P.G * P.M, P.R0, P.r_peri
    a, e = R0 + 60e3, 0.03
    states, expect = [], []
    for nu0 in (0.0, 0.7, 2.5, 3.6):
        p_ = a * (1 - e * e)
        r0 = p_ / (1 + e * np.cos(nu0))
        vr, vt = np.sqrt(GM / p_) * e * np.sin(nu0), np.sqrt(GM / p_) * (1 + e * np.cos(nu0))
        ph = 0.2 + nu0                                   # polar angle from +y towards -x, as the ascent flies
        Xp, Yp = -r0 * np.sin(ph), r0 * np.cos(ph)
        VXp, VYp = -vr * np.sin(ph) - vt * np.cos(ph), vr * np.cos(ph) - vt * np.sin(ph)
        states.append([Xp / Sx, (Yp - R0) / Sx, VXp / Sx, VYp / Sx])
        E0 = 2 * np.arctan2(np.sqrt(1 - e) * np.sin(nu0 / 2), np.sqrt(1 + e) * np.cos(nu0 / 2))
        M0 = (E0 - e * np.sin(E0)) % (2 * np.pi)
        expect.append(((np.pi - M0) % (2 * np.pi)) / np.sqrt(GM / a ** 3))
    st = np.ascontiguousarray(np.array(states).T)
    c = A.coast_batch(np.repeat(P.as_row()[None], 4, 0), st, coast_nodes=64)
    assert np.abs(c["tf"] * P.T_scale - np.array(expect)).max() < 1e-6
    assert np.abs(c["apoapsis_alt"] - (a * (1 + e) - R0)).max() < 1e-5 and np.abs(c["periapsis_alt"] - (a * (1 - e) - R0)).max() < 1e-5
    Xe, Ye = c["traj"][0, -1] * Sx, c["traj"][1, -1] * Sx + R0
    assert np.abs(np.hypot(Xe, Ye) - a * (1 + e)).max() < 1e-4
    assert np.abs(c["traj"][:, 0, :] - st).max() < 1e-9


def test_pcr_newton_variant_and_small_batch_dispatch(coracle, monkeypatch):
    """The dense-block path solves its Newton systems by the serial Riccati recursion (one wavefront per NLP) or by
    parallel cyclic reduction over the collocation nodes (one wavefront per node, 15x15 blocks on MFMA FP64 tiles;
    ASCENT_DENSE_NEWTON=riccati|pcr, automatic: PCR up to 64 NLPs).  Same Newton step to rounding; same iterates, so the
    same iteration counts and answers -- also the oracle's.  A handful of NLPs on a long grid are routed to the dense-block path
    with PCR automatically (ASCENT_SMALL_BATCH=off keeps the persistent kernels, scheme 2's included): all agree."""
    nt = 40
    S = A.sweep_isp_drymass(2, 2)
    blobs = _interior_blobs(coracle, S, nt, 2)
    mu = np.array([0.1, 0.02, 1e-3, 0.05]); dw = np.array([0.0, 1e-4, 1e-2, 1.0])
    steps = {}
    for mode in ("riccati", "pcr"):
        monkeypatch.setenv("ASCENT_DENSE_NEWTON", mode)
        steps[mode] = A.kkt_step(S, blobs, mu, dw, nt, path="dense", scheme=2)
    assert np.array_equal(steps["riccati"][1], steps["pcr"][1]) and not steps["pcr"][1].any()
    assert np.abs(steps["riccati"][0] - steps["pcr"][0]).max() <= 1e-9 * np.abs(steps["riccati"][0]).max()
    for B, ntf, scheme in ((1, 200, 0), (5, 200, 1), (3, 200, 2), (1, 1000, 0), (2, 1000, 2), (9, 60, 0)):
        P = np.vstack([A.AscentParams().as_row()[None], A.sweep_isp_drymass(3, 3)])[:B]
        res = {}
        for name, env in (("pcr", {"ASCENT_DENSE_NEWTON": "pcr"}), ("riccati", {"ASCENT_DENSE_NEWTON": "riccati"}),
                          ("auto", {}), ("hand", {"ASCENT_SMALL_BATCH": "off"})):
            monkeypatch.delenv("ASCENT_DENSE_NEWTON", raising=False); monkeypatch.delenv("ASCENT_SMALL_BATCH", raising=False)
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            res[name] = A.solve_batch(P, ntf, tol=1e-9, scheme=scheme, max_iter=500, path="dense" if name in ("pcr", "riccati") else "auto")
            assert np.all(res[name].status == 0)
        auto_is_pcr = B <= (0 if ntf - 1 < 400 else min(6, (ntf - 1) // 300))     # the dispatch rule (ascent_solver.hip)
        monkeypatch.delenv("ASCENT_SMALL_BATCH", raising=False)
        assert A.default_path(B, ntf, scheme=scheme) == ("dense" if auto_is_pcr else "persist")
        for name in res:
            # (cold starts and the coarse levels of the nested iteration run through inertia corrections, where the PCR variant's
            #  curvature rule and the exact inertia of the recursions legitimately choose different regularisations now and
            #  then: same optimum, a few iterations more or less.  Variants with the same rule take the same path.)
            same_rule = name == "pcr" or (name == "auto" and auto_is_pcr)
            if ntf >= 200:
                assert np.array_equal(res[name].iters, res["pcr" if same_rule else "riccati"].iters), (B, ntf, scheme, name)
                assert np.abs(res[name].iters.astype(int) - res["pcr"].iters).max() <= 4, (B, ntf, scheme, name)
            assert np.abs(res[name].tf - res["pcr"].tf).max() <= (1e-11 if ntf >= 200 else 1e-9)
        if scheme < 2:
            ref = coracle.solve_batch(P, ntf, 500, 1e-9, scheme=scheme)
            coracle.set_scheme(0)
            assert np.abs(res["riccati"].tf - ref["tf"]).max() <= 1e-9 and np.abs(res["riccati"].iters.astype(int) - ref["iters"]).max() <= 1
            assert np.abs(res["auto"].tf - ref["tf"]).max() <= 1e-9 and np.abs(res["auto"].iters.astype(int) - ref["iters"]).max() <= 4


def test_move_penalty_matches_independent_fixtures():
    """a12, Launch_Optimiser.py:99: the MV's DCOST as an l1 term (ascent_opts.move_penalty = 1: the control as the eighth state of
    a stage, the slack pair reduced to one pivot -- through the default dispatch, i.e. the persistent kernel for schemes 0 / 1
    and the dense-block path for scheme 2, and through the dense-block path for every scheme) against tests/golden/dcost_fixtures.json -- the numpy
    generic-LU oracle's solutions with and without the penalty (scripts/make_dcost_fixtures.py): backward Euler, trapezoid and
    Hermite-Simpson, dcost 1e-5 (the reference's) to 1e-3, nominal and off-nominal parameters, nested grids.  t_f to 2e-8
    (fixtures at tol 1e-10, GPU at 1e-9), the control's total variation to 0.2 %, the control itself to 5e-3 of its [-1, 1]
    range; switched off, the same call reproduces the unpenalised fixture (dcost is ignored)."""
    import json, os
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dcost_fixtures.json")))
    assert len(fx["cases"]) >= 6
    for c in fx["cases"]:
        P = A.AscentParams(**c["params"])
        for on, path in ((False, "auto"), (True, "auto"), (True, "dense")):
            r = A.solve_batch(P, c["nt"], tol=1e-9, scheme=c["scheme"], max_iter=500, move_penalty=on, path=path)
            ref = c["on" if on else "off"]
            u = r.traj[8, 1:, 0]
            tv = np.abs(np.diff(np.concatenate([[0.0], u]))).sum()
            assert r.status[0] == 0, (c["nt"], c["scheme"], c["dcost"], on)
            assert abs(r.tf[0] - ref["tf"]) <= 2e-8, (c["nt"], c["scheme"], c["dcost"], on, r.tf[0], ref["tf"])
            if on:
                assert abs(tv - ref["total_variation"]) <= 2e-3 * ref["total_variation"]
                assert np.abs(u - np.array(ref["u"])).max() <= 5e-3
        assert c["on"]["tf"] > c["off"]["tf"] and c["on"]["total_variation"] < c["off"]["total_variation"]
    nominal = [c for c in fx["cases"] if c["nt"] == 200 and c["dcost"] == 1e-5][0]
    assert 1.0e-3 < (nominal["on"]["tf"] - nominal["off"]["tf"]) * 470.0 < 2.0e-3       # +1.5e-3 s on the reference's problem


def test_move_penalty_batch_and_dispatch():
    """A batch through move_penalty = 1: every NLP converges, the penalty raises t_f and lowers the control's total variation on
    every problem; the default dispatch reports the persistent kernel for it (schemes 0 / 1); bad values are refused."""
    S = A.sweep_isp_drymass(3, 3)
    assert np.all(S[:, 15] == 0.0)                      # (this package's sweeps carry no weight of their own)
    from lunar_module_ascent_trajectory_optimiser_amd._lib import AscentLibraryError
    with pytest.raises(AscentLibraryError, match="dcost > 0"):
        A.solve_batch(S, 100, tol=1e-9, move_penalty=True)
    S[:, 15] = 1e-5                                     # the reference's DCOST
    assert A.default_path(9, 100, move_penalty=True) == "persist" and A.default_path(4096, 200, move_penalty=True) == "persist"
    assert A.default_path(9, 100, scheme=2, move_penalty=True) == "dense"
    off = A.solve_batch(S, 100, tol=1e-9)
    on = A.solve_batch(S, 100, tol=1e-9, move_penalty=True, max_iter=500)
    assert np.all(off.status == 0) and np.all(on.status == 0)
    tv = lambda r: np.abs(np.diff(r.traj[8], axis=0)).sum(axis=0)
    assert np.all(on.tf > off.tf) and np.all(on.tf - off.tf < 1e-4) and np.all(tv(on) < tv(off))
    # a larger weight flattens the control further
    S2 = S.copy(); S2[:, 15] = 1e-3
    on2 = A.solve_batch(S2, 100, tol=1e-9, move_penalty=True, max_iter=500)
    assert np.all(on2.status == 0) and np.all(tv(on2) < tv(on)) and np.all(on2.tf > on.tf)
    # both Newton solvers of the dense path carry the penalty: parallel cyclic reduction over the nodes (<= 32 NLPs by default;
    # 16x16 node blocks: 8 states, 8 multipliers, delta eliminated into the movement equation) and the Riccati recursion
    for env in ("pcr", "riccati"):
        os.environ["ASCENT_DENSE_NEWTON"] = env
        try:
            rr = A.solve_batch(S, 100, tol=1e-9, move_penalty=True, max_iter=500, path="dense")
        finally:
            del os.environ["ASCENT_DENSE_NEWTON"]
        assert np.all(rr.status == 0) and np.abs(rr.tf - on.tf).max() <= 2e-8 and np.abs(rr.iters.astype(int) - on.iters).max() <= 4
    # nine orders of magnitude of the weight in one batch: all converge, t_f and the total variation are monotone in it,
    # and the rise of t_f is bounded by the penalty the unpenalised control would pay
    W = np.array([1e-9, 1e-7, 1e-5, 1e-3, 1e-1, 1.0])
    Sw = np.tile(A.AscentParams().as_row(), (len(W), 1)); Sw[:, 15] = W
    rw = A.solve_batch(Sw, 200, tol=1e-9, move_penalty=True, max_iter=500)
    r0 = A.solve_batch(Sw[:1], 200, tol=1e-9)
    assert np.all(rw.status == 0) and rw.iters.max() <= 60
    assert np.all(np.diff(rw.tf) > 0) and np.all(np.diff(tv(rw)) < 0)
    assert np.all(rw.tf - r0.tf[0] <= W * tv(r0)[0] + 1e-9) and rw.tf[0] - r0.tf[0] < 1e-8


def test_move_penalty_across_the_config4_box_and_on_long_grids():
    """move_penalty = 1 away from the nominal problem and on long grids: 64 problems across BASELINE config 4's box (the
    reference's weight 1e-5) all converge, t_f rises by 1e-3 .. 2e-3 s on each; N = 1000 trapezoid and N = 2000 Hermite-Simpson
    with the ellipse terminal condition (config 5's grid) converge, t_f + 1.6e-3 s, the control's total variation 7.3 -> 4.4."""
    S = A.sweep_config4()[::4099][:64].copy()
    S[:, 15] = 1e-5
    off = A.solve_batch(S, 200, tol=1e-9, want_traj=False)
    on = A.solve_batch(S, 200, tol=1e-9, want_traj=False, move_penalty=True, max_iter=500)
    assert np.all(off.status == 0) and np.all(on.status == 0)
    shift = (on.tf - off.tf) * 470.0
    assert shift.min() > 5e-4 and shift.max() < 3e-3
    P = A.AscentParams(dcost=1e-5)
    tv = lambda r: np.abs(np.diff(np.concatenate([[0.0], r.traj[8, 1:, 0]]))).sum()
    for nt, scheme, term in ((1000, 1, "reference"), (2000, 2, "ellipse")):
        r0 = A.solve_batch(P, nt, tol=1e-9, scheme=scheme, terminal=term, max_iter=500)
        r1 = A.solve_batch(P, nt, tol=1e-9, scheme=scheme, terminal=term, max_iter=500, move_penalty=True)
        assert r0.status[0] == 0 and r1.status[0] == 0
        assert 1.0e-3 < (r1.tf[0] - r0.tf[0]) * 470.0 < 2.5e-3 and tv(r1) < 0.7 * tv(r0)


@pytest.mark.gpu
def test_pcr_newton_step_at_a_wrong_inertia_iterate(coracle, monkeypatch):
    """What the cyclic-reduction variant guarantees where the Riccati forms refuse: at an iterate with strongly negative curvature
    (flipped defect multipliers) the recursion reports the wrong inertia; parallel cyclic reduction exposes no pivots, so it either
    refuses too (its curvature test along the step) or hands back a step -- and then that step must be the solution of the KKT system
    as it stands (generic sparse LU of the full matrix, nothing stage-structured): a correct Newton step of an indefinite system,
    whose acceptance is the line search's business.  Also with a regularisation large enough to restore the inertia: both forms and
    the LU agree.  (include/ascent.h documents the weaker guarantee; the dispatch routes only a handful of NLPs on long grids here.)"""
    from conftest import generic_lu_newton_step, params_of_row, random_interior_blob
    nt = 40
    Kk = nt - 1
    S = A.sweep_isp_drymass(1, 1)
    blob = random_interior_blob(nt, 2, S[0], coracle)
    bad = blob.copy()
    bad[8 * Kk:15 * Kk] *= -50.0
    out = {}
    for mode in ("riccati", "pcr"):
        monkeypatch.setenv("ASCENT_DENSE_NEWTON", mode)
        out[mode] = A.kkt_step(S, bad[:, None], 1e-6, 0.0, nt, path="dense")
        out[mode + "_reg"] = A.kkt_step(S, bad[:, None], 1e-6, 1e4, nt, path="dense")
    assert out["riccati"][1][0] == 1                                     # the exact inertia says no
    lu = generic_lu_newton_step(params_of_row(S[0]), nt, bad, 1e-6, 0.0)
    lu = lu[0] if isinstance(lu, tuple) else lu
    if out["pcr"][1][0] == 0:
        assert np.abs(out["pcr"][0][:, 0] - lu).max() <= 1e-7 * max(1.0, np.abs(lu).max())
    # a regularisation that makes the (z, u) block positive definite again: everybody solves the same system
    lur = generic_lu_newton_step(params_of_row(S[0]), nt, bad, 1e-6, 1e4)
    lur = lur[0] if isinstance(lur, tuple) else lur
    for mode in ("riccati_reg", "pcr_reg"):
        assert out[mode][1][0] == 0, mode
        assert np.abs(out[mode][0][:, 0] - lur).max() <= 1e-7 * max(1.0, np.abs(lur).max()), mode
