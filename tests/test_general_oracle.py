"""CPU tests of the generalised oracle (oracle/ascent_general.py: sympy-generated derivatives, generic sparse LU): it is
the checker of the widened rows -- Hermite-Simpson (scheme 2), the ellipse-proper terminal constraints, the DCOST l1
movement penalty -- so it is itself checked here against the hand-written numpy oracle where the two overlap, and pinned
by mesh refinement where the reference has nothing to pin it with (parity unpinned: the reference's NODES=2 is backward
Euler, its target speed is LO:72-78's mean-radius circular speed, and GEKKO is not installed)."""
import json
import os

import numpy as np
import pytest

from oracle.ascent_general import GeneralNLP, kepler_elements
from oracle.ascent_numpy import AscentNLP, Params, accel, solve_ip

HERE = os.path.dirname(__file__)


@pytest.mark.parametrize("scheme", [0, 1])
def test_symbolic_derivatives_equal_the_hand_written_oracle(scheme):
    P = Params()
    g, a = GeneralNLP(P, ((59, "burn"),), scheme), AscentNLP(P, 60, 0, scheme=scheme)
    rng = np.random.default_rng(scheme)
    v = a.initial_guess() + 0.01 * rng.standard_normal(a.n)
    lam = rng.standard_normal(a.m)
    assert np.abs(g.constraints(v) - a.constraints(v)).max() < 1e-13
    assert abs(g.jacobian(v) - a.jacobian(v)).max() < 1e-13
    assert abs(g.hessian(v, lam) - a.hessian(v, lam)).max() < 1e-12


def test_hermite_simpson_derivatives_finite_difference():
    """The scheme-2 Jacobian and Hessian (generated symbolically) against central differences of the constraint function."""
    P = Params()
    g = GeneralNLP(P, ((7, "burn"),), 2)
    rng = np.random.default_rng(5)
    v = g.initial_guess() + 0.01 * rng.standard_normal(g.n)
    lam = rng.standard_normal(g.m)
    J = g.jacobian(v).toarray()
    H = g.hessian(v, lam).toarray()
    eps = 1e-6
    for col in rng.choice(g.n, 12, replace=False):
        e = np.zeros(g.n); e[col] = eps
        fd = (g.constraints(v + e) - g.constraints(v - e)) / (2 * eps)
        assert np.abs(fd - J[:, col]).max() < 1e-7 * max(1.0, np.abs(J[:, col]).max())
        gl = (g.jacobian(v + e).T @ lam - g.jacobian(v - e).T @ lam) / (2 * eps)
        assert np.abs(gl - H[:, col]).max() < 1e-6 * max(1.0, np.abs(H[:, col]).max())


def test_hermite_simpson_mesh_refinement():
    """HS on 50 nodes is already within 0.04 s of the mesh-converged ascent time of the trapezoid / Richardson limit
    (435.225 s: trapezoid at N = 2000 gives 435.2248 s, SURVEY.md Appendix C 435.217-435.227 s), where the reference's
    backward Euler needs N = 2000 to get within 0.12 s; the committed fixtures hold N = 200 and 400 (435.2268, 435.2252 s)."""
    nlp = GeneralNLP(Params(), ((49, "burn"),), 2)
    v, _, info = solve_ip(nlp, tol=1e-9, max_iter=300)
    assert info["status"] == "converged"
    assert abs(nlp.outputs(v)["final_time"] - 435.2585) < 2e-3
    fx = json.load(open(os.path.join(HERE, "golden", "hs_fixtures.json")))
    by = {(c["nt"], c["scheme"], c["terminal"]): c for c in fx["cases"]}
    assert abs(by[(50, 2, "reference")]["final_time"] - nlp.outputs(v)["final_time"]) < 1e-6
    t200, t400 = by[(200, 2, "reference")]["final_time"], by[(400, 2, "reference")]["final_time"]
    assert abs(t200 - 435.2248) < 0.01 and abs(t400 - 435.2248) < 0.01 and abs(t400 - 435.2248) < abs(t200 - 435.2248) + 1e-4


def test_ellipse_proper_terminal_constraints():
    """terminal = "periapsis": LO:158-173 with the vis-viva speed at the periapsis of the (r_peri, r_apo) ellipse instead of
    LO:72-78's circular speed of the mean radius -- the burnout orbit is then the 17.7 x 88.6 km ellipse of README.md:7,
    to a millimetre, where the reference's own terminal state lies on an orbit whose periapsis is under the surface."""
    P = Params()
    nlp = GeneralNLP(P, ((59, "burn"),), 0, terminal="periapsis")
    v, _, info = solve_ip(nlp, tol=1e-9, max_iter=300)
    assert info["status"] == "converged"
    o = nlp.outputs(v)
    peri, apo = kepler_elements(P, o["x"][-1], o["y"][-1], o["xdot"][-1], o["ydot"][-1])
    assert abs(peri - P.r_peri) < 0.01 and abs(apo - P.r_apo) < 0.01
    ref = GeneralNLP(P, ((59, "burn"),), 0)
    vr, _, _ = solve_ip(ref, tol=1e-9, max_iter=300)
    orr = ref.outputs(vr)
    pr, ar = kepler_elements(P, orr["x"][-1], orr["y"][-1], orr["xdot"][-1], orr["ydot"][-1])
    assert pr < 0 and abs(ar - P.r_peri) < 1.0          # the reference's insertion point is the APOAPSIS of its orbit
    assert o["final_time"] > orr["final_time"] + 4.0    # the proper ellipse needs ~33 m/s more


def test_dcost_both_ways_and_what_it_explains(golden):
    """a12, LO:99: DCOST = 1e-5 as an l1 movement penalty with slack pairs, switched on and off.  Measured against
    Numerical_results.png: it moves t_f by +1.5e-3 s towards the golden value (20 % of the 7.7e-3 s gap), removes 40 % of the
    control's total variation, and does NOT explain the 1.1e-3 gap in the final ydoubledot (it widens it)."""
    P, G = Params(), golden["current"]
    out = {}
    for dc in (0.0, 1e-5):
        nlp = GeneralNLP(P, ((199, "burn"),), 0, dcost=dc)
        v, _, info = solve_ip(nlp, tol=1e-9, max_iter=400)
        assert info["status"] == "converged"
        o = nlp.outputs(v)
        out[dc] = (o["final_time"], np.abs(np.diff(o["angledoubledot"])).sum(), o["final_ydoubledot"], nlp, v)
        if dc:       # the slack pairs reproduce |u_k - u_{k-1}| at the solution, up to the barrier's mu / multiplier = 1e-10 / 1e-5
            U = np.concatenate([[0.0], v[nlp.ucol]])
            assert np.abs(v[nlp.ip] + v[nlp.in_] - np.abs(np.diff(U))).max() < 5e-5
    t0, tv0, ay0 = out[0.0][:3]
    t1, tv1, ay1 = out[1e-5][:3]
    assert 1.0e-3 < t1 - t0 < 2.0e-3                      # +1.5e-3 s
    assert abs(t0 - G["final_time"]) < 8e-3 and abs(t1 - G["final_time"]) < abs(t0 - G["final_time"])
    assert tv1 < 0.7 * tv0
    assert abs(ay1 - G["final_ydoubledot"]) > abs(ay0 - G["final_ydoubledot"])     # not the explanation of that gap
    assert (t1 - t0) / 470.0 <= 1e-5 * tv0                # SURVEY 7.3.3's bound: the pull on tf is at most dcost * sum|du|


def test_final_acceleration_gap_is_a_flat_direction_of_the_objective(golden):
    """Weak point of the parity table: the restatement's final ydoubledot is 1.1e-3 (relative) from Numerical_results.png
    while every other printed number is within 1e-4.  Cause, measured: the final thrust angle sits on the singular arc
    and is barely determined by the objective.  Pinning the last node's angle to the value the golden accelerations imply
    (88.5003 deg; the free optimum has 88.5263 deg) costs 2.3e-7 s of ascent time (5e-10 relative, far below the
    reference's own OTOL = RTOL = 1e-3, LO:31-32) and brings BOTH accelerations within 3e-5 of the golden values."""
    from scipy.optimize import brentq
    P, G = Params(), golden["current"]
    d = P.derived(); S = d["S"]
    m_f = d["beta"] * G["final_time"]
    a_g = brentq(lambda a: accel(G["final_x"] / S, G["final_y"] / S, a, m_f, P)[1] * S - G["final_ydoubledot"], 0.4, 0.6)
    assert abs(3 * a_g * 180 / np.pi - 88.5003) < 1e-3           # final control angle 88.5003 deg
    free = GeneralNLP(P, ((199, "burn"),), 0)
    vf, _, i1 = solve_ip(free, tol=1e-10, max_iter=400)
    pinned = GeneralNLP(P, ((199, "burn"),), 0)
    ia = pinned.col[-1] + 4
    pinned.lb[ia], pinned.ub[ia] = a_g - 1e-6, a_g + 1e-6
    vp, _, i2 = solve_ip(pinned, tol=1e-10, max_iter=400)
    assert i1["status"] == i2["status"] == "converged"
    of, op = free.outputs(vf), pinned.outputs(vp)
    assert abs(of["final_ydoubledot"] / G["final_ydoubledot"] - 1) > 1e-3           # the gap of the free optimum
    assert 0 <= op["final_time"] - of["final_time"] < 1e-6                             # ... costs nothing to close
    assert abs(op["final_ydoubledot"] / G["final_ydoubledot"] - 1) < 1e-4
    assert abs(op["final_xdoubledot"] / G["final_xdoubledot"] - 1) < 1e-4


def test_dcost_fixture_is_what_the_generator_produces():
    """tests/golden/dcost_fixtures.json (the anchors of ascent_opts.move_penalty on the GPU) against a fresh solve of its smallest
    case by the numpy generic-LU oracle; every case: the penalty raises t_f by at most dcost * (total variation without it) and
    lowers the control's total variation."""
    import json, os
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dcost_fixtures.json")))
    c = [c for c in fx["cases"] if c["nt"] == 60 and c["scheme"] == 0 and c["dcost"] == 1e-3][0]
    nlp = GeneralNLP(Params(), ((59, "burn"),), 0, dcost=1e-3)
    v, _, info = solve_ip(nlp, tol=1e-10, max_iter=600)
    assert info["status"] == "converged"
    assert abs(float(v[nlp.itf]) - c["on"]["tf"]) < 1e-10 and np.abs(v[nlp.ucol] - np.array(c["on"]["u"])).max() < 1e-6
    assert {(k["nt"], k["scheme"]) for k in fx["cases"]} >= {(60, 0), (60, 1), (41, 2), (200, 0)}
    for k in fx["cases"]:
        assert 0.0 < k["on"]["tf"] - k["off"]["tf"] <= k["dcost"] * k["off"]["total_variation"]
        assert k["on"]["total_variation"] < k["off"]["total_variation"]


def test_terminal_ellipse_free_burnout_anywhere_on_the_target_ellipse():
    """terminal "ellipse" of the generalised oracle (= ascent_opts.terminal 2): burnout anywhere on the (r_peri, r_apo) ellipse --
    angular momentum >= and specific energy <= those of the ellipse, no r.v = 0 -- BASELINE config 5's burn--coast problem
    with the coast arc eliminated exactly.  Its hand-written terminal Jacobian / Hessian against central differences; the
    converged burnout orbit is the target ellipse to 1 mm with both conditions active, burnout a few metres off the periapsis,
    and the burn is never longer than the periapsis insertion's (terminal "periapsis" = ascent_opts.terminal 1)."""
    P = Params()
    g = GeneralNLP(P, ((11, "burn"),), 0, terminal="ellipse")
    assert g.m == 7 * 11 + 2
    rng = np.random.default_rng(11)
    v = g.initial_guess() + 0.01 * rng.standard_normal(g.n)
    lam = rng.standard_normal(g.m)
    J, H = g.jacobian(v).toarray(), g.hessian(v, lam).toarray()
    eps = 1e-6
    for col in range(g.col[-1], g.col[-1] + 4):          # the last node's x, y, xdot, ydot: where the terminal rows live
        e = np.zeros(g.n); e[col] = eps
        assert np.abs((g.constraints(v + e) - g.constraints(v - e)) / (2 * eps) - J[:, col]).max() < 1e-7
        assert np.abs((g.jacobian(v + e).T @ lam - g.jacobian(v - e).T @ lam) / (2 * eps) - H[:, col]).max() < 1e-6
    res = {}
    for term in ("periapsis", "ellipse"):
        nlp = GeneralNLP(P, ((59, "burn"),), 0, terminal=term)
        v, lam, info = solve_ip(nlp, tol=1e-10, max_iter=500)
        assert info["status"] == "converged"
        o = nlp.outputs(v)
        peri, apo = kepler_elements(P, o["x"][-1], o["y"][-1], o["xdot"][-1], o["ydot"][-1])
        assert abs(peri - P.r_peri) < 1e-3 and abs(apo - P.r_apo) < 1e-3
        res[term] = (o["final_time"], lam[-2:], v[nlp.is1], v[nlp.is2])
    assert 0.0 < res["periapsis"][0] - res["ellipse"][0] < 5e-3                    # 0.3 ms less burn
    assert np.all(res["ellipse"][1] < -1e-3) and res["ellipse"][2] < 1e-8 and res["ellipse"][3] < 1e-8      # both conditions active
