"""The GEKKO-style problem-definition surface.
CPU tests inject the plain-C oracle as the batch solver (the HIP library has no CPU fallback); the
GPU test runs the example end to end on the HIP path and compares its prints with the golden vector."""
import importlib.util
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _example():
    spec = importlib.util.spec_from_file_location("apollo11_example", os.path.join(ROOT, "examples", "apollo11.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _oracle_solver(P, nt, tol, max_iter, formulation=0, **kw):
    # (the stand-in for the GPU on the CPU box: the plain-C restatement, with the script's DCOST as the l1 move penalty when the
    #  front door passes `move_penalty`)
    from oracle import c_oracle
    from lunar_module_ascent_trajectory_optimiser_amd.params import pack
    from lunar_module_ascent_trajectory_optimiser_amd.solver import BatchResult
    P = pack(P)
    r = c_oracle.solve_batch(P, nt, max_iter, tol, formulation=formulation, scheme=kw.get("scheme", 0),
                             move_penalty=bool(kw.get("move_penalty", False)))
    c_oracle.set_formulation(0); c_oracle.set_scheme(0)
    return BatchResult(P, nt, np.ascontiguousarray(np.moveaxis(r["traj"], 0, 2)), r["tf"], r["status"], r["iters"], None, 0.0)


def test_model_recognition_recovers_the_reference_parameters():
    import lunar_module_ascent_trajectory_optimiser_amd as A
    ex = _example()
    m, _, _ = ex.build()
    P, R, tf = m._extract()
    ref = A.AscentParams()
    for f in ("G", "M", "R0", "Ft", "M0", "ang_acc_max", "r_peri", "T_scale", "angle_ub", "tf_lb", "tf_ub"):
        assert getattr(P, f) == pytest.approx(getattr(ref, f), rel=1e-12), f
    assert P.r_apo == pytest.approx(ref.r_apo, rel=1e-9)                     # recovered from the target speed
    assert P.mdot / P.fuel_mass == pytest.approx(ref.mdot / ref.fuel_mass, rel=1e-12)
    assert P.dcost == 1e-5


def test_unrecognised_models_fail_loudly():
    from lunar_module_ascent_trajectory_optimiser_amd.gekko_shim import GEKKO, ModelNotRecognised
    ex = _example()
    m, v, _ = ex.build()
    m.options.NODES = 3                                  # a different collocation scheme
    with pytest.raises(ModelNotRecognised):
        m._extract()
    m, v, _ = ex.build()
    m._equations[7] = (v["ydd"] == 2.0 * m._equations[7].rhs)     # tampered dynamics
    with pytest.raises(ModelNotRecognised):
        m._extract()
    m = GEKKO()
    m.time = np.linspace(0, 1, 10)
    z = m.Var(name="z")
    m.Equation(z.dt() == -z)
    m.Minimize(z)
    with pytest.raises(ModelNotRecognised):
        m.solve(disp=False)


def test_example_prints_match_golden_with_oracle_solver(coracle, golden):
    ex = _example()
    m, v, v_ins = ex.build(solver=_oracle_solver)
    buf = io.StringIO()
    with redirect_stdout(buf):
        m.solve(disp=True)
        ft = ex.report(m, v, v_ins)
    g = golden["current"]
    assert abs(ft - g["final_time"]) <= 1e-4 * g["final_time"]
    out = buf.getvalue().splitlines()
    vals = {ln.rsplit(" ", 1)[0]: float(ln.rsplit(" ", 1)[1]) for ln in out if ln.startswith("final ")}
    assert abs(vals["final x"] - g["final_x"]) <= 1e-4 * abs(g["final_x"])
    assert abs(vals["final y"] - g["final_y"]) <= 1e-4 * abs(g["final_x"])
    assert abs(vals["final xdot"] - g["final_xdot"]) <= 1e-4 * 1655
    assert abs(vals["final xdoubledot"] - g["final_xdoubledot"]) <= 2e-3 * abs(g["final_xdoubledot"])
    assert any(ln.startswith("Optimal Solution (final time): ") for ln in out)
    assert float(out[[i for i, ln in enumerate(out) if ln.startswith("Optimal")][0] + 1]) == pytest.approx(g["periapsis_v"], rel=1e-9)
    assert len(v["x"].value) == 200 and len(v["tf"].value) == 200 and v["x"].value[0] == 0.0


def test_example_plots(tmp_path, coracle):
    ex = _example()
    m, v, v_ins = ex.build(nt=40, solver=_oracle_solver)
    m.solve(disp=False)
    ex.plots(m, v, str(tmp_path))
    for f in ("takeoff_contextualized.png", "Angle_vs_Time.png", "takeoff_trajectory.png"):
        assert os.path.getsize(tmp_path / f) > 10000


def test_reference_script_runs_unmodified_on_the_shim(coracle, golden, tmp_path, monkeypatch):
    """Drop-in check (CPU, only where /root/reference is present): the reference script, byte for byte,
    with `gekko` resolved to compat/gekko and the oracle standing in for the GPU.  Checks the surface
    (every call of Appendix D is accepted, prints and figures are produced), not parity."""
    script = "/root/reference/Launch_Optimiser.py"
    if not os.path.exists(script):
        pytest.skip("reference not present on this machine")
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from lunar_module_ascent_trajectory_optimiser_amd import gekko_shim
    monkeypatch.setattr(gekko_shim, "DEFAULT_SOLVER", _oracle_solver)
    monkeypatch.setattr(plt, "show", lambda *a, **k: None)
    monkeypatch.syspath_prepend(os.path.join(ROOT, "compat"))
    monkeypatch.chdir(tmp_path)
    for k in [k for k in sys.modules if k == "gekko" or k.startswith("gekko.")]:
        monkeypatch.delitem(sys.modules, k)
    buf = io.StringIO()
    with redirect_stdout(buf):
        exec(compile(open(script).read(), script, "exec"), {"__name__": "reference_script"})
    out = buf.getvalue()
    ft = float([ln for ln in out.splitlines() if ln.startswith("final time")][0].split()[-1])
    assert abs(ft - golden["current"]["final_time"]) <= 1e-4 * golden["current"]["final_time"]
    for f in ("takeoff_contextualized.png", "Angle_vs_Time.png", "takeoff_trajectory.png"):
        assert (tmp_path / f).exists()


def _build_v1(solver=None):
    """A v1-style declaration (PDF p26-28 as described in SURVEY.md B.2): the angle is the MV, circular target
    53108.4 m, mass_scalar 2576 with mflow 5.053/2376.  Written here from that description, not copied."""
    from lunar_module_ascent_trajectory_optimiser_amd.gekko_shim import GEKKO
    m = GEKKO(solver=solver)
    nt = 200
    m.time = np.linspace(0, 1, nt)
    m.options.NODES, m.options.IMODE, m.options.SOLVER, m.options.MAX_ITER = 2, 6, 3, 20000
    tf = m.FV(value=0, lb=0, ub=1); tf.STATUS = 1
    G, M, R0 = m.Const(6.674e-11, name="G"), m.Const(7.346e22, name="M"), m.Const(1738100, name="R0")
    Ft, M0 = m.Const(15346, name="Ft"), m.Const(4821, name="M0")
    S, mS = m.Const(53108.4, name="distance Scale"), m.Const(2576, name="mass Scale")
    mflow, T = 5.053 / 2376, 470
    v_orb = (6.674e-11 * 7.346e22 / (1738100 + 53108.4)) ** 0.5
    mass = m.Var(value=0, lb=0, ub=1, name="mass")
    y, ydot, ydd = m.Var(value=0, name="y"), m.Var(name="ydot"), m.Var(name="ydoubledot")
    x, xdot, xdd = m.Var(value=0, name="x"), m.Var(name="xdot"), m.Var(name="xdoubledot")
    angle = m.MV(name="angle", lb=0, ub=np.pi / 3); angle.STATUS = 1; angle.DCOST = 1e-5
    for var, rate in ((y, ydot), (ydot, ydd), (x, xdot), (xdot, xdd)):
        m.Equation(var.dt() == tf * rate * T)
    m.Equation(mass.dt() == mflow * T * tf)
    X, Y = x * S, y * S + R0
    r = (X ** 2 + Y ** 2) ** (1 / 2)
    m.Equation(ydd == ((Ft / ((M0 - mS * mass) * r)) * (Y * m.cos(3 * angle) + X * m.sin(3 * angle)) - Y * (G * M / r ** 3)) / S)
    m.Equation(xdd == ((Ft / ((M0 - mS * mass) * r)) * (X * m.cos(3 * angle) - Y * m.sin(3 * angle)) - X * (G * M / r ** 3)) / S)
    for v in (y, x, ydot, xdot, angle, mass):
        m.fix(v, pos=0, val=0)
    c1 = np.full(nt, S + R0 + 1); c1[-1] = 0
    c2 = np.zeros(nt); c2[-1] = 1
    p1, p2 = m.Param(value=c1), m.Param(value=c2)
    m.Equation(((y + R0 / S) ** 2 + x ** 2) ** (1 / 2) + p1 >= (R0 + S) / S)
    m.Equation(xdot ** 2 + ydot ** 2 >= (v_orb / S) ** 2 * p2)
    m.Equation((Y * (ydot * S) + X * (xdot * S)) * p2 == 0)
    m.Minimize(tf)
    return m, dict(tf=tf, x=x, y=y, xdot=xdot, ydot=ydot, xdd=xdd, ydd=ydd, angle=angle)


def _check_v1(v, golden):
    g, S = golden["v1"], 53108.4
    assert abs(v["tf"].value[0] - g["tf"]) <= 1e-4 * g["tf"]
    assert abs(v["tf"].value[0] * 470 - g["final_time"]) <= 1e-4 * g["final_time"]
    assert abs(-v["x"].value[-1] * S - g["final_x_flipped"]) <= 1e-4 * g["final_x_flipped"]
    assert abs(v["y"].value[-1] * S - g["final_y"]) <= 1e-4 * g["final_x_flipped"]
    assert abs(-v["xdot"].value[-1] * S - g["final_xdot_flipped"]) <= 1e-4 * g["final_xdot_flipped"]
    assert abs(v["ydd"].value[-1] * S - g["final_ydoubledot"]) <= 2e-3 * abs(g["final_ydoubledot"])
    ang = 3 * np.asarray(v["angle"].value) * 180 / np.pi
    assert 33 < ang[1] < 37 and 109 < ang[-1] < 113


def test_v1_style_script_on_the_shim_with_oracle_solver(coracle, golden):
    """The second golden vector (PDF p30) through the same front door."""
    m, v = _build_v1(solver=_oracle_solver)
    P, R, tf = m._extract()
    assert m._formulation == 1 and P.mass_scalar == 2576 and P.r_apo == pytest.approx(P.r_peri, rel=1e-9)
    m.solve(disp=False)
    _check_v1(v, golden)


@pytest.mark.gpu
def test_v1_style_script_end_to_end_on_gpu(golden):
    m, v = _build_v1()
    m.solve(disp=False)
    _check_v1(v, golden)


@pytest.mark.gpu
def test_example_end_to_end_on_gpu(golden):
    ex = _example()
    m, v, v_ins = ex.build()
    buf = io.StringIO()
    with redirect_stdout(buf):
        m.solve(disp=True)
        ft = ex.report(m, v, v_ins)
    assert abs(ft - golden["current"]["final_time"]) <= 1e-4 * golden["current"]["final_time"]
    assert "libascent (MI355X)" in buf.getvalue() and m.options.ITERATIONS > 5
    assert abs(3 * v["angle"].value[-1] * 180 / np.pi - golden["qualitative"]["angle_final_deg"]) < 0.2


def test_solver_options_of_the_script_are_honoured(golden):
    """m.options.OTOL / RTOL (Launch_Optimiser.py:31-32) bound the KKT tolerance from above but never loosen it beyond 1e-9 (the
    banner says so); DCOST (:99) is applied by default (weight convention and 'parity unpinned' stated in the banner, OBJFCNVAL =
    tf + the penalty) and reported, once, when switched off; NODES other than 2 is refused with a pointer to the
    ASCENT_SCHEME extension, which reaches the trapezoid scheme through the front door."""
    import warnings
    from lunar_module_ascent_trajectory_optimiser_amd.gekko_shim import GEKKO, ModelNotRecognised
    seen = {}

    def spy(P, nt, tol, max_iter, formulation=0, **kw):
        seen.clear(); seen.update(tol=tol, kw=kw)
        return _oracle_solver(P, nt, tol, max_iter, formulation, **kw)

    ex = _example()
    GEKKO._dcost_warned = False
    m, v, _ = ex.build(nt=60, solver=spy)
    m.options.OTOL = m.options.RTOL = 1e-3                       # the reference's values
    buf = io.StringIO()
    with redirect_stdout(buf), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m.solve(disp=True)
    assert seen["tol"] == 1e-9 and "never looser than 1e-9" in buf.getvalue()
    # the script's DCOST is part of its model: passed on as the move penalty by default, no warning ...
    assert seen["kw"].get("move_penalty") is True and "DCOST 1.0e-05: applied" in buf.getvalue() and "parity unpinned" in buf.getvalue()
    assert not any("DCOST" in str(x.message) for x in w)
    tv = np.abs(np.diff(m.result.field("angledoubledot")[:, 0])).sum()
    assert m.options.OBJFCNVAL == pytest.approx(v["tf"].value[0] + 1e-5 * tv, rel=1e-12) and m.options.OBJFCNVAL > v["tf"].value[0]
    # a solver hook without **kw gets only the options it declares
    m2, v2, _ = ex.build(nt=60, solver=lambda P, nt, tol, max_iter, formulation=0: _oracle_solver(P, nt, tol, max_iter, formulation))
    m2.solve(disp=False)
    assert abs(v2["tf"].value[0] - v["tf"].value[0]) < 1e-4
    # ... and reported, once, when it is switched off
    m.options.ASCENT_DCOST = 0
    buf = io.StringIO()
    with redirect_stdout(buf), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m.solve(disp=True)
    assert "move_penalty" not in seen["kw"] or not seen["kw"]["move_penalty"]
    assert any("DCOST" in str(x.message) for x in w) and "DCOST 1.0e-05: not applied" in buf.getvalue()
    m.options.OTOL = 1e-11
    m.solve(disp=False)
    assert seen["tol"] == 1e-11
    t_be = v["x"].value[-1]
    m.options.OTOL = 1e-6
    m.options.ASCENT_SCHEME = 1
    m.solve(disp=False)
    assert seen["kw"].get("scheme") == 1 and v["x"].value[-1] != t_be
    m.options.NODES = 3
    with pytest.raises(ModelNotRecognised, match="ASCENT_SCHEME"):
        m.solve(disp=False)
    m.options.NODES = 2
    m.options.ASCENT_DCOST = 1                                   # the move penalty reaches the solver as move_penalty=True
    m.solve(disp=False)
    assert seen["kw"].get("move_penalty") is True


@pytest.mark.gpu
def test_hermite_simpson_and_ellipse_through_the_front_door():
    """ASCENT_SCHEME = 2 / ASCENT_TERMINAL = 1 through the GEKKO-style surface on the HIP path: the high-order answer of the
    reference's problem (435.227 s at 200 nodes) and the insertion into the 17.7 x 88.6 km ellipse proper (440.844 s)."""
    ex = _example()
    m, v, _ = ex.build()
    m.options.ASCENT_DCOST = 0                                   # (the two anchors below are solutions without the move penalty)
    m.options.ASCENT_SCHEME = 2
    m.solve(disp=False)
    assert abs(m.options.OBJFCNVAL * 470.0 - 435.226762) < 1e-4
    m.options.ASCENT_TERMINAL = 1
    m.solve(disp=False)
    assert abs(m.options.OBJFCNVAL * 470.0 - 440.844369) < 1e-4


@pytest.mark.gpu
def test_dcost_through_the_front_door(golden):
    """The script's own `angledoubledot.DCOST = 1e-5` (Launch_Optimiser.py:99) is applied on the HIP path by default
    (m.options.ASCENT_DCOST = 1 -> ascent_opts.move_penalty; 0 switches it off).  t_f moves by +1.5e-3 s towards Numerical_results.png (20 % of the 7.7e-3 s gap, as the
    numpy oracle measured), the banner says "applied", no warning."""
    import json, os, warnings
    from lunar_module_ascent_trajectory_optimiser_amd.gekko_shim import GEKKO
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dcost_fixtures.json")))
    nominal = [c for c in fx["cases"] if c["nt"] == 200 and c["dcost"] == 1e-5][0]
    ex = _example()
    m, v, _ = ex.build()
    m.options.ASCENT_DCOST = 0
    m.solve(disp=False)
    t_off = m.options.OBJFCNVAL                                  # (no penalty: the objective is tf)
    assert t_off == v["tf"].value[0]
    GEKKO._dcost_warned = False
    m.options.ASCENT_DCOST = 1                                   # (the default)
    buf = io.StringIO()
    with redirect_stdout(buf), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m.solve(disp=True)
    t_on = v["tf"].value[0]
    tv = np.abs(np.diff(m.result.field("angledoubledot")[:, 0])).sum()
    assert m.options.OBJFCNVAL == pytest.approx(t_on + 1e-5 * tv, rel=1e-12)      # the objective that was minimised: tf + DCOST * sum|du|
    assert "DCOST 1.0e-05: applied" in buf.getvalue() and not any("DCOST" in str(x.message) for x in w)
    assert abs(t_on - nominal["on"]["tf"]) <= 2e-8 and abs(t_off - nominal["off"]["tf"]) <= 2e-8
    G = golden["current"]["final_time"]
    assert 1.0e-3 < (t_on - t_off) * 470.0 < 2.0e-3 and abs(t_on * 470.0 - G) < abs(t_off * 470.0 - G)
