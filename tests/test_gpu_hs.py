"""GPU parity tests of the persistent Hermite-Simpson kernel (csrc/ascent_hs.hip: h_solve, scheme 2 in the layout of the
persistent kernel -- structured step Jacobians, the step's cross Hessian as a rank-4 term) against the generalised oracle's
generic sparse LU (sympy-generated derivatives), the independent fixtures of tests/golden/hs_fixtures.json, and the dense-block
path (csrc/ascent_dense.hip), the second HIP implementation of the same scheme."""
import json
import os

import numpy as np
import pytest

import lunar_module_ascent_trajectory_optimiser_amd as A
from test_gpu_dense import _interior_blobs, _fixtures

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nt,B", [(40, 4), (14, 3), (75, 2)])
def test_hs_persistent_newton_step_matches_generic_lu(coracle, nt, B):
    """One Newton step of the barrier problem through h_solve (probe round: node blocks -> structured backward sweep with the
    rank-4 midpoint term and two border columns -> forward and adjoint sweeps) against the generic sparse-LU solve of the full KKT
    matrix assembled from the sympy-generated derivatives, and against the dense-block path's step; with and without primal
    regularisation.  Grids that end inside a chunk, span several chunks (12 nodes per chunk and NLP here: four NLPs ...) -- and one
    NLP per wavefront (48-node chunks), which the library takes for batches <= 1024."""
    from conftest import generic_lu_newton_step, params_of_row
    K = nt - 1
    S = A.sweep_isp_drymass(2, 2)[:B]
    blobs = _interior_blobs(coracle, S, nt, 2)
    mu = np.array([0.1, 0.02, 1e-3, 0.05])[:B]; dw = np.array([0.0, 1e-4, 1e-2, 1.0])[:B]
    steps = {}
    for wide in ("1", "0"):
        os.environ["ASCENT_PERSIST_WIDE"] = wide
        try:
            steps[wide] = A.kkt_step(S, blobs, mu, dw, nt, path="persist", scheme=2)
        finally:
            del os.environ["ASCENT_PERSIST_WIDE"]
    os.environ["ASCENT_DENSE_NEWTON"] = "riccati"      # (the cyclic-reduction variant tests curvature, not inertia: it may accept what these refuse)
    try:
        dense, ind = A.kkt_step(S, blobs, mu, dw, nt, path="dense", scheme=2)
    finally:
        del os.environ["ASCENT_DENSE_NEWTON"]
    n_ok = 0
    for wide, (step, inertia) in steps.items():
        assert np.array_equal(inertia, ind)
        for b in range(B):
            if inertia[b]:
                continue
            n_ok += 1
            lu, _, _, _ = generic_lu_newton_step(params_of_row(S[b]), nt, np.ascontiguousarray(blobs[:, b]), mu[b], dw[b], 2)
            for lo, hi in ((0, 8 * K), (8 * K, 15 * K), (15 * K, 21 * K), (21 * K, 21 * K + 10)):
                sc = max(1.0, np.abs(lu[lo:hi]).max())
                assert np.abs(step[lo:hi, b] - lu[lo:hi]).max() <= 1e-8 * sc, (wide, b, lo)
                assert np.abs(step[lo:hi, b] - dense[lo:hi, b]).max() <= 1e-8 * sc, (wide, b, lo)
    assert n_ok >= 2 * (B - 1)


def test_hs_persistent_matches_independent_fixtures(monkeypatch):
    """tests/golden/hs_fixtures.json (generalised numpy oracle, scripts/make_hs_fixtures.py) through the persistent kernel at every
    size -- the default dispatch sends one NLP on a long grid to the dense blocks with cyclic reduction; ASCENT_SMALL_BATCH=off
    keeps it here --: t_f to 1e-9, final state to 1e-6; reference, periapsis and anywhere-on-the-ellipse terminal constraints."""
    monkeypatch.setenv("ASCENT_SMALL_BATCH", "off")
    n = 0
    for (nt, scheme, terminal), c in sorted(_fixtures().items()):
        if scheme != 2:
            continue
        term = {"reference": "reference", "periapsis": "ellipse", "ellipse": "ellipse_free"}[terminal]
        assert A.default_path(1, nt, scheme=2, terminal=term) == "persist"
        r = A.solve_batch(A.AscentParams(), nt, tol=1e-10, scheme=2, terminal=term, max_iter=500)
        assert r.status[0] == 0
        assert abs(r.tf[0] - c["tf"]) <= 1e-9, (nt, terminal, r.tf[0], c["tf"])
        fs = np.array([r.traj[f][-1, 0] for f in (0, 1, 2, 3, 6, 7, 9)])
        assert np.abs(fs - np.array(c["final_state"])).max() <= 1e-6
        n += 1
    assert n >= 5


def test_hs_persistent_equals_dense_blocks(monkeypatch):
    """Two HIP implementations of Hermite-Simpson: the persistent kernel (one NLP per wavefront up to 1024 NLPs, four above) and the
    dense-block path.  Same interior point, same regularisation rule: identical iteration counts and t_f to rounding on sweeps over
    grid sizes that exercise the chunk padding, the three terminal conditions, a batch beyond 1024 and ragged last wavefronts."""
    cases = [(200, 20, "reference"), (37, 7, "reference"), (61, 5, "ellipse"), (150, 9, "ellipse_free"), (200, 6, "ellipse_free"), (49, 1030, "reference")]
    for nt, B, term in cases:
        S = A.sweep_config4()[:: 262144 // B][:B] if B > 64 else np.vstack([A.AscentParams().as_row()[None], A.sweep_isp_drymass(5, 4)])[:B]
        res = {}
        for path in ("persist", "dense"):
            monkeypatch.setenv("ASCENT_PIPELINE", path)
            monkeypatch.setenv("ASCENT_DENSE_NEWTON", "riccati")
            assert A.default_path(B, nt, scheme=2, terminal=term) == path
            res[path] = A.solve_batch(S, nt, tol=1e-9, scheme=2, terminal=term, max_iter=500)
        p, d = res["persist"], res["dense"]
        assert np.all(p.status == 0) and np.all(d.status == 0), (nt, B, term)
        # (terminal 2's two conditions are nearly dependent -- multipliers of -3 and -3000 --: the two paths' roundings part ways in the
        #  inertia corrections of the coarse grids and arrive a few iterations apart, 35-49 against 37-51 on the 150-node sweep)
        di = np.abs(p.iters.astype(int) - d.iters)
        if B > 64:      # (the corners of config 4's box on a coarse grid go through inertia corrections: a few of a thousand part ways there)
            assert di.max() <= 4 and (di == 0).mean() >= 0.97, (nt, B, term, di.max(), (di == 0).mean())
        else:
            assert di.max() <= (8 if term == "ellipse_free" else 0), (nt, B, term)
        assert np.abs(p.tf - d.tf).max() <= (2e-9 if term == "ellipse_free" or B > 64 else 1e-11), (nt, B, term)
        # (x y xdot ydot | accelerations, angle | angledot, the control -- around the junction of the saturated and the singular arc the
        #  control is only weakly determined by a KKT point, with terminal 2 the whole singular arc is: 0.06 of its range between two
        #  solutions whose t_f agree to 1e-9 -- | mass)
        tols = (1e-5, 1e-5, 1e-5, 1e-5, 1e-4, 1e-4, 1e-4, None, None, 1e-5) if term == "ellipse_free" or B > 64 else (1e-6, 1e-6, 1e-6, 1e-6, 1e-5, 1e-5, 1e-5, 1e-3, 1e-3, 1e-6)
        for f, tol in enumerate(tols):
            if tol is not None:
                assert np.abs(p.traj[f] - d.traj[f]).max() <= tol * max(1.0, np.abs(d.traj[f]).max()), (nt, B, term, f)


def test_hs_persistent_config5_batch():
    """BASELINE.json configs[4] as a batch through the default dispatch (the persistent Hermite-Simpson kernel): 256 NLPs at N = 2000
    with the periapsis condition and with burnout anywhere on the ellipse: all converge, the burnout orbit is the target ellipse to 1 m,
    terminal 2 never burns longer than terminal 1."""
    S = np.vstack([A.AscentParams().as_row()[None], A.sweep_isp_drymass(16, 16)])[:256]
    assert A.default_path(256, 2000, scheme=2) == "persist"
    r1 = A.solve_batch(S, 2000, tol=1e-9, scheme=2, terminal="ellipse", max_iter=500)
    r2 = A.solve_batch(S, 2000, tol=1e-9, scheme=2, terminal="ellipse_free", max_iter=500)
    for r in (r1, r2):
        assert np.all(r.status == 0)
        o = r.orbit()
        assert np.abs(o["periapsis_alt"] - 17703.0).max() < 1.0 and np.abs(o["apoapsis_alt"] - 88615.0).max() < 1.0
    assert np.all(r2.tf <= r1.tf + 1e-9)
    by = _fixtures()
    assert abs(r1.tf[0] - by[(2000, 2, "periapsis")]["tf"]) < 1e-7 and abs(r2.tf[0] - by[(2000, 2, "ellipse")]["tf"]) < 1e-7


def test_hs_persistent_unusual_grid_sizes(monkeypatch):
    """Grids from three nodes up, ending one short of / exactly on / one past the chunk boundaries of both kernel forms (12 and 48 nodes),
    single-grid and nested: the persistent Hermite-Simpson kernel and the dense-block path arrive at the same t_f in the same number of
    iterations (one more or less where the KKT error passes the tolerance within rounding of it)."""
    S = np.vstack([A.AscentParams().as_row()[None], A.sweep_isp_drymass(2, 2)])[:3]
    for nt in (3, 4, 5, 12, 13, 14, 25, 37, 48, 49, 50, 97, 98, 145):
        res = {}
        for path in ("persist", "dense"):
            monkeypatch.setenv("ASCENT_PIPELINE", path)
            monkeypatch.setenv("ASCENT_DENSE_NEWTON", "riccati")
            res[path] = A.solve_batch(S, nt, tol=1e-9, scheme=2, max_iter=500)
        p, d = res["persist"], res["dense"]
        assert np.array_equal(p.status, d.status), (nt, p.status, d.status)
        ok = p.status == 0
        assert ok.sum() >= 2 or nt < 5, (nt, p.status)         # (three- and four-node grids of the off-nominal problems need not have a solution)
        assert np.abs(p.iters[ok].astype(int) - d.iters[ok]).max(initial=0) <= 1, (nt, p.iters, d.iters)
        assert np.abs(p.tf[ok] - d.tf[ok]).max(initial=0.0) <= 1e-9, nt
