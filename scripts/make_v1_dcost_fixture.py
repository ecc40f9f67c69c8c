#!/usr/bin/env python3
"""Adds the v1 script's `angle.DCOST = 1e-5` (PDF p26) cases to tests/golden/dcost_fixtures.json: the hand-written numpy NLP of the v1
formulation (oracle/ascent_numpy.py: 5 states + the angle as the control) wrapped in MovePenaltyNLP (slack pairs and movement
equations as explicit unknowns and rows), solved by the generic sparse-LU interior point at tol 1e-10.  No stage structure, no
embedding into the 7-state layout, nothing reduced -- the independent anchor of ascent_opts.move_penalty with formulation 1.
    python scripts/make_v1_dcost_fixture.py"""
import dataclasses, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ascent_numpy import AscentNLP, MovePenaltyNLP, solve_ip, v1_params  # noqa: E402

path = os.path.join(ROOT, "tests", "golden", "dcost_fixtures.json")
fx = json.load(open(path))
fx["v1_cases"] = []
for nt, dcost in ((60, 1e-5), (200, 1e-5), (200, 1e-3)):
    P = dataclasses.replace(v1_params(), dcost=dcost)
    base = AscentNLP(P, nt, 1)
    nlp = MovePenaltyNLP(base, np.arange(nt - 1) * base.nw + base.ia, dcost, 0.0)
    v, lam, info = solve_ip(nlp, tol=1e-10, max_iter=800)
    v0, _, i0 = solve_ip(base, tol=1e-10, max_iter=800)
    assert info["status"] == "converged" and i0["status"] == "converged"
    o, o0 = nlp.outputs(v), base.outputs(v0)
    ang, ang0 = np.asarray(o["angle"]), np.asarray(o0["angle"])
    rec = dict(nt=nt, dcost=dcost, params=dataclasses.asdict(P),
               on=dict(tf=float(v[nlp.itf]), total_variation=float(np.abs(np.diff(ang)).sum()), angle=[float(a) for a in ang[1:]]),
               off=dict(tf=float(v0[base.itf]), total_variation=float(np.abs(np.diff(ang0)).sum())))
    print(nt, dcost, info["iters"], rec["on"]["tf"], rec["off"]["tf"], rec["on"]["total_variation"], rec["off"]["total_variation"])
    fx["v1_cases"].append(rec)
json.dump(fx, open(path, "w"), indent=1)
