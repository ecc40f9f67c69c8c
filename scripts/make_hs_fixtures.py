#!/usr/bin/env python3
"""Generates tests/golden/hs_fixtures.json: solutions of the Hermite-Simpson (scheme 2) transcription and of the
"ellipse proper" terminal constraints by the generalised numpy oracle (oracle/ascent_general.py: sympy-generated
derivatives, generic sparse LU, no stage structure).  The reference has neither (its NODES=2 is backward Euler and its
target speed is LO:72-78's mean-radius circular speed), so these rows are parity-unpinned by the reference; what pins
them is (i) this independent implementation and (ii) mesh refinement towards the trapezoid / Richardson limit of the
reference scheme (435.22 s).   Run on the CPU box:  python scripts/make_hs_fixtures.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ascent_general import GeneralNLP, kepler_elements  # noqa: E402
from oracle.ascent_numpy import Params, solve_ip  # noqa: E402


def solve(P, nt, scheme, terminal, v0=None, mu0=0.1):
    nlp = GeneralNLP(P, ((nt - 1, "burn"),), scheme, terminal=terminal)
    t = time.time()
    v, lam, info = solve_ip(nlp, v0=v0, tol=1e-10, max_iter=500, mu0=mu0)
    assert info["status"] == "converged", info
    o = nlp.outputs(v)
    peri, apo = kepler_elements(P, o["x"][-1], o["y"][-1], o["xdot"][-1], o["ydot"][-1])
    rec = dict(nt=nt, scheme=scheme, terminal=terminal, tf=float(o["tf"]), final_time=float(o["final_time"]),
               final_state=[float(o[k][-1]) for k in ("x", "y", "xdot", "ydot", "angle", "angledot", "mass")],
               orbit_periapsis_alt_m=float(peri), orbit_apoapsis_alt_m=float(apo), iters=int(info["iters"]),
               # every 10th node of the state trajectory (the control on the singular arc is only weakly determined)
               x=[float(a) for a in o["x"][::max(1, (nt - 1) // 20)]], y=[float(a) for a in o["y"][::max(1, (nt - 1) // 20)]],
               angle=[float(a) for a in o["angle"][::max(1, (nt - 1) // 20)]])
    print(f"nt={nt} scheme={scheme} terminal={terminal}: t_f={rec['final_time']:.6f} s, orbit {peri/1e3:.4f} x {apo/1e3:.4f} km, "
          f"{info['iters']} iterations, {time.time()-t:.1f} s", flush=True)
    return rec, nlp, v


def main():
    P = Params()
    out = {"_comment": "generalised numpy oracle (oracle/ascent_general.py), tol 1e-10, cold start; made by "
                       "scripts/make_hs_fixtures.py; scaled units as the reference's GEKKO variables", "cases": []}
    # terminal "ellipse" = ascent_opts.terminal 2: burnout anywhere on the (r_peri, r_apo) ellipse (angular momentum and energy)
    for nt, scheme, terminal in ((50, 2, "reference"), (200, 2, "reference"), (200, 0, "periapsis"), (200, 2, "periapsis"),
                                 (400, 2, "reference"), (60, 0, "ellipse"), (200, 0, "ellipse"), (200, 2, "ellipse")):
        out["cases"].append(solve(P, nt, scheme, terminal)[0])
    if "--n2000" in sys.argv:       # ~10 minutes each: BASELINE config 5's grid
        out["cases"].append(solve(P, 2000, 2, "periapsis")[0])
    if "--n2000-ellipse" in sys.argv:
        out["cases"].append(solve(P, 2000, 2, "ellipse")[0])
    path = os.path.join(ROOT, "tests", "golden", "hs_fixtures.json")
    if os.path.exists(path):      # keep the N=2000 cases made earlier (unless they are being made again)
        old = json.load(open(path))
        have = {(c["nt"], c["scheme"], c["terminal"]) for c in out["cases"]}
        out["cases"] += [c for c in old["cases"] if c["nt"] == 2000 and (c["nt"], c["scheme"], c["terminal"]) not in have]
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
