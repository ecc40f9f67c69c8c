"""Generates tests/golden/dcost_fixtures.json: solutions of the ascent NLP WITH the reference's MV move penalty (LO:99, DCOST as an l1
term with a slack pair per step) from the numpy generic-LU oracle (oracle/ascent_general.py: sympy-free for schemes 0/1, no stage
structure) -- the independent anchor of ascent_opts.move_penalty = 1 on the GPU.  python scripts/make_dcost_fixtures.py"""
import json, os, sys, dataclasses
sys.path.insert(0, ".")
import numpy as np
from oracle.ascent_numpy import Params, solve_ip
from oracle.ascent_general import GeneralNLP
from oracle.c_oracle import PARAM_FIELDS

cases = []
base = Params()
for nt, scheme, dc, prm in ((60, 0, 1e-5, base), (60, 0, 1e-3, base), (60, 1, 1e-4, base), (41, 2, 1e-4, base), (200, 0, 1e-5, base), (200, 0, 1e-3, base),
                            (200, 0, 1e-5, dataclasses.replace(base, Ft=base.Ft * 1.04, M0=base.M0 * 0.98))):
    out = {}
    for d in (0.0, dc):
        nlp = GeneralNLP(prm, ((nt - 1, "burn"),), scheme, dcost=d)
        v, _, info = solve_ip(nlp, tol=1e-10, max_iter=600)
        assert info["status"] == "converged", (nt, scheme, d, info)
        o = nlp.outputs(v)
        U = np.concatenate([[0.0], v[nlp.ucol]])
        out[d] = dict(tf=float(v[nlp.itf] if np.ndim(v[nlp.itf]) == 0 else np.ravel(v[nlp.itf])[0]), total_variation=float(np.abs(np.diff(U)).sum()),
                      u=[float(x) for x in v[nlp.ucol]], final_time=float(o["final_time"]))
    p = {f: float(getattr(prm, f)) for f in PARAM_FIELDS}
    p["dcost"] = dc
    cases.append(dict(nt=nt, scheme=scheme, dcost=dc, params=p, off=out[0.0], on=out[dc]))
    print(nt, scheme, dc, "tf off/on", out[0.0]["tf"], out[dc]["tf"], "TV off/on", out[0.0]["total_variation"], out[dc]["total_variation"], flush=True)
json.dump(dict(generator="scripts/make_dcost_fixtures.py (numpy generic-LU oracle, tol 1e-10)", cases=cases),
          open(os.path.join("tests", "golden", "dcost_fixtures.json"), "w"), indent=1)
