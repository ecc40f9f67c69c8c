#!/usr/bin/env python3
"""Generates tests/golden/sweep_corners.json: independent anchors away from the nominal problem.

The numpy oracle (oracle/ascent_numpy.py: generic sparse LU on the full KKT matrix, no stage structure) solves the
four corners of the BASELINE config-3 box (Isp x dry mass) and the sixteen corners of the config-4 box (x target
apoapsis x angular-acceleration cap) from its own cold start at tol 1e-10.  The C oracle (CPU test) and the HIP
path (-m gpu) are asserted against these numbers, so a modelling error shared by the two stage-structured
implementations away from nominal cannot pass unnoticed.  Run on the CPU box:  python scripts/make_sweep_corners.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ascent_numpy import AscentNLP, Params, solve_ip  # noqa: E402
from lunar_module_ascent_trajectory_optimiser_amd.params import PARAM_FIELDS, sweep_config4, sweep_isp_drymass  # noqa: E402


def solve_row(row, nt=200):
    P = Params(**{f: float(v) for f, v in zip(PARAM_FIELDS, row)})
    nlp = AscentNLP(P, nt, 0)
    v, lam, info = solve_ip(nlp, tol=1e-10, max_iter=500)
    assert info["status"] == "converged", info
    o = nlp.outputs(v)
    return dict(params={f: float(v_) for f, v_ in zip(PARAM_FIELDS, row)}, tf=float(o["tf"]),
                final_x=float(o["x"][-1]), final_y=float(o["y"][-1]), final_xdot=float(o["xdot"][-1]),
                final_ydot=float(o["ydot"][-1]), final_angle=float(o["angle"][-1]), final_mass=float(o["mass"][-1]),
                max_angle=float(o["angle"].max()), iters=int(info["iters"]))


def main():
    out = {"_comment": "numpy generic-LU oracle (oracle/ascent_numpy.py), nt=200, backward Euler, tol 1e-10, cold start; "
                       "made by scripts/make_sweep_corners.py; scaled units as the reference's GEKKO variables",
           "nt": 200, "config3": [], "config4": []}
    g3 = sweep_isp_drymass(2, 2)                               # the 4 corners of the 64x64 box
    for row in g3:
        out["config3"].append(solve_row(row))
        print("config3", out["config3"][-1]["tf"], out["config3"][-1]["iters"], flush=True)
    g4 = sweep_config4(2, 2, 2, 2)                             # the 16 corners of the 64x64x8x8 box
    for row in g4:
        out["config4"].append(solve_row(row))
        print("config4", out["config4"][-1]["tf"], out["config4"][-1]["iters"], flush=True)
    path = os.path.join(ROOT, "tests", "golden", "sweep_corners.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
