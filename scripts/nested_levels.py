"""Developer experiment (CPU, C restatement under the test tree): grid-level policies of the nested iteration.
Cost = sum over levels of iterations x nodes, in units of one iteration on the finest grid.

    python scripts/nested_levels.py [n_nodes] [scheme] [tol]"""
import json, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle as O

NT = 201 if len(sys.argv) < 2 else int(sys.argv[1])
SCHEME = 0 if len(sys.argv) < 3 else int(sys.argv[2])
TOL = 1e-9 if len(sys.argv) < 4 else float(sys.argv[3])


def run(S, levels, mu0s, ctol=1e-3, tol=TOL):
    its = np.zeros((len(levels), len(S)))
    r = O.solve_batch(S, levels[0], 300, max(tol, ctol) if len(levels) > 1 else tol, want_blob=True, coarse_nodes=-1, scheme=SCHEME)
    its[0] = r["iters"]
    ok = r["status"] == 0
    for li in range(1, len(levels)):
        g = np.stack([O.prolong(b, levels[li - 1], levels[li]) for b in r["blob"]])
        last = li == len(levels) - 1
        r = O.solve_batch(S, levels[li], 300, tol if last else max(tol, ctol), guess_blob=g, want_blob=True, warm_start=2, mu_init=mu0s[li - 1], scheme=SCHEME)
        its[li] = r["iters"]
        ok &= r["status"] == 0
    cost = sum(its[i].mean() * (levels[i] - 1) for i in range(len(levels))) / (NT - 1)
    worst = sum(its[i] * (levels[i] - 1) for i in range(len(levels))).max() / (NT - 1)
    return its.mean(axis=1), its.max(axis=1), cost, worst, ok.sum(), r["tf"]


def old_levels(nt):
    lv = [nt]
    while lv[-1] >= 64:
        lv.append(max(14, (lv[-1] + 5) // 11))
    return lv[::-1]


def new_levels(nt, num=3, den=10, mn=40):
    lv = [nt]
    while lv[-1] >= mn:
        c = max(14, (num * lv[-1] + 5) // den)
        if c > 17 and 1 <= (c - 1) % 16 <= 3:
            c -= (c - 1) % 16
        lv.append(c)
    return lv[::-1]


if __name__ == "__main__":
    sets = {"config3 sample": A.sweep_isp_drymass()[::67][:48], "config4 sample": A.sweep_config4()[::5471][:48]}
    cj = json.load(open("tests/golden/sweep_corners.json"))
    for name, S in sets.items():
        lo, ln = old_levels(NT), new_levels(NT)
        print(name, "old", lo, "new", ln)
        for lv, mus in [(lo, [1e-5] * 8), (ln, [1e-7] + [1e-8] * 8), (ln, [1e-6] + [1e-8] * 8), (ln, [1e-5] + [1e-7] * 8), (ln, [1e-6] + [1e-7] * 8)]:
            m, mx, cost, worst, nok, tf = run(S, lv, mus[:len(lv) - 1])
            print(f"  {str(lv):26s} mu0 {str(mus[:len(lv) - 1]):26s}: mean its {np.round(m, 2)} max {mx} cost {cost:.2f} worst {worst:.2f} converged {nok}/{len(S)}", flush=True)
