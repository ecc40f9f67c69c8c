#!/usr/bin/env python3
"""Why BASELINE config 5's "two-phase burn--coast" is not solved as a two-phase NLP with a free coast duration
(DESIGN.md section 7).  CPU only: oracle/ascent_general.py with phases = (burn, coast), the coast arc
discretised like the burn (thrust off, two-body gravity, its own free duration theta_2), arrival at the apoapsis of the
(r_peri, r_apo) ellipse.  Three observations, printed and kept in profiles/r03_two_phase_degeneracy.txt:

 1. the interior point (generic sparse LU, exact-inertia rule) on that NLP from the natural starts: iterations, status, the
    regularisation it needs;
 2. at the composed solution (burn to the periapsis of the ellipse = terminal 1, then the Kepler arc to its apoapsis) the
    reduced Hessian of the Lagrangian on the null space of the active constraints: its smallest eigenvalues against its
    largest -- the flat valley along which burn end point and coast length trade against each other;
 3. the same problem with the coast arc ELIMINATED exactly (two-body motion conserves angular momentum and energy):
    terminal "ellipse" = ascent_opts.terminal 2 -- well posed, converges, and gains 0.7 ms of burn over the periapsis insertion.

    python scripts/two_phase_degeneracy.py [n_burn n_coast]
"""
import math
import os
import sys
import time

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ascent_general import GeneralNLP, kepler_elements  # noqa: E402
from oracle.ascent_numpy import Params, solve_ip  # noqa: E402

K1 = int(sys.argv[1]) if len(sys.argv) > 1 else 60
K2 = int(sys.argv[2]) if len(sys.argv) > 2 else 40
P = Params()
lines = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    lines.append(s)


say(f"two-phase burn ({K1} steps) + coast ({K2} steps), backward Euler, nominal Apollo-11 parameters")
one = GeneralNLP(P, ((K1, "burn"),), 0, terminal="periapsis")
v1, lam1, info1 = solve_ip(one, tol=1e-10, max_iter=500)
o1 = one.outputs(v1)
say(f"[terminal 1] burn to the periapsis of the ellipse: {info1['status']} in {info1['iters']} iterations, t_f = {o1['final_time']:.6f} s")

two = GeneralNLP(P, ((K1, "burn"), (K2, "coast")), 0)
say(f"[two-phase] unknowns {two.n}, equalities {two.m}, terminal = {two.terminal} (arrival at the apoapsis, r.v = 0)")
# 1. the interior point from the natural starts
starts = {"built-in guess (straight-line burn + Kepler arc of the target ellipse)": two.initial_guess()}
vc = two.initial_guess()
burn_cols = (two.col[:K1, None] + np.arange(8)).ravel()
vc[burn_cols] = v1[(one.col[:, None] + np.arange(8)).ravel()]
vc[two.ith[0]] = v1[one.ith[0]]
starts["composed: the terminal-1 burn + the Kepler arc from its periapsis"] = vc
for name, v0 in starts.items():
    for inertia in ("curvature", "exact"):
        t = time.time()
        try:
            v, lam, info = solve_ip(two, v0=v0, tol=1e-8, max_iter=300, inertia=inertia)
            o = two.outputs(v)
            peri, apo = kepler_elements(P, o["x"][K1], o["y"][K1], o["xdot"][K1], o["ydot"][K1])
            say(f"  start: {name}; inertia rule {inertia}: {info['status']} after {info['iters']} iterations, {info.get('reg', '?')} regularised "
                f"factorisations, burn {o['theta'][0] * P.T_scale:.4f} s, coast {o['theta'][1] * P.T_scale:.1f} s, burnout orbit {peri / 1e3:.3f} x {apo / 1e3:.3f} km, {time.time() - t:.1f} s")
        except Exception as e:  # noqa: BLE001
            say(f"  start: {name}; inertia rule {inertia}: failed with {type(e).__name__}: {e}")

# 2. reduced Hessian at the composed point (multipliers by least squares on the stationarity rows)
v = vc.copy()
c = two.constraints(v)
v[two.is1], v[two.is2] = max(c[-2] + v[two.is1], 0.0), max(c[-1] + v[two.is2], 0.0)
c = two.constraints(v)
J = two.jacobian(v).toarray()
g = two.grad_objective(v)
active = np.concatenate([np.flatnonzero(np.isfinite(two.lb) & (v - two.lb < 1e-7)), np.flatnonzero(np.isfinite(two.ub) & (two.ub - v < 1e-7))])
A = np.vstack([J, np.eye(two.n)[active]]) if len(active) else J
mult = np.linalg.lstsq(A.T, -g, rcond=None)[0]
lam = mult[: two.m]
W = two.hessian(v, lam).toarray()
Z = sla.null_space(A, rcond=1e-12)
Hr = Z.T @ W @ Z
ev = np.linalg.eigvalsh(0.5 * (Hr + Hr.T))
say(f"[composed point] |c|_inf = {np.abs(c).max():.2e}, stationarity residual {np.abs(g + A.T @ mult).max():.2e}, {len(active)} active bounds, null space dimension {Z.shape[1]}")
say(f"  reduced Hessian eigenvalues: smallest {ev[:6]}, largest {ev[-1]:.3e}; condition {abs(ev[-1]) / max(abs(ev).min(), 1e-300):.2e}; negative: {(ev < -1e-12 * abs(ev[-1])).sum()}")
sv = np.linalg.svd(A, compute_uv=False)
say(f"  constraint Jacobian singular values: smallest {sv[-3:]}, largest {sv[0]:.3e}")

# 3. the coast arc eliminated exactly
ell = GeneralNLP(P, ((K1, "burn"),), 0, terminal="ellipse")
v3, lam3, info3 = solve_ip(ell, tol=1e-10, max_iter=500)
o3 = ell.outputs(v3)
peri, apo = kepler_elements(P, o3["x"][-1], o3["y"][-1], o3["xdot"][-1], o3["ydot"][-1])
say(f"[terminal 2] burnout anywhere on the ellipse (angular momentum and energy): {info3['status']} in {info3['iters']} iterations, t_f = {o3['final_time']:.6f} s "
    f"({(o1['final_time'] - o3['final_time']) * 1e3:.3f} ms less than terminal 1), burnout orbit {peri / 1e3:.4f} x {apo / 1e3:.4f} km, "
    f"burnout {math.hypot(o3['x'][-1], o3['y'][-1] + ell.d['rho0']) * ell.d['S'] - P.R0 - P.r_peri:.1f} m above the periapsis radius")
with open(os.path.join(ROOT, "profiles", "r03_two_phase_degeneracy.txt"), "w") as f:
    f.write("\n".join(lines) + "\n")
