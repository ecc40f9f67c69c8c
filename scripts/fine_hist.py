"""Iterations per grid level of the config-3 sweep (default nested iteration): histograms, and how they sit in wavefronts of four.
MP=1: with the move penalty (the reference's DCOST = 1e-5)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()
mp = os.environ.get("MP", "0") == "1"
S[:, 15] = 1e-5
tot = A.solve_batch(S, 200, tol=1e-9, want_traj=False, move_penalty=mp).iters.astype(int)
upto60 = A.solve_batch(S, 60, tol=1e-3, want_traj=False, move_penalty=mp).iters.astype(int)
upto17 = A.solve_batch(S, 17, tol=1e-3, want_traj=False, coarse_nodes=-1, move_penalty=mp).iters.astype(int)
for name, it in (("17-node level", upto17), ("60-node level", upto60 - upto17), ("200-node level", tot - upto60)):
    w = it.reshape(-1, 4).max(axis=1)
    print(f"{name}: hist {np.bincount(it)[it.min():].tolist()} from {it.min()}; mean {it.mean():.2f} max {it.max()}; wavefront maxima: mean {w.mean():.2f} max {w.max()}")
fine = tot - upto60
print("fine-level iterations over the 64 x 64 grid (rows: Isp, every 8th; columns: dry mass, every 4th):")
print(fine.reshape(64, 64)[::8, ::4])
# what a single launch over all levels could gain: sum over levels of the slowest wavefront vs the slowest sum
cost = {17: 1.0 * 16, 60: 1.0 * 64, 200: 1.0 * 208}      # node-rounds per round (chunks x 16)
lv = {17: upto17, 60: upto60 - upto17, 200: tot - upto60}
wave = {n: (it + 1).reshape(-1, 4).max(axis=1) * cost[n] for n, it in lv.items()}
a = sum(w.max() for w in wave.values()); b = (wave[17] + wave[60] + wave[200]).max(); m = (wave[17] + wave[60] + wave[200]).mean()
print(f"sum of per-level maxima {a:.0f}, maximum of per-wavefront sums {b:.0f} ({100 * (1 - b / a):.1f} % less), mean wavefront {m:.0f}")
