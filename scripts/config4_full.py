"""BASELINE.json config 4 whole (262 144 NLPs) on ONE GPU, as one batch and as eight shards: convergence, iteration counts, rate."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
full = A.sweep_config4()
A.solve_batch(full[:4096], 200, want_traj=False)
t = time.time(); r = A.solve_batch(full, 200, want_traj=False); wall = time.time() - t
print(f"one batch of {len(full)}: status counts {np.bincount(r.status, minlength=4)}, iters {r.iters.min()}/{r.iters.mean():.2f}/{r.iters.max()}, "
      f"kernel {r.kernel_ms:.1f} ms -> {len(full) / (r.kernel_ms * 1e-3):.0f} NLPs/s (wall {wall:.2f} s with host transfers), t_f {r.tf.min() * 470:.1f} .. {r.tf.max() * 470:.1f} s", flush=True)
tot = 0.0; ok = 0
for s in range(8):
    rs = A.solve_batch(full[s * 32768:(s + 1) * 32768], 200, want_traj=False)
    tot += rs.kernel_ms; ok += int((rs.status == 0).sum())
    assert np.array_equal(rs.tf, r.tf[s * 32768:(s + 1) * 32768]) and np.array_equal(rs.iters, r.iters[s * 32768:(s + 1) * 32768])
print(f"eight shards of 32768 one after another: {ok} converged, {tot:.1f} ms of kernels -> {len(full) / (tot * 1e-3):.0f} NLPs/s; results identical to the single batch")
