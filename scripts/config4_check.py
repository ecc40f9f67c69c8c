"""Robustness over BASELINE.json config 4's parameter box: shard 0 (32768 NLPs) + a spread sample of the 262144."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
full = A.sweep_config4()
print("config 4 grid", full.shape)
for name, S in (("shard 0", full[:32768]), ("every 8th", np.ascontiguousarray(full[::8]))):
    t = time.time(); r = A.solve_batch(S, 200, want_traj=False); el = time.time() - t
    print(f"{name}: {len(S)} NLPs, status counts {np.bincount(r.status, minlength=4)}, iters {r.iters.min()}/{r.iters.mean():.1f}/{r.iters.max()}, "
          f"kernel {r.kernel_ms:.1f} ms -> {len(S)/(r.kernel_ms*1e-3):.0f} NLP/s, tf range {r.tf.min()*470:.1f}..{r.tf.max()*470:.1f} s")
    bad = np.nonzero(r.status != 0)[0]
    f = A.PARAM_FIELDS
    for b in bad[:10]:
        print("   failed:", b, "status", r.status[b], "iters", r.iters[b], {k: S[b, f.index(k)] for k in ("mdot", "M0", "r_apo", "ang_acc_max")})
