"""Developer tool: PROF_N (default 3) default-dispatch solves of one workload and nothing else on the GPU -- the command rocprofv3
runs for profiles/traffic.json (scripts/regen_traffic.sh).  PROF_WORKLOAD:
    config3        BASELINE configs[2]: 4096-NLP Isp x dry-mass sweep, N=200, backward Euler, WITH the reference's DCOST (the bench headline)
    config3_nodcost   the same without the move penalty (rounds 1 / 2's headline)
    config4        first 32768-NLP shard of BASELINE configs[3]'s box (no move penalty, no trajectories)
    hs4096         the config-3 sweep with Hermite-Simpson (scheme 2: the persistent kernel h_solve)
    config5        256 NLPs of the sweep at N=2000, Hermite-Simpson, terminal 1 (BASELINE configs[4] as a batch)
    config5_free   the same batch with terminal 2 (burnout anywhere on the ellipse)
    config5_one    the nominal problem at N=2000, Hermite-Simpson, terminal 1 (one NLP: dense blocks + PCR)
PROF_BATCH overrides the batch size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A

W = os.environ.get("PROF_WORKLOAD", "config3")
kw = dict(tol=1e-9)
nt = 200
if W in ("config3", "config3_nodcost"):
    B = int(os.environ.get("PROF_BATCH", "4096"))
    S = A.sweep_isp_drymass()[:: max(1, 4096 // B)][:B]
    kw["move_penalty"] = W == "config3"
elif W == "config4":
    B = int(os.environ.get("PROF_BATCH", "32768"))
    S = np.ascontiguousarray(A.sweep_config4()[:B]); kw["want_traj"] = False
elif W == "hs4096":
    B = int(os.environ.get("PROF_BATCH", "4096"))
    S = A.sweep_isp_drymass()[:: max(1, 4096 // B)][:B]; kw.update(scheme=2, max_iter=500)
elif W == "config5":
    B = int(os.environ.get("PROF_BATCH", "256"))
    S = A.sweep_isp_drymass(16, 16)[:B]; nt = 2000; kw.update(scheme=2, terminal="ellipse", max_iter=500, want_traj=False)
elif W == "config5_free":
    B = int(os.environ.get("PROF_BATCH", "256"))
    S = A.sweep_isp_drymass(16, 16)[:B]; nt = 2000; kw.update(scheme=2, terminal="ellipse_free", max_iter=500, want_traj=False)
elif W == "config5_one":
    S = A.AscentParams().as_row()[None]; B = 1; nt = 2000; kw.update(scheme=2, terminal="ellipse", max_iter=500)
else:
    raise SystemExit(f"unknown PROF_WORKLOAD {W}")
S = S.copy(); S[:, 15] = 1e-5
for _ in range(int(os.environ.get("PROF_N", "3"))):
    r = A.solve_batch(S, nt, **kw)
print("workload", W, "batch", B, "nt", nt, "path", A.default_path(B, nt, scheme=kw.get("scheme", 0), move_penalty=kw.get("move_penalty", False), terminal=kw.get("terminal", 0)),
      "iters", r.iters.mean(), "converged", (r.status == 0).sum(), "kernel ms", A.last_kernel_ms())
