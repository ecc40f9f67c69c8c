"""Developer probe: HIP path vs the oracle on a GPU box, with timings. Not part of the test suite."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle as co

nt = 200; K = nt - 1
P0 = A.AscentParams()
p16 = P0.as_row()
# 1. single nominal
t = time.time(); r = A.solve_batch(P0, nt, want_blob=True); el = time.time() - t
print("nominal: status", r.status, "iters", r.iters, "tf*T", r.final_time(), "wall", el, "kernel_ms", r.kernel_ms)
o = co.solve_batch(p16[None], nt, 300, 1e-9, want_blob=True)
print("oracle : status", o["status"], "iters", o["iters"], "tf*T", o["tf"] * 470)
print("tf rel diff", abs(r.tf[0] - o["tf"][0]) / o["tf"][0], "traj maxdiff", np.abs(r.traj[:, :, 0] - o["traj"][0]).max())
print("oracle kkt error at GPU solution", co.kkt_error(p16, nt, np.ascontiguousarray(r.blob[:, 0])))
# 2. newton step parity at the oracle's mid-solve iterate (use cold-start point from oracle after few its)
o3 = co.solve_batch(p16[None], nt, 5, 1e-9, want_blob=True)
blob = o3["blob"][0]
rc, st = co.newton_step(p16, nt, blob, 0.02, 0.0)
stg, inert = A.kkt_step(P0, blob[:, None], 0.02, 0.0, nt)
print("kkt_step inertia", rc, inert, "max abs diff", np.abs(stg[:, 0] - st).max(), "scale", np.abs(st).max())
# 3. batches
for (ni, nd) in ((8, 8), (64, 64)):
    S = A.sweep_isp_drymass(ni, nd)
    t = time.time(); r = A.solve_batch(S, nt, want_traj=False); el = time.time() - t
    print(f"batch {ni*nd}: converged {r.converged.sum()}/{len(r.tf)} iters min/mean/max {r.iters.min()}/{r.iters.mean():.1f}/{r.iters.max()} "
          f"wall {el:.3f}s kernel {r.kernel_ms:.1f} ms -> {len(r.tf)/ (r.kernel_ms*1e-3):.0f} NLP/s; status counts {np.bincount(r.status, minlength=4)}")
    idx = np.linspace(0, len(S) - 1, 6).astype(int)
    oo = co.solve_batch(S[idx], nt, 300, 1e-9)
    print("   sample tf rel diffs", np.abs(r.tf[idx] - oo["tf"]) / oo["tf"], "oracle status", oo["status"], "iters", oo["iters"], "gpu iters", r.iters[idx])
if len(sys.argv) > 1:
    n = int(sys.argv[1])
    S = np.tile(A.sweep_isp_drymass(64, 64), (n // 4096, 1))
    t = time.time(); r = A.solve_batch(S, nt, want_traj=False); el = time.time() - t
    print(f"batch {len(S)}: converged {r.converged.sum()} wall {el:.3f}s kernel {r.kernel_ms:.1f} ms -> {len(S)/(r.kernel_ms*1e-3):.0f} NLP/s")
