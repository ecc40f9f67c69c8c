"""Developer check: ascent_opts.move_penalty = 1 (the reference's MV DCOST as an l1 term) on the GPU against tests/golden/dcost_fixtures.json."""
import json, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
fx = json.load(open("tests/golden/dcost_fixtures.json"))
for c in fx["cases"]:
    P = A.AscentParams(**c["params"])
    for mp in (False, True):
        r = A.solve_batch(P, c["nt"], tol=1e-9, scheme=c["scheme"], max_iter=500, move_penalty=mp)
        u = r.traj[8, 1:, 0]
        tv = np.abs(np.diff(np.concatenate([[0.0], u]))).sum()
        ref = c["on" if mp else "off"]
        print(f"nt {c['nt']} scheme {c['scheme']} dcost {c['dcost']:g} penalty {int(mp)}: status {r.status[0]} iters {r.iters[0]} tf {r.tf[0]:.12f} (ref {ref['tf']:.12f}, diff {r.tf[0] - ref['tf']:.1e}) "
              f"TV {tv:.4f} (ref {ref['total_variation']:.4f}) max|u - ref| {np.abs(u - np.array(ref['u'])).max():.1e}", flush=True)

S = A.sweep_isp_drymass()
S[:, 15] = 1e-5
for B in (1, 64, 1024, 4096):
    Sb = S[:: max(1, 4096 // B)][:B]
    off = A.solve_batch(Sb, 200, tol=1e-9, want_traj=False)
    ms = [A.solve_batch(Sb, 200, tol=1e-9, want_traj=False, move_penalty=True, max_iter=500).kernel_ms for _ in range(3)]
    on = A.solve_batch(Sb, 200, tol=1e-9, want_traj=False, move_penalty=True, max_iter=500)
    print(f"batch {B}: move penalty on: {min(ms):.1f} ms, converged {(on.status == 0).sum()}/{B}, iterations {on.iters.min()}-{on.iters.max()}, t_f shift mean {(on.tf - off.tf).mean() * 470:.2e} s (max {(on.tf - off.tf).max() * 470:.2e}); off: {off.kernel_ms:.1f} ms", flush=True)
