"""Stability soak: many solves of mixed sizes / paths / schemes; checks determinism and device-memory drift."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lunar_module_ascent_trajectory_optimiser_amd as A
ref = {}
free0 = None
t0 = time.time()
for rep in range(12):
    for (B, nt, kw) in ((1, 200, {}), (100, 60, {}), (4096, 200, {}), (20000, 200, {}), (300, 200, dict(scheme=1)),
                        (64, 200, dict(formulation=1, params=A.AscentParams(r_peri=53108.4, r_apo=53108.4, mass_scalar=2576.0))),
                        (16, 200, dict(scheme=2)), (2000, 150, dict(scheme=2, terminal="ellipse")), (30, 500, dict(scheme=2, terminal="ellipse_free")), (100, 120, dict(move_penalty=True, params=A.AscentParams(dcost=1e-5)))):
        base = kw.pop("params", None) if "params" in kw else None
        kw2 = {k: v for k, v in kw.items()}
        n = int(round(B ** 0.5))
        while B % n: n -= 1
        S = A.sweep_isp_drymass(n, B // n, base=base)
        r = A.solve_batch(S, nt, want_traj=False, max_iter=500, **kw2)
        key = (B, nt, tuple(sorted(kw2.items())))
        sig = (r.tf.sum(), int(r.iters.sum()), int((r.status == 0).sum()))
        if key in ref:
            assert ref[key] == sig, (key, ref[key], sig)
        ref[key] = sig
        assert sig[2] == B, (key, sig)
        if base is not None: kw["params"] = base
    free, total = torch.cuda.mem_get_info(0)
    if free0 is None: free0 = free
    print(f"rep {rep}: free device memory {free/2**30:.2f} GiB (start {free0/2**30:.2f}), elapsed {time.time()-t0:.1f}s", flush=True)
assert free0 - free < 64 * 2**20, "device memory drift"
print("soak ok: bit-identical results across repetitions, no device-memory drift")
