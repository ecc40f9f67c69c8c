"""PCR variant of the dense path's Newton solve: step parity with the Riccati variant, solves, single-NLP latency."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle

nt = 40; K = nt - 1
S = A.sweep_isp_drymass(2, 2)
blobs = np.stack([c_oracle.solve_batch(row[None], nt, 3 + b, 1e-9, want_blob=True, coarse_nodes=-1)["blob"][0] for b, row in enumerate(S)], 1)
mu = np.array([0.1, 0.02, 1e-3, 0.05]); dw = np.array([0.0, 0.0, 1e-2, 1.0])
out = {}
for mode in ("riccati", "pcr"):
    os.environ["ASCENT_DENSE_NEWTON"] = mode
    for scheme in (0, 2):
        out[mode, scheme] = A.kkt_step(S, blobs, mu, dw, nt, path="dense", scheme=scheme)
for scheme in (0, 2):
    a, b = out["riccati", scheme], out["pcr", scheme]
    print("scheme", scheme, "inertia", a[1], b[1], "max step diff", np.abs(a[0] - b[0]).max(axis=0), "scale", np.abs(a[0]).max(axis=0))
for nt_, scheme, term in ((200, 0, 0), (200, 2, 0), (2000, 2, 1)):
    for B in (1, 8):
        P = np.vstack([A.AscentParams().as_row()[None], A.sweep_isp_drymass(3, 3)])[:B]
        res = {}
        for mode in ("riccati", "pcr"):
            os.environ["ASCENT_DENSE_NEWTON"] = mode
            A.solve_batch(P, nt_, tol=1e-9, scheme=scheme, terminal=term, path="dense", max_iter=500)
            t = time.time()
            r = A.solve_batch(P, nt_, tol=1e-9, scheme=scheme, terminal=term, path="dense", max_iter=500)
            res[mode] = (r, time.time() - t)
        r0, r1 = res["riccati"][0], res["pcr"][0]
        print(f"nt {nt_} scheme {scheme} B {B}: riccati {res['riccati'][1]*1e3:.1f} ms (kernel {r0.kernel_ms:.1f}) iters {r0.iters}  |  pcr {res['pcr'][1]*1e3:.1f} ms (kernel {r1.kernel_ms:.1f}) "
              f"iters {r1.iters} status {r1.status}  tf diff {np.abs(r0.tf - r1.tf).max():.1e}")
os.environ.pop("ASCENT_DENSE_NEWTON")
t = time.time(); h = A.solve_batch(A.AscentParams(), 200, tol=1e-9); print("hand-tuned single NLP:", (time.time() - t) * 1e3, "ms kernel", h.kernel_ms)
t = time.time(); o = c_oracle.solve_batch(A.AscentParams().as_row()[None], 200, 300, 1e-9); print("C oracle single NLP, one core:", (time.time() - t) * 1e3, "ms")
