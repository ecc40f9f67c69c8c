"""The l1 move penalty inside the persistent kernel (p_solve<.,0,1>): Newton step against the C restatement and the dense-block
path; solves against tests/golden/dcost_fixtures.json, the C restatement (iteration counts) and the dense-block path; timing of
the config-3 sweep with the reference's DCOST."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle

c_oracle.build()
S = A.sweep_isp_drymass(2, 2); S[:, 15] = 1e-5
for scheme in (0, 1):
    nt = 40; K = nt - 1
    blobs = []
    for b, row in enumerate(S):
        rng = np.random.default_rng(300 + b)
        r = c_oracle.solve_batch(row[None], nt, 3 + b % 4, 1e-9, want_blob=True, coarse_nodes=-1, scheme=scheme)
        blob = r["blob"][0].copy()
        blob[8 * K:15 * K] += 0.05 * rng.standard_normal(7 * K)
        blob[15 * K:21 * K] *= rng.uniform(0.7, 1.3, 6 * K)
        blobs.append(blob)
    blobs = np.stack(blobs, axis=1)
    mu = np.array([0.1, 0.02, 1e-3, 0.05]); dw = np.array([0.0, 1e-4, 1e-2, 1.0])
    for path in ("persist", "dense"):
        step, inertia = A.kkt_step(S, blobs, mu, dw, nt, path=path, scheme=scheme, move_penalty=True)
        for b in range(4):
            rc, ref = c_oracle.newton_step(S[b], nt, np.ascontiguousarray(blobs[:, b]), mu[b], dw[b], scheme=scheme, move_penalty=True)
            err = np.abs(step[:, b] - ref).max() / max(1.0, np.abs(ref).max())
            print(f"step scheme {scheme} {path} problem {b}: inertia {inertia[b]} oracle {rc} rel err {err:.2e}")

fx = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "dcost_fixtures.json")))
for c in fx["cases"]:
    if c["scheme"] == 2:
        continue
    P = A.AscentParams(**c["params"])
    for path in ("auto", "dense"):
        r = A.solve_batch(P, c["nt"], tol=1e-9, scheme=c["scheme"], max_iter=500, move_penalty=True, path=path)
        ref = c["on"]
        u = r.traj[8, 1:, 0]
        tv = np.abs(np.diff(np.concatenate([[0.0], u]))).sum()
        o = c_oracle.solve_batch(P.as_row()[None], c["nt"], 500, 1e-9, scheme=c["scheme"], move_penalty=True)
        print(f"nt {c['nt']} scheme {c['scheme']} dcost {c['dcost']} {path}: status {r.status[0]} iters {r.iters[0]} (oracle {o['iters'][0]}) "
              f"tf-fixture {r.tf[0] - ref['tf']:.2e} tf-oracle {r.tf[0] - o['tf'][0]:.2e} tv {tv:.5f}/{ref['total_variation']:.5f} du {np.abs(u - np.array(ref['u'])).max():.1e}")
print("default path 4096 mp:", A.default_path(4096, 200, move_penalty=True), " 1:", A.default_path(1, 200, move_penalty=True))

Sw = A.sweep_isp_drymass(64, 64); Sw[:, 15] = 1e-5
for mp in (False, True):
    for rep in range(3):
        t = time.time()
        r = A.solve_batch(Sw, 200, tol=1e-9, want_traj=False, move_penalty=mp)
        wall = time.time() - t
    print(f"4096 NLPs move_penalty={mp}: kernel {r.kernel_ms:.2f} ms wall {wall*1e3:.1f} ms converged {(r.status == 0).sum()} iters {r.iters.min()}-{r.iters.max()} mean {r.iters.mean():.1f}")
    if mp:
        o = c_oracle.solve_batch(Sw[::257], 200, 300, 1e-9, move_penalty=True)
        print("   vs oracle on 16: tf diff", np.abs(r.tf[::257] - o["tf"]).max(), "iters", r.iters[::257], o["iters"])
for B in (1, 16, 256, 1024, 16384):
    Sx = np.tile(Sw, (max(1, B // 4096), 1))[:B]
    for rep in range(2):
        r = A.solve_batch(Sx, 200, tol=1e-9, want_traj=False, move_penalty=True)
    print(f"batch {B} with the move penalty: kernel {r.kernel_ms:.2f} ms, converged {(r.status == 0).sum()}")
