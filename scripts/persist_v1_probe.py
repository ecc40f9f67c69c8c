"""Developer check: one round of the persistent kernel, formulation 1 (the v1 script: the angle is the MV), against the C restatement."""
import sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle as O
nt = 60; K = nt - 1
base = A.AscentParams(r_peri=53108.4, r_apo=53108.4, mass_scalar=2576.0)
S = A.sweep_isp_drymass(2, 3, base=base)
blobs = []
for b, row in enumerate(S):
    rng = np.random.default_rng(100 + b)
    r = O.solve_batch(row[None], nt, 3 + b % 4, 1e-9, want_blob=True, coarse_nodes=-1, formulation=1)
    blob = r["blob"][0].copy()
    blob[8 * K:15 * K] += 0.05 * rng.standard_normal(7 * K)
    blob[15 * K:21 * K] *= rng.uniform(0.7, 1.3, 6 * K)
    blobs.append(blob)
O.set_formulation(0)
blobs = np.stack(blobs, axis=1)
mu = np.array([0.1, 0.02, 1e-3, 0.05, 0.01, 0.2]); dw = np.array([0.0, 0.0, 1e-2, 1.0, 0.0, 1e-4])
for path in ("split_wide", "persist"):
    step, inertia = A.kkt_step(S, blobs, mu, dw, nt, path=path, formulation=1)
    for b in range(len(S)):
        rc, ref = O.newton_step(S[b], nt, np.ascontiguousarray(blobs[:, b]), mu[b], dw[b], formulation=1)
        O.set_formulation(0)
        err = [np.abs(step[lo:hi, b] - ref[lo:hi]).max() / max(1.0, np.abs(ref[lo:hi]).max()) for lo, hi in ((0, 7 * K), (7 * K, 8 * K), (8 * K, 15 * K), (15 * K, 21 * K), (21 * K, 21 * K + 10))]
        print(f"{path:10s} NLP {b}: inertia gpu {inertia[b]} oracle {rc}; rel err dz {err[0]:.1e} du {err[1]:.1e} dl {err[2]:.1e} dzb {err[3]:.1e} scal {err[4]:.1e}")
