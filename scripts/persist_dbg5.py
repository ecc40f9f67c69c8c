import os, sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()[::64][:16]
P = S[11:12]; K = 199
for mi in (26, 27, 28):
    o = {}
    for mode in ("split", "persist"):
        os.environ["ASCENT_PIPELINE"] = mode
        o[mode] = A.solve_batch(P, 200, tol=1e-12, max_iter=mi, coarse_nodes=-1, want_blob=True)
    a, b = o["split"].blob[:, 0], o["persist"].blob[:, 0]
    d = np.abs(a - b)
    Z = d[:7*K].reshape(K, 7); Lm = d[8*K:15*K].reshape(K, 7); ZB = d[15*K:21*K].reshape(K, 6)
    print(mi, "z max per field", Z.max(0), "at node", Z.argmax(0))
    print("   u", d[7*K:8*K].max(), int(d[7*K:8*K].argmax()), "lam per field", Lm.max(0), "node", Lm.argmax(0))
    print("   zb per bound", ZB.max(0), "node", ZB.argmax(0), "scal", d[21*K:])
