"""Measured: parallel cyclic reduction over the collocation nodes vs block elimination serial in the node index, on KKT-shaped
systems (15x15 blocks, two border columns), device time of the solve proper (HIP events)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A

rng = np.random.default_rng(0)
print(f"{'n':>5s} {'batch':>6s} {'thomas ms':>10s} {'pcr ms':>8s} {'ratio':>6s} {'max err':>9s}")
for n in (200, 2000):
    for B in (1, 16, 64, 256):
        bs, nb = 15, 2
        D = rng.standard_normal((B, n, bs, bs)) + 16.0 * np.eye(bs); D = 0.5 * (D + np.swapaxes(D, 2, 3))
        L = rng.standard_normal((B, n, bs, bs)); U = np.zeros_like(L); U[:, :-1] = np.swapaxes(L[:, 1:], 2, 3)
        bor = rng.standard_normal((B, n, bs, nb)); bd = rng.standard_normal((B, nb, nb)) + 3 * np.eye(nb)
        rhs = rng.standard_normal((B, n * bs + nb))
        out = {}
        for algo in ("thomas", "pcr"):
            A.kkt_solve(D, L, U, rhs, bor, bd, algo=algo)
            ts = []
            for _ in range(3):
                sol, ms = A.kkt_solve(D, L, U, rhs, bor, bd, algo=algo)
                ts.append(ms)
            out[algo] = (min(ts), sol)
        err = np.abs(out["thomas"][1] - out["pcr"][1]).max()
        print(f"{n:5d} {B:6d} {out['thomas'][0]:10.3f} {out['pcr'][0]:8.3f} {out['thomas'][0]/out['pcr'][0]:6.2f} {err:9.1e}", flush=True)
