"""Developer tool: nested iteration probe -- solve on a coarse grid, prolong the primal-dual solution to N = 200 on
the host (numpy), warm-start the fine solve; iterations and device time of both legs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lunar_module_ascent_trajectory_optimiser_amd as A

def prolong(blob, Kc, Kf):
    B = blob.shape[1]
    tc = np.arange(0, Kc + 1) / Kc            # includes the fixed node 0
    tf_ = np.arange(1, Kf + 1) / Kf
    out = np.empty((21 * Kf + 10, B))
    def interp(rows_c, n, zero0=True, first=None):
        # rows_c: (Kc, n, B) values at nodes 1..Kc -> (Kf, n, B)
        v0 = np.zeros((1, n, B)) if zero0 else rows_c[:1]
        full = np.concatenate([v0, rows_c], 0)                    # nodes 0..Kc
        idx = np.minimum((tf_ * Kc).astype(int), Kc - 1)
        w = (tf_ * Kc - idx)[:, None, None]
        return (1 - w) * full[idx] + w * full[idx + 1]
    z = blob[:7 * Kc].reshape(Kc, 7, B)
    u = blob[7 * Kc:8 * Kc].reshape(Kc, 1, B)
    lam = blob[8 * Kc:15 * Kc].reshape(Kc, 7, B)
    zb = blob[15 * Kc:21 * Kc].reshape(Kc, 6, B)
    out[:7 * Kf] = interp(z, 7).reshape(7 * Kf, B)
    out[7 * Kf:8 * Kf] = interp(u, 1, zero0=False).reshape(Kf, B)
    out[8 * Kf:15 * Kf] = interp(lam, 7, zero0=False).reshape(7 * Kf, B)
    out[15 * Kf:21 * Kf] = (interp(zb, 6, zero0=False) * (Kc / Kf)).reshape(6 * Kf, B)
    out[21 * Kf:] = blob[21 * Kc:]
    return out

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
if which == "c3":
    S = A.sweep_isp_drymass()
else:
    P4 = A.sweep_config4(); S = P4[np.random.default_rng(1).choice(len(P4), 4096, replace=False)]
NT = 200
ref = A.solve_batch(S, NT, want_traj=False)
print(which, "cold:", ref.kernel_ms, "ms iters", ref.iters.mean(), ref.iters.max(), "conv", int(ref.converged.sum()))
# Richardson in h: two coarse grids (the second warm-started from the first), both prolonged to the fine grid,
# extrapolated linearly in h to the fine step
for (n1, n2) in ((10, 19), (14, 27), (18, 35)):
    c1 = A.solve_batch(S, n1, want_traj=False, want_blob=True)
    c2 = A.solve_batch(S, n2, guess=prolong(c1.blob, n1 - 1, n2 - 1), warm_start=2, mu_init=1e-5, want_traj=False, want_blob=True)
    K1, K2, Kf = n1 - 1, n2 - 1, NT - 1
    w = (1 / K2 - 1 / Kf) / (1 / K1 - 1 / K2)
    p1, p2 = prolong(c1.blob, K1, Kf), prolong(c2.blob, K2, Kf)
    g = p2 + w * (p2 - p1)
    # keep multipliers / slacks positive (rows 15K..21K are bound multipliers; scalars zlt zut s1 s2 zs1 zs2)
    g[15 * Kf:21 * Kf] = np.maximum(g[15 * Kf:21 * Kf], 0.1 * p2[15 * Kf:21 * Kf])
    g[21 * Kf + 1:21 * Kf + 7] = np.maximum(g[21 * Kf + 1:21 * Kf + 7], 0.1 * p2[21 * Kf + 1:21 * Kf + 7])
    for mu0 in (1e-5, 1e-6, 1e-7, 1e-8):
        f = A.solve_batch(S, NT, guess=g, warm_start=2, mu_init=mu0, want_traj=False)
        ok = f.converged & ref.converged
        print(f"richardson {n1},{n2}: {c1.kernel_ms:.2f}+{c2.kernel_ms:.2f} ms iters {c1.iters.mean():.1f} {c2.iters.mean():.1f} | fine mu0={mu0:g}: {f.kernel_ms:6.2f} ms iters {f.iters.mean():.1f}/{f.iters.max()} "
              f"conv {int(f.converged.sum())} | total {c1.kernel_ms + c2.kernel_ms + f.kernel_ms:6.2f} ms | max |tf - cold| {np.abs(f.tf - ref.tf)[ok].max():.1e}", flush=True)
sys.exit(0)
for ntc in (14, 18, 22, 26, 34):
    c = A.solve_batch(S, ntc, want_traj=False, want_blob=True)
    g = prolong(c.blob, ntc - 1, NT - 1)
    for mu0 in (1e-4, 3e-5, 1e-5, 3e-6):
        f = A.solve_batch(S, NT, guess=g, warm_start=2, mu_init=mu0, want_traj=False)
        ok = f.converged & ref.converged
        print(f"coarse N={ntc}: {c.kernel_ms:6.2f} ms iters {c.iters.mean():.1f}/{c.iters.max()} conv {int(c.converged.sum())} | fine mu0={mu0:g}: {f.kernel_ms:6.2f} ms iters {f.iters.mean():.1f}/{f.iters.max()} "
              f"conv {int(f.converged.sum())} | total {c.kernel_ms + f.kernel_ms:6.2f} ms | max |tf - cold| {np.abs(f.tf - ref.tf)[ok].max():.1e}", flush=True)
# three levels
for chain in ((14, 51), (10, 42), (8, 30)):
    c = A.solve_batch(S, chain[0], want_traj=False, want_blob=True)
    m = A.solve_batch(S, chain[1], guess=prolong(c.blob, chain[0] - 1, chain[1] - 1), warm_start=2, mu_init=1e-5, want_traj=False, want_blob=True)
    f = A.solve_batch(S, NT, guess=prolong(m.blob, chain[1] - 1, NT - 1), warm_start=2, mu_init=1e-5, want_traj=False)
    print(f"chain {chain} -> 200: {c.kernel_ms:.2f} + {m.kernel_ms:.2f} + {f.kernel_ms:.2f} = {c.kernel_ms + m.kernel_ms + f.kernel_ms:.2f} ms; iters {c.iters.mean():.1f} {m.iters.mean():.1f} {f.iters.mean():.1f}/{f.iters.max()} conv {int(f.converged.sum())}")
