"""Developer study: the residual K step - rhs of one Newton step of the persistent kernel and the split pipeline in the full sparse KKT
system (assembled by the numpy oracle), by row type, and the step error against a generic sparse LU by node and component.
python scripts/kkt_residual.py <index in the 64-NLP config-4 sample> <tolerance of the solve whose end point is the iterate>
<factor on the terminal slack multipliers: != 1 leaves nu + z != 0, as a truncated dual step does> <scheme>   (LIB= variant build).
This is how the contraction problem of DESIGN.md section 4a was found: kkt_residual.py 42 1e-8 0.9 1."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import scipy.sparse as sp
import os
from lunar_module_ascent_trajectory_optimiser_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['LIB'])
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle as O
from conftest import generic_lu_newton_step, params_of_row
np.set_printoptions(linewidth=220, precision=2)
S = A.sweep_config4()[::4099][:64]
i = int(sys.argv[1]) if len(sys.argv) > 1 else 0; nt = 200; K = nt - 1
tolb = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
scheme = int(sys.argv[4]) if len(sys.argv) > 4 else 1
sol = O.solve_batch(S[i:i+1], nt, 500, tolb, want_blob=True, scheme=scheme); O.set_scheme(0)
blob = sol["blob"][0].copy(); mu = 1e-9; dw = 0.0
f = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
blob[21 * K + 5] *= f; blob[21 * K + 6] *= f
print('scalars', blob[21 * K:])
lu, nlp, v, lam = generic_lu_newton_step(params_of_row(S[i]), nt, blob, mu, dw, scheme)
hasL, hasU = np.isfinite(nlp.lb), np.isfinite(nlp.ub)
zL = np.zeros(nlp.n); zU = np.zeros(nlp.n)
zb = blob[15 * K:21 * K].reshape(K, 6); sc = blob[21 * K:]; base = np.arange(K) * 8
zL[base + 4], zU[base + 4], zL[base + 6], zU[base + 6], zL[base + 7], zU[base + 7] = zb.T
zL[nlp.itf], zU[nlp.itf] = sc[1], sc[2]; zL[nlp.is1], zL[nlp.is2] = sc[5], sc[6]
dL = np.where(hasL, v - nlp.lb, 1.0); dU = np.where(hasU, nlp.ub - v, 1.0)
c = nlp.constraints(v); J = nlp.jacobian(v); W = nlp.hessian(v, lam)
Sig = np.where(hasL, zL / dL, 0) + np.where(hasU, zU / dU, 0)
gphi = nlp.grad_objective(v) - np.where(hasL, mu / dL, 0) + np.where(hasU, mu / dU, 0)
Kmat = sp.bmat([[W + sp.diags(Sig + dw), J.T], [J, None]], format="csr")
rhs = -np.concatenate([gphi + J.T @ lam, c])
def unpack(step):
    dx = np.zeros(nlp.n); dlam = np.zeros(nlp.m)
    dW = dx[:8 * K].reshape(K, 8)
    dW[:, :7] = step[:7 * K].reshape(K, 7); dW[:, 7] = step[7 * K:8 * K]
    dlam[:7 * K] = step[8 * K:15 * K]
    s = step[21 * K:]
    dx[nlp.itf], dx[nlp.is1], dx[nlp.is2] = s[0], s[3], s[4]
    dlam[-3], dlam[-2], dlam[-1] = s[7], s[8], s[9]
    return np.concatenate([dx, dlam])
names = "x y vx vy a w m u".split()
for tag, st in (("lu", lu),) + tuple((p, A.kkt_step(S[i:i+1], blob[:, None], mu, dw, nt, path=p, scheme=scheme)[0][:, 0]) for p in ("persist", "split_wide")):
    r = Kmat @ unpack(st) - rhs
    rd = r[:8 * K].reshape(K, 8); rp = r[nlp.n:nlp.n + 7 * K].reshape(K, 7)
    print(tag, "dual rows max by var", dict(zip(names, np.abs(rd).max(0))))
    print(tag, "   argmax node", np.abs(rd).argmax(0), " tf s1 s2 rows", r[nlp.itf], r[nlp.is1], r[nlp.is2])
    print(tag, "   primal rows max by comp", np.abs(rp).max(0), "terminal", r[-3:])
    print(tag, "   dual u-row residual at nodes 0,1,2,100,K-3..K-1", rd[[0, 1, 2, 100, K - 3, K - 2, K - 1], 7])
    print(tag, "   dual w-row residual ", rd[[0, 1, 2, 100, K - 3, K - 2, K - 1], 5])
for p in ("persist", "split_wide"):
    v = A.kkt_step(S[i:i+1], blob[:, None], mu, dw, nt, path=p, scheme=scheme)[0][:, 0]
    print(f"{p:10s} scal err {v[21*K:] - lu[21*K:]}")
    print(f"{p:10s} scal     {lu[21*K:]}")
    dz = (v[:7*K]-lu[:7*K]).reshape(K,7); du = v[7*K:8*K]-lu[7*K:8*K]; dl = (v[8*K:15*K]-lu[8*K:15*K]).reshape(K,7)
    for k in (0, 1, 100, K-3, K-2, K-1):
        print(f"{p:10s} node {k:3d} dz err {dz[k]} du err {du[k]:.2e}  | dz {lu[:7*K].reshape(K,7)[k]} du {lu[7*K+k]:.2e}")
    for k in (0, 1, 100, K-3, K-2, K-1):
        print(f"{p:10s} node {k:3d} dl err {dl[k]} | dl {lu[8*K:15*K].reshape(K,7)[k]}")
