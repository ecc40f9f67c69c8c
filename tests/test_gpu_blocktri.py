"""GPU tests of the generic bordered block-tridiagonal solver (ascent_kkt_solve; csrc/ascent_blocktri.hip: 16x16 blocks on
v_mfma_f64_16x16x4_f64, block elimination serial in the node index vs parallel cyclic reduction over the nodes) against
dense LAPACK solves -- on random systems and on the real Newton system of the ascent NLP reordered by
collocation node."""
import numpy as np
import pytest

import lunar_module_ascent_trajectory_optimiser_amd as A

pytestmark = pytest.mark.gpu


def _random_system(rng, B, n, bs, nb, symmetric):
    D = rng.standard_normal((B, n, bs, bs)) + 4.0 * bs ** 0.5 * np.eye(bs)
    L = rng.standard_normal((B, n, bs, bs)); U = rng.standard_normal((B, n, bs, bs))
    if symmetric:
        D = 0.5 * (D + np.swapaxes(D, 2, 3))
        U[:, :-1] = np.swapaxes(L[:, 1:], 2, 3)
    bor = rng.standard_normal((B, n, bs, nb)); bd = rng.standard_normal((B, nb, nb)) + 3.0 * np.eye(nb)
    rhs = rng.standard_normal((B, n * bs + nb))
    return D, L, U, bor, bd, rhs


def _dense(D, L, U, bor, bd, b):
    n, bs = D.shape[1], D.shape[2]
    nb = bor.shape[-1]
    N = n * bs + nb
    M = np.zeros((N, N))
    for i in range(n):
        M[i * bs:(i + 1) * bs, i * bs:(i + 1) * bs] = D[b, i]
        if i:
            M[i * bs:(i + 1) * bs, (i - 1) * bs:i * bs] = L[b, i]
        if i < n - 1:
            M[i * bs:(i + 1) * bs, (i + 1) * bs:(i + 2) * bs] = U[b, i]
        M[i * bs:(i + 1) * bs, n * bs:] = bor[b, i]
        M[n * bs:, i * bs:(i + 1) * bs] = bor[b, i].T
    M[n * bs:, n * bs:] = bd[b]
    return M


@pytest.mark.parametrize("algo", ["thomas", "pcr"])
@pytest.mark.parametrize("n,bs,nb,B,sym", [(1, 5, 0, 2, False), (2, 16, 3, 2, False), (5, 7, 2, 3, True), (200, 15, 2, 3, True),
                                           (333, 8, 1, 2, False), (64, 3, 15, 2, False)])
def test_random_bordered_block_tridiagonal_systems(algo, n, bs, nb, B, sym):
    rng = np.random.default_rng(n * 100 + bs)
    D, L, U, bor, bd, rhs = _random_system(rng, B, n, bs, max(nb, 1), sym)
    if nb == 0:
        sol, _ = A.kkt_solve(D, L, U, rhs[:, :n * bs], algo=algo)
    else:
        sol, _ = A.kkt_solve(D, L, U, rhs, bor, bd, algo=algo)
    for b in range(B):
        if nb == 0:
            M = _dense(D, L, U, bor[..., :0], bd[:, :0, :0], b)
            ref = np.linalg.solve(M, rhs[b, :n * bs])
        else:
            M = _dense(D, L, U, bor, bd, b)
            ref = np.linalg.solve(M, rhs[b])
        assert np.abs(sol[b] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


def test_singular_block_is_reported():
    D = np.zeros((1, 3, 4, 4)); L = np.zeros_like(D); U = np.zeros_like(D)
    from lunar_module_ascent_trajectory_optimiser_amd._lib import AscentLibraryError
    with pytest.raises(AscentLibraryError, match="singular"):
        A.kkt_solve(D, L, U, np.ones((1, 12)), algo="pcr")


@pytest.mark.parametrize("algo", ["thomas", "pcr"])
def test_newton_system_of_the_ascent_nlp_by_node(coracle, algo):
    """The Newton (KKT) system of the ascent NLP at an interior iterate, assembled by the numpy oracle, reordered by
    collocation node: it IS block tridiagonal with 15x15 blocks (7 states, the control, 7 defect multipliers) plus a border
    of six unknowns (tf, the two terminal slacks, the three terminal multipliers) -- asserted here entry by entry -- and
    the generic solver reproduces the sparse-LU Newton step on it, by serial block elimination and by parallel cyclic
    reduction over the 59 nodes (no pivoting inside blocks: the (z,u) part carries the primal regularisation)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from conftest import generic_lu_newton_step, random_interior_blob
    from oracle.ascent_numpy import Params
    P = Params(); nt = 60; K = nt - 1
    p16 = coracle.pack_params(P)
    blob = random_interior_blob(nt, 1, p16, coracle)
    mu, dw = 0.03, 0.2
    step, nlp, v, lam = generic_lu_newton_step(P, nt, blob, mu, dw, 0)
    n, m = nlp.n, nlp.m
    hasL, hasU = np.isfinite(nlp.lb), np.isfinite(nlp.ub)
    zb = blob[15 * K:21 * K].reshape(K, 6); sc = blob[21 * K:]
    zL = np.zeros(n); zU = np.zeros(n); base = np.arange(K) * 8
    zL[base + 4], zU[base + 4], zL[base + 6], zU[base + 6], zL[base + 7], zU[base + 7] = zb.T
    zL[nlp.itf], zU[nlp.itf], zL[nlp.is1], zL[nlp.is2] = sc[1], sc[2], sc[5], sc[6]
    dL = np.where(hasL, v - nlp.lb, 1.0); dU = np.where(hasU, nlp.ub - v, 1.0)
    J = nlp.jacobian(v); W = nlp.hessian(v, lam)
    Sig = np.where(hasL, zL / dL, 0) + np.where(hasU, zU / dU, 0)
    gphi = nlp.grad_objective(v) - np.where(hasL, mu / dL, 0) + np.where(hasU, mu / dU, 0)
    Kmat = sp.bmat([[W + sp.diags(Sig + dw), J.T], [J, None]], format="csr").toarray()
    rhs = -np.concatenate([gphi + J.T @ lam, nlp.constraints(v)])
    ref = spla.splu(sp.csc_matrix(Kmat)).solve(rhs)
    # node order: (w_k, lambda_k) for k = 1..K, then the border (tf, s1, s2, nu3, nu1, nu2)
    perm = np.concatenate([np.concatenate([np.arange(8 * k, 8 * k + 8), n + np.arange(7 * k, 7 * k + 7)]) for k in range(K)]
                          + [np.array([nlp.itf, nlp.is1, nlp.is2, n + 7 * K, n + 7 * K + 1, n + 7 * K + 2])])
    Kp = Kmat[np.ix_(perm, perm)]; rp = rhs[perm]
    bs, nb = 15, 6
    D = np.stack([Kp[i * bs:(i + 1) * bs, i * bs:(i + 1) * bs] for i in range(K)])[None]
    Lo = np.stack([Kp[i * bs:(i + 1) * bs, (i - 1) * bs:i * bs] if i else np.zeros((bs, bs)) for i in range(K)])[None]
    Up = np.stack([Kp[i * bs:(i + 1) * bs, (i + 1) * bs:(i + 2) * bs] if i < K - 1 else np.zeros((bs, bs)) for i in range(K)])[None]
    bor = np.stack([Kp[i * bs:(i + 1) * bs, K * bs:] for i in range(K)])[None]
    bd = Kp[K * bs:, K * bs:][None]
    # the structure claim: nothing outside the block tridiagonal + border
    rest = Kp.copy()
    for i in range(K):
        rest[i * bs:(i + 1) * bs, max(0, i - 1) * bs:min(K, i + 2) * bs] = 0.0
    rest[:, K * bs:] = 0.0; rest[K * bs:, :] = 0.0
    assert np.abs(rest).max() == 0.0
    sol, ms = A.kkt_solve(D, Lo, Up, rp[None], bor, bd, algo=algo)
    x = np.empty_like(ref); x[perm] = sol[0]
    assert np.abs(x - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max())
    # ... which is the Newton step every solver path of this repository computes (here: primal part)
    assert np.abs(x[:8 * K].reshape(K, 8)[:, :7].ravel() - step[:7 * K]).max() <= 1e-7 * max(1.0, np.abs(step).max())
