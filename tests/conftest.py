import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def coracle():
    """The plain-C oracle (built on demand with gcc)."""
    from oracle import c_oracle
    c_oracle.build()
    return c_oracle


@pytest.fixture(scope="session")
def nominal_oracle_solution(coracle):
    from oracle.ascent_numpy import Params
    p16 = coracle.pack_params(Params())
    r = coracle.solve_batch(p16[None], 200, 300, 1e-9, want_blob=True)
    assert r["status"][0] == 0
    return p16, r


def random_interior_blob(nt, seed, p16, coracle):
    """A strictly interior primal-dual iterate: a few oracle IP iterations from the cold start,
    then multipliers perturbed (seeded)."""
    rng = np.random.default_rng(seed)
    r = coracle.solve_batch(p16[None], nt, 3 + seed % 4, 1e-9, want_blob=True, coarse_nodes=-1)
    blob = r["blob"][0].copy()
    K = nt - 1
    blob[8 * K:15 * K] += 0.05 * rng.standard_normal(7 * K)           # lambda
    blob[15 * K:21 * K] *= rng.uniform(0.7, 1.3, 6 * K)               # bound multipliers
    return blob


def generic_lu_newton_step(P, nt, blob, mu, dw, scheme=0, move_penalty=False):
    """The Newton step of the barrier problem at a primal-dual iterate (blob layout of include/ascent.h) from a generic
    sparse LU of the full KKT matrix assembled by the numpy oracle -- no stage structure, no Riccati recursion, no
    border elimination: the independent anchor for every stage-structured implementation (C oracle, HIP paths).
    Returns the step in the same blob layout (current formulation).  move_penalty: the NLP with the l1 move penalty
    (P.dcost; slack pairs and movement equations as explicit unknowns and rows of the generalised oracle -- nothing reduced);
    the slack pairs, which the blob does not carry, are set by the warm-start rule of include/ascent.h / c_oracle.newton_step."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from oracle.ascent_numpy import AscentNLP
    K = nt - 1
    if scheme == 2 or move_penalty:      # the generalised oracle (sympy-generated derivatives), same variable layout
        from oracle.ascent_general import GeneralNLP
        nlp = GeneralNLP(P, ((K, "burn"),), scheme, dcost=P.dcost if move_penalty else 0.0)
    else:
        nlp = AscentNLP(P, nt, 0, scheme=scheme)
    v = np.zeros(nlp.n); lam = np.zeros(nlp.m); zL = np.zeros(nlp.n); zU = np.zeros(nlp.n)
    Wk = v[:8 * K].reshape(K, 8)
    Wk[:, :7] = blob[:7 * K].reshape(K, 7); Wk[:, 7] = blob[7 * K:8 * K]
    lam[:7 * K] = blob[8 * K:15 * K]
    zb = blob[15 * K:21 * K].reshape(K, 6); sc = blob[21 * K:]
    base = np.arange(K) * 8
    zL[base + 4], zU[base + 4], zL[base + 6], zU[base + 6], zL[base + 7], zU[base + 7] = zb.T
    v[nlp.itf], zL[nlp.itf], zU[nlp.itf] = sc[0], sc[1], sc[2]
    v[nlp.is1], v[nlp.is2], zL[nlp.is1], zL[nlp.is2] = sc[3], sc[4], sc[5], sc[6]
    lam[-3], lam[-2], lam[-1] = sc[7], sc[8], sc[9]
    if move_penalty:
        du_ = np.diff(np.concatenate([[0.0], blob[7 * K:8 * K]]))
        v[nlp.ip], v[nlp.in_] = np.maximum(du_, 0.0) + 1e-4, np.maximum(-du_, 0.0) + 1e-4
        zL[nlp.ip] = zL[nlp.in_] = P.dcost
        lam[nlp.rmove] = 0.0
    hasL, hasU = np.isfinite(nlp.lb), np.isfinite(nlp.ub)
    dL = np.where(hasL, v - nlp.lb, 1.0); dU = np.where(hasU, nlp.ub - v, 1.0)
    c = nlp.constraints(v)
    J = nlp.jacobian(v); W = nlp.hessian(v, lam)
    Sig = np.where(hasL, zL / dL, 0) + np.where(hasU, zU / dU, 0)
    gphi = nlp.grad_objective(v) - np.where(hasL, mu / dL, 0) + np.where(hasU, mu / dU, 0)
    Kmat = sp.bmat([[W + sp.diags(Sig + dw), J.T], [J, None]], format="csc")
    sol = spla.splu(Kmat).solve(-np.concatenate([gphi + J.T @ lam, c]))
    dx, dlam = sol[:nlp.n], sol[nlp.n:]
    dzL = np.where(hasL, mu / dL - zL - zL / dL * dx, 0)
    dzU = np.where(hasU, mu / dU - zU + zU / dU * dx, 0)
    step = np.zeros_like(blob)
    dW = dx[:8 * K].reshape(K, 8)
    step[:7 * K] = dW[:, :7].ravel(); step[7 * K:8 * K] = dW[:, 7]
    step[8 * K:15 * K] = dlam[:7 * K]
    step[15 * K:21 * K] = np.stack([dzL[base + 4], dzU[base + 4], dzL[base + 6], dzU[base + 6], dzL[base + 7], dzU[base + 7]], 1).ravel()
    step[21 * K:] = [dx[nlp.itf], dzL[nlp.itf], dzU[nlp.itf], dx[nlp.is1], dx[nlp.is2], dzL[nlp.is1], dzL[nlp.is2],
                     dlam[-3], dlam[-2], dlam[-1]]
    return step, nlp, v, lam


def params_of_row(row):
    from oracle.ascent_numpy import Params
    from oracle.c_oracle import PARAM_FIELDS
    return Params(**{f: float(x) for f, x in zip(PARAM_FIELDS, row)})
