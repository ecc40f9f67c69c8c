"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/libascent_oracle.so (plain-C oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libascent_oracle.so")
PARAM_FIELDS = ("G", "M", "R0", "Ft", "M0", "mdot", "fuel_mass", "mass_scalar", "ang_acc_max",
                "r_peri", "r_apo", "T_scale", "angle_ub", "tf_lb", "tf_ub", "dcost")
TRAJ_FIELDS = ("x", "y", "xdot", "ydot", "xdoubledot", "ydoubledot", "angle", "angledot",
               "angledoubledot", "mass")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    if not os.path.exists(_LIB):
        build()
    L = C.CDLL(_LIB)
    L.oracle_kkt_error.restype = C.c_double
    return L


def set_scheme(scheme: int):
    """0 = backward Euler (reference NODES=2), 1 = trapezoid with zero-order-hold control.  Thread-local in the
    library; every call below that takes `scheme=` sets it first."""
    lib().oracle_set_scheme(int(scheme))


def set_formulation(form: int):
    """0 = current script (u = angledoubledot is the MV), 1 = v1 script (the angle is the MV; embedded in the
    same state layout, see ascent_oracle.c).  Thread-local in the library."""
    lib().oracle_set_formulation(int(form))


def pack_params(P) -> np.ndarray:
    """Params dataclass (oracle.ascent_numpy.Params) or dict -> 16 doubles."""
    get = (lambda k: P[k]) if isinstance(P, dict) else (lambda k: getattr(P, k))
    return np.array([get(k) for k in PARAM_FIELDS], dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def blob_size(nt):
    return lib().oracle_blob_size(nt)


def newton_step(params16, nt, blob, mu, delta_w, scheme=0, formulation=0, move_penalty=False):
    """One Newton step of the barrier problem at `blob`.  move_penalty: with the l1 move penalty (LO:99); the slack pairs,
    which the external blob does not carry, are set around the iterate's own movement (p = max(du, 0) + 1e-4,
    n = max(-du, 0) + 1e-4, z_p = z_n = dcw, lambda_u = 0), as every warm start sets them."""
    L = lib()
    L.oracle_set_scheme(int(scheme))
    L.oracle_set_formulation(int(formulation))
    L.oracle_set_move_penalty(int(bool(move_penalty)))
    step = np.zeros_like(blob)
    try:
        rc = L.oracle_newton_step(_p(params16), nt, _p(blob), C.c_double(mu), C.c_double(delta_w), _p(step))
    finally:
        L.oracle_set_move_penalty(0)
    return rc, step


def constraints(params16, nt, blob, scheme=0):
    lib().oracle_set_scheme(int(scheme))
    c = np.zeros(7 * (nt - 1) + 3)
    lib().oracle_constraints(_p(params16), nt, _p(blob), _p(c))
    return c


def kkt_error(params16, nt, blob, mu=0.0, scheme=0):
    lib().oracle_set_scheme(int(scheme))
    return lib().oracle_kkt_error(_p(params16), nt, _p(blob), C.c_double(mu))


def accel(params16, x, y, a, m, px, py):
    n = len(x)
    arrs = [np.ascontiguousarray(v, dtype=np.float64) for v in (x, y, a, m, px, py)]
    ax, ay = np.zeros(n), np.zeros(n)
    gax, gay, H = np.zeros((n, 4)), np.zeros((n, 4)), np.zeros((n, 10))
    lib().oracle_accel(_p(params16), n, *[_p(v) for v in arrs], _p(ax), _p(ay), _p(gax), _p(gay), _p(H))
    return ax, ay, gax, gay, H


def solve_batch(params, nt=200, max_iter=300, tol=1e-9, guess_blob=None, want_blob=False, scheme=0, formulation=0,
                coarse_nodes=0, warm_start=1, mu_init=0.0, move_penalty=False):
    """params: (batch,16).  Returns dict(traj (batch,10,nt), tf, status, iters[, blob]).
    Cold starts use the nested iteration (coarse_nodes: 0 automatic, -1 single grid, > 0 explicit coarse grid);
    with guess_blob (batch, blob) it is a warm start (1 primal, 2 primal-dual; mu_init <= 0: 1e-4).
    iters counts the iterations of all grid levels.  move_penalty: the reference's MV DCOST (LO:99) as an l1 term with the
    `dcost` of each parameter row (off: dcost is ignored)."""
    lib().oracle_set_move_penalty(int(bool(move_penalty)))
    lib().oracle_set_scheme(int(scheme))
    lib().oracle_set_formulation(int(formulation))
    lib().oracle_set_coarse_nodes(int(coarse_nodes))
    lib().oracle_set_warm_start(int(warm_start), C.c_double(mu_init))
    params = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
    B = params.shape[0]
    traj = np.zeros((B, 10, nt))
    tf = np.zeros(B)
    status = np.zeros(B, dtype=np.int32)
    iters = np.zeros(B, dtype=np.int32)
    bs = blob_size(nt)
    blob = np.zeros((B, bs)) if want_blob else None
    g = None if guess_blob is None else np.ascontiguousarray(guess_blob, dtype=np.float64)
    lib().oracle_solve_batch(_p(params), B, nt, max_iter, C.c_double(tol),
                             _p(g) if g is not None else None, _p(traj), _p(tf),
                             status.ctypes.data_as(_ip), iters.ctypes.data_as(_ip),
                             _p(blob) if blob is not None else None)
    lib().oracle_set_move_penalty(0)
    out = dict(traj=traj, tf=tf, status=status, iters=iters)
    if want_blob:
        out["blob"] = blob
    return out


def prolong(blob_c, nt_c, nt_f, formulation=0):
    """Prolongation of one primal-dual blob from an nt_c-node grid to an nt_f-node grid (nested iteration)."""
    lib().oracle_set_formulation(int(formulation))
    bc = np.ascontiguousarray(blob_c, dtype=np.float64)
    bf = np.zeros(blob_size(nt_f))
    lib().oracle_prolong(_p(bc), int(nt_c), _p(bf), int(nt_f))
    return bf
