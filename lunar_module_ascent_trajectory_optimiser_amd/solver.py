"""Host-side API over libascent (include/ascent.h): the replacement of the reference's
`m.solve()` call (/root/reference/Launch_Optimiser.py:177) for batches of ascent NLPs."""
from __future__ import annotations

import ctypes as C
import dataclasses

import numpy as np

from . import _lib
from .params import AscentParams, pack

TRAJ_FIELDS = ("x", "y", "xdot", "ydot", "xdoubledot", "ydoubledot", "angle", "angledot",
               "angledoubledot", "mass")
STATUS_NAMES = {0: "converged", 1: "max_iter", 2: "linesearch_failed", 3: "regularisation_failed"}


def blob_rows(nt: int) -> int:
    return 21 * (nt - 1) + 10


SCHEMES = {"backward_euler": 0, "trapezoid": 1, "hermite_simpson": 2}
TERMINALS = {"reference": 0, "ellipse": 1, "ellipse_free": 2}


FORMULATIONS = {"current": 0, "v1": 1}


def _opts(nt, max_iter, tol, warm_start, mu_init, scheme=0, formulation=0, coarse_nodes=0, terminal=0, path="auto", move_penalty=False):
    scheme = SCHEMES.get(scheme, scheme)
    formulation = FORMULATIONS.get(formulation, formulation)
    terminal = TERMINALS.get(terminal, terminal)
    if path not in ("auto", "dense"):
        raise ValueError('solver path must be "auto" or "dense"')
    return _lib.AscentOptsC(n_nodes=nt, scheme=int(scheme), max_iter=max_iter, warm_start=warm_start, tol=tol,
                            mu_init=mu_init, formulation=int(formulation), coarse_nodes=int(coarse_nodes),
                            terminal=int(terminal), solver_path=_lib.PATHS[path], move_penalty=int(bool(move_penalty)), reserved=0)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclasses.dataclass
class BatchResult:
    """Solutions of a batch. Arrays keep the library's layout: problem index last."""
    params: np.ndarray          # (batch, 16)
    nt: int
    traj: np.ndarray            # (10, nt, batch) scaled, fields TRAJ_FIELDS (reference .value lists)
    tf: np.ndarray              # (batch,)  scaled final time (tf.value[0])
    status: np.ndarray          # (batch,) int32, see STATUS_NAMES
    iters: np.ndarray           # (batch,) int32
    blob: np.ndarray | None     # (21K+10, batch) primal-dual solution (warm-start material)
    kernel_ms: float            # device time of the solve kernel

    @property
    def converged(self) -> np.ndarray:
        return self.status == 0

    def field(self, name: str) -> np.ndarray:
        """(nt, batch) array of one of TRAJ_FIELDS, in the reference's scaled units."""
        return self.traj[TRAJ_FIELDS.index(name)]

    def final_time(self) -> np.ndarray:
        """seconds: tf.value[0]*final_time (Launch_Optimiser.py:194)."""
        return self.tf * self.params[:, 11]

    def orbit(self) -> dict:
        """Kepler-exact two-body orbit through every problem's final state (the reference's
        v1 script propagated the end state with explicit Euler, PDF p28-29; this is the closed form).  Returns
        periapsis / apoapsis altitudes above R0 (m), semi-major axis (m), eccentricity and flight-path angle (rad).
        Note for the reference's own target: it asks for the circular speed of the *mean* radius at the
        17.7 km insertion altitude (Launch_Optimiser.py:72-78), which is below the local circular speed."""
        P = self.params
        S, R0, GM = P[:, 9], P[:, 2], P[:, 0] * P[:, 1]
        X, Y = self.traj[0, -1] * S, self.traj[1, -1] * S + R0
        VX, VY = self.traj[2, -1] * S, self.traj[3, -1] * S
        r, v2 = np.hypot(X, Y), VX * VX + VY * VY
        a = 1.0 / (2.0 / r - v2 / GM)
        h = X * VY - Y * VX
        e = np.sqrt(np.maximum(0.0, 1.0 - h * h / (GM * a)))
        return dict(periapsis_alt=a * (1 - e) - R0, apoapsis_alt=a * (1 + e) - R0, semi_major_axis=a, eccentricity=e,
                    flight_path_angle=np.arcsin(np.clip((X * VX + Y * VY) / (r * np.sqrt(v2)), -1, 1)))

    def coast(self, coast_nodes: int = 200, device: int = 0) -> dict:
        """The second phase: coast from every problem's burnout state to the apoapsis of its orbit, propagated on the
        device (Kepler-exact; see coast_batch).  With terminal="ellipse" the arc ends at r_apo above the surface."""
        return coast_batch(self.params, np.ascontiguousarray(self.traj[:4, -1, :]), coast_nodes, device)

    def outputs(self, i: int = 0) -> dict:
        """The quantities the reference prints/plots for problem i (Launch_Optimiser.py:178-202):
        physical t, x_pos (sign flipped as :200), y_pos (:201), theta in degrees (:202), mass, and
        the polar (r, theta_polar) of north_star."""
        P = self.params[i]
        S, R0, T, M0, ms = P[9], P[2], P[11], P[4], P[7]
        tr = self.traj[:, :, i]
        x, y = tr[0], tr[1]
        X, Y = x * S, y * S + R0
        return dict(
            t=np.linspace(0.0, 1.0, self.nt) * self.tf[i] * T,
            x_pos=-X, y_pos=Y, theta_deg=3.0 * tr[6] * 180.0 / np.pi,
            r=np.hypot(X, Y), theta_polar=np.arctan2(-X, Y), control_angle=3.0 * tr[6],
            mass_kg=M0 - ms * tr[9],
            final_y=y[-1] * S, final_x=x[-1] * S, final_ydot=tr[3, -1] * S, final_xdot=tr[2, -1] * S,
            final_ydoubledot=tr[5, -1] * S, final_xdoubledot=tr[4, -1] * S,
            final_time=self.tf[i] * T, tf=self.tf[i],
        )


def solve_batch(params, nt: int = 200, tol: float = 1e-9, max_iter: int = 300, guess: np.ndarray | None = None,
                warm_start: int | None = None, mu_init: float = 0.0, device: int = 0, want_traj: bool = True,
                want_blob: bool = False, scheme=0, formulation=0, coarse_nodes: int = 0, terminal=0,
                path: str = "auto", move_penalty: bool = False) -> BatchResult:
    """Solve a batch of ascent NLPs on one GPU.  params: AscentParams | list | (batch,16) array.
    guess: (21K+10, batch) blob, with warm_start 1 (primal only) or 2 (primal-dual).
    scheme: 0 / "backward_euler" (the reference's NODES=2), 1 / "trapezoid" or 2 / "hermite_simpson" (both with the
    control held over the step; scheme 2 runs in a persistent kernel of its own, with the move penalty on the dense-block path).
    terminal: 0 / "reference" (Launch_Optimiser.py:72-78), 1 / "ellipse" (the (r_peri, r_apo) ellipse proper: vis-viva
    speed at its periapsis; `BatchResult.coast()` then ends at its apoapsis) or 2 / "ellipse_free" (burnout anywhere on that
    ellipse: its angular momentum and energy, no r.v = 0; the coast starts at whatever true anomaly the burn ends at).
    path: "auto" or "dense" (the dense-block path for any scheme).
    move_penalty: apply the reference's MV DCOST (Launch_Optimiser.py:99): objective tf + dcost * sum |u_k - u_{k-1}| with the
    `dcost` of each parameter set (schemes 0 / 1: inside the persistent kernel, the control as the eighth state of a stage;
    scheme 2: dense-block path; default off: `dcost` is then ignored).
    formulation: 0 / "current" or 1 / "v1" (the PDF appendix script: the angle is the MV; see include/ascent.h).
    coarse_nodes: nested iteration for cold starts (0 automatic, -1 single grid, > 0 explicit coarse grid);
    `iters` then counts the iterations of all grid levels."""
    L = _lib.load()
    P = pack(params)
    B = P.shape[0]
    rows = blob_rows(nt)
    if guess is not None:
        guess = np.ascontiguousarray(guess, dtype=np.float64)
        if guess.shape != (rows, B):
            raise ValueError(f"guess must have shape {(rows, B)}")
        if warm_start is None:
            warm_start = 1
    warm_start = warm_start or 0
    traj = np.empty((10, nt, B)) if want_traj else None
    blob = np.empty((rows, B)) if want_blob else None
    tf = np.empty(B)
    status = np.empty(B, dtype=np.int32)
    iters = np.empty(B, dtype=np.int32)
    o = _opts(nt, max_iter, tol, warm_start, mu_init, scheme, formulation, coarse_nodes, terminal, path, move_penalty)
    _lib.check(L.ascent_solve_batch(_ptr(P), B, C.byref(o), _ptr(guess), _ptr(traj), _ptr(tf), _ptr(status),
                                    _ptr(iters), _ptr(blob), device, None, 0))
    return BatchResult(P, nt, traj, tf, status, iters, blob, L.ascent_last_kernel_ms(device))


def eval_nodes(params, iterate: np.ndarray, nt: int = 200, device: int = 0, path="auto", scheme=0, formulation=0):
    """Per-step defects (7K,batch), Jacobian blocks (8K,batch), Hessian blocks (10K,batch).
    path: "auto" (the kernels solve_batch would run for this batch), "fused", "split_lane", "split_wide"
    (enum ascent_path, include/ascent.h); the split paths take scheme 1 and formulation 1 as well."""
    L = _lib.load()
    P = pack(params)
    B, K = P.shape[0], nt - 1
    it = np.ascontiguousarray(iterate, dtype=np.float64)
    if it.shape != (blob_rows(nt), B):
        raise ValueError("iterate has the wrong shape")
    d, j, h = np.empty((7 * K, B)), np.empty((8 * K, B)), np.empty((10 * K, B))
    o = _opts(nt, 0, 1.0, 0, 0.0, scheme, formulation)
    _lib.check(L.ascent_eval_nodes_path(_ptr(P), B, C.byref(o), _ptr(it), _ptr(d), _ptr(j), _ptr(h), device,
                                        _lib.PATHS[path]))
    return d, j, h


def kkt_step(params, iterate: np.ndarray, mu, delta_w, nt: int = 200, device: int = 0, path="auto", scheme=0,
             formulation=0, terminal=0, move_penalty: bool = False):
    """One Newton step of the barrier problem at `iterate` -> (step blob, inertia flags); `path` as in eval_nodes.
    move_penalty (paths "persist" and "dense"): with the l1 move penalty; the slack pairs, which the blob does not carry, are
    set around the iterate's own movement (p = max(du, 0) + 1e-4, n = max(-du, 0) + 1e-4, z_p = z_n = dcost, lambda_u = 0)."""
    L = _lib.load()
    P = pack(params)
    B = P.shape[0]
    it = np.ascontiguousarray(iterate, dtype=np.float64)
    if it.shape != (blob_rows(nt), B):
        raise ValueError("iterate has the wrong shape")
    mu = np.ascontiguousarray(np.broadcast_to(np.asarray(mu, dtype=np.float64), (B,)))
    dw = np.ascontiguousarray(np.broadcast_to(np.asarray(delta_w, dtype=np.float64), (B,)))
    step = np.empty_like(it)
    inertia = np.empty(B, dtype=np.int32)
    o = _opts(nt, 0, 1.0, 0, 0.0, scheme, formulation, terminal=terminal, move_penalty=move_penalty)
    _lib.check(L.ascent_kkt_step_path(_ptr(P), B, C.byref(o), _ptr(it), _ptr(mu), _ptr(dw), _ptr(step), _ptr(inertia),
                                      device, _lib.PATHS[path]))
    return step, inertia


def dense_records(params, iterate: np.ndarray, nt: int = 200, scheme=2, device: int = 0) -> np.ndarray:
    """The dense stage records of the dense-block path at `iterate`: (batch, K, 6, 8, 8) -- grids d c_k/d z_{k-1}, d c_k/d z_k,
    the three Hessian blocks of lambda_k'c_k and the vector grid (rows c_k, d c_k/du_k, d c_k/d tf, the two (z, tf) Hessian
    columns); see ascent_dense_records in include/ascent.h."""
    L = _lib.load()
    P = pack(params)
    B, K = P.shape[0], nt - 1
    it = np.ascontiguousarray(iterate, dtype=np.float64)
    if it.shape != (blob_rows(nt), B):
        raise ValueError("iterate has the wrong shape")
    rec = np.empty((B, K, 6, 8, 8))
    o = _opts(nt, 0, 1.0, 0, 0.0, scheme)
    _lib.check(L.ascent_dense_records(_ptr(P), B, C.byref(o), _ptr(it), _ptr(rec), device))
    return rec


def coast_batch(params, final_state: np.ndarray, coast_nodes: int = 200, device: int = 0) -> dict:
    """Kepler-exact coast arc from every problem's burnout state (4, batch: scaled x, y, xdot, ydot) to the next apoapsis of
    its orbit (ascent_coast_batch): dict(traj (4, coast_nodes+1, batch), tf (batch,) = duration / T_scale,
    periapsis_alt, apoapsis_alt (m))."""
    L = _lib.load()
    P = pack(params)
    B = P.shape[0]
    fs = np.ascontiguousarray(final_state, dtype=np.float64)
    if fs.shape != (4, B):
        raise ValueError(f"final_state must have shape {(4, B)}")
    traj = np.empty((4, coast_nodes + 1, B)); tf = np.empty(B); aps = np.empty((2, B))
    _lib.check(L.ascent_coast_batch(_ptr(P), B, _ptr(fs), coast_nodes, _ptr(traj), _ptr(tf), _ptr(aps), device, None, 0))
    return dict(traj=traj, tf=tf, periapsis_alt=aps[0], apoapsis_alt=aps[1])


def kkt_solve(diag, lower, upper, rhs, border=None, border_diag=None, algo="pcr", device: int = 0):
    """Generic bordered block-tridiagonal solve on the GPU (ascent_kkt_solve, include/ascent.h).
    diag / lower / upper: (batch, n, bs, bs); rhs: (batch, n*bs + nb); border: (batch, n, bs, nb); border_diag: (batch, nb, nb).
    algo: "thomas" (block elimination serial in the node index) or "pcr" (parallel cyclic reduction over the nodes).
    Returns (solution (batch, n*bs + nb), device milliseconds of the solve)."""
    L = _lib.load()
    D = np.ascontiguousarray(diag, dtype=np.float64)
    B, n, bs, _ = D.shape
    Lo = np.ascontiguousarray(lower, dtype=np.float64); Up = np.ascontiguousarray(upper, dtype=np.float64)
    nb = 0 if border is None else int(np.shape(border)[-1])
    r = np.ascontiguousarray(rhs, dtype=np.float64)
    if Lo.shape != D.shape or Up.shape != D.shape or r.shape != (B, n * bs + nb):
        raise ValueError("inconsistent shapes")
    bo = bd = None
    if nb:
        bo = np.ascontiguousarray(border, dtype=np.float64); bd = np.ascontiguousarray(border_diag, dtype=np.float64)
        if bo.shape != (B, n, bs, nb) or bd.shape != (B, nb, nb):
            raise ValueError("inconsistent border shapes")
    sol = np.empty_like(r)
    _lib.check(L.ascent_kkt_solve(B, n, bs, nb, _ptr(D), _ptr(Lo), _ptr(Up), _ptr(bo), _ptr(bd), _ptr(r), _ptr(sol), device,
                                  {"thomas": 0, "pcr": 1}[algo]))
    return sol, L.ascent_last_kernel_ms(device)


def solve_batch_torch(params_t, nt: int = 200, tol: float = 1e-9, max_iter: int = 300, guess_t=None,
                      warm_start: int = 0, mu_init: float = 0.0, want_traj: bool = True, want_blob: bool = False,
                      out: dict | None = None, sync: bool = False, coarse_nodes: int = 0, scheme=0,
                      formulation=0, move_penalty: bool = False, terminal=0, path: str = "auto") -> dict:
    """Device-resident variant: `params_t` is a torch float64 CUDA tensor (batch,16); all outputs are
    torch CUDA tensors (allocated here unless passed in `out`).  Enqueues on torch's current stream
    and returns without waiting unless sync=True.  torch is only the owner of device memory/streams.
    Options as solve_batch (scheme, formulation, terminal, path, move_penalty: the weights params_t[:, 15] must be
    positive then -- checked here on the device, the library cannot look into device memory from the host)."""
    import torch
    L = _lib.load()
    _lib.require_single_hip_runtime()
    if not (params_t.is_cuda and params_t.dtype == torch.float64 and params_t.is_contiguous()):
        raise ValueError("params_t must be a contiguous float64 CUDA tensor")
    B = params_t.shape[0]
    dev = params_t.device
    rows = blob_rows(nt)
    out = out if out is not None else {}
    def buf(name, shape, dtype):
        t = out.get(name)
        if t is None:
            t = out[name] = torch.empty(shape, dtype=dtype, device=dev)
        elif not (t.device == dev and t.dtype == dtype and tuple(t.shape) == tuple(shape) and t.is_contiguous()):
            raise ValueError(f"out[{name!r}] must be a contiguous {dtype} tensor of shape {tuple(shape)} on {dev}")
        return t
    if params_t.dim() != 2 or params_t.shape[1] != 16:
        raise ValueError("params_t must have shape (batch, 16)")
    if move_penalty and not bool((params_t[:, 15] > 0).all()):
        raise ValueError("move_penalty needs dcost > 0 (column 15 of params_t) for every problem")
    if warm_start not in (0, 1, 2) or (warm_start and guess_t is None):
        raise ValueError("warm_start 1/2 needs guess_t")
    if guess_t is not None and not (guess_t.device == dev and guess_t.dtype == torch.float64 and guess_t.is_contiguous()
                                    and tuple(guess_t.shape) == (rows, B)):
        raise ValueError(f"guess_t must be a contiguous float64 tensor of shape {(rows, B)} on {dev}")
    tf = buf("tf", (B,), torch.float64)
    status = buf("status", (B,), torch.int32)
    iters = buf("iters", (B,), torch.int32)
    traj = buf("traj", (10, nt, B), torch.float64) if want_traj else None
    blob = buf("blob", (rows, B), torch.float64) if want_blob else None
    o = _opts(nt, max_iter, tol, warm_start, mu_init, scheme, formulation, coarse_nodes, terminal, path, move_penalty)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(L.ascent_solve_batch(params_t.data_ptr(), B, C.byref(o),
                                    guess_t.data_ptr() if guess_t is not None else None,
                                    traj.data_ptr() if traj is not None else None, tf.data_ptr(), status.data_ptr(),
                                    iters.data_ptr(), blob.data_ptr() if blob is not None else None,
                                    dev.index or 0, C.c_void_p(stream), 1))
    if sync:
        torch.cuda.synchronize(dev)
    return out


def default_path(batch: int, nt: int = 200, scheme=0, formulation=0, move_penalty: bool = False, terminal=0) -> str:
    """The kernels solve_batch runs for a batch of this size (include/ascent.h: ascent_default_path): a key of _lib.PATHS."""
    o = _opts(nt, 300, 1e-9, 0, 0.0, scheme, formulation, move_penalty=move_penalty, terminal=terminal)
    code = _lib.load().ascent_default_path(int(batch), C.byref(o))
    return {v: k for k, v in _lib.PATHS.items()}[code]


def last_kernel_ms(device: int = 0) -> float:
    """HIP-event time of the most recent solve kernel on `device` (waits for it)."""
    return _lib.load().ascent_last_kernel_ms(device)
