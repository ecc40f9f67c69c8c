"""Multi-GPU sharding of a batch of independent ascent NLPs.

The NLPs are independent, so the data path needs no collective: rank r of W solves problems
r, r+W, r+2W, ... (interleaved, so a sweep that is monotone in difficulty stays balanced) on its
own GPU.  The only exchange is the final gather of per-problem results to rank 0, one
torch.distributed collective (RCCL over xGMI when the backend is "nccl"; gloo on CPU in tests).
"""
from __future__ import annotations

import numpy as np

from .params import pack
from .solver import solve_batch, solve_batch_torch


def shard_indices(n: int, rank: int, world: int) -> np.ndarray:
    """Problem indices owned by `rank` (interleaved)."""
    return np.arange(rank, n, world)


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def solve_sharded(params, nt: int = 200, tol: float = 1e-9, max_iter: int = 300, device: int | None = None,
                  gather_traj: bool = False, solver=None, **kw):
    """Solve this rank's shard of `params` (every rank passes the same full array) and gather the
    results on rank 0.  Returns on rank 0 a dict(tf, status, iters[, traj]) in the original problem
    order; on other ranks None.  Without an initialised process group it solves everything locally.

    With the "nccl" backend (RCCL) and the default solver the shard stays in HBM end to end: parameters are uploaded
    once, `solve_batch_torch` leaves its results on the device and the gather reads them from there.  `solver` is a
    stand-in hook for CPU rehearsals of the sharding and the gather (gloo backend).  A rank with an empty shard (more
    ranks than problems) or a failing solver still joins the collective; the failure is raised on that rank and on
    rank 0 after the gather, so no rank is left blocked in it."""
    import torch
    P = pack(params)
    n = P.shape[0]
    dist = _dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
    if device is None:
        device = rank % max(1, torch.cuda.device_count()) if torch.cuda.is_available() else 0
    idx = shard_indices(n, rank, world)
    m = (n + world - 1) // world                       # padded shard length, equal on every rank
    width = 3 + (10 * nt if gather_traj else 0)
    k = len(idx)
    on_device = solver is None and dist is not None and dist.get_backend() == "nccl"
    failure = None
    if on_device:
        dev = torch.device("cuda", device)
        local = torch.zeros((m + 1, width), dtype=torch.float64, device=dev)   # last row: this rank's error flag
        if k:
            try:
                o = solve_batch_torch(torch.from_numpy(np.ascontiguousarray(P[idx])).to(dev), nt, tol=tol, max_iter=max_iter,
                                      want_traj=gather_traj, **kw)
                local[:k, 0], local[:k, 1], local[:k, 2] = o["tf"], o["status"].double(), o["iters"].double()
                if gather_traj:
                    local[:k, 3:] = o["traj"].permute(2, 0, 1).reshape(k, -1)
            except Exception as e:                     # noqa: BLE001 -- reported through the collective
                failure = e
                local[m, 0] = 1.0
    else:
        local = np.zeros((m + 1, width))
        if k:                                          # more ranks than problems: an empty shard contributes padding only
            try:
                res = (solver or solve_batch)(P[idx], nt=nt, tol=tol, max_iter=max_iter, device=device,
                                              want_traj=gather_traj, **kw)
                local[:k, 0], local[:k, 1], local[:k, 2] = res.tf, res.status, res.iters
                if gather_traj:
                    local[:k, 3:] = np.moveaxis(res.traj, 2, 0).reshape(k, -1)
            except Exception as e:                     # noqa: BLE001
                failure = e
                local[m, 0] = 1.0
    if not dist:
        if failure:
            raise failure
        gathered = [local]
    else:
        if on_device:
            t = local
        else:
            t = torch.from_numpy(local)
            if dist.get_backend() == "nccl":
                t = t.to(torch.device("cuda", device))
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, bufs, dst=0)                    # the single collective of the whole job
        if failure:
            raise failure
        if rank != 0:
            return None
        gathered = [b.cpu().numpy() for b in bufs]
        bad = [r for r, g in enumerate(gathered) if g[m, 0] != 0.0]
        if bad:
            raise RuntimeError(f"solve_sharded: the solver failed on rank(s) {bad}")
    out = np.zeros((n, width))
    for r, g in enumerate(gathered):
        ids = shard_indices(n, r, world)
        out[ids] = g[: len(ids)]
    result = dict(tf=out[:, 0], status=out[:, 1].astype(np.int32), iters=out[:, 2].astype(np.int32))
    if gather_traj:
        result["traj"] = np.moveaxis(out[:, 3:].reshape(n, 10, nt), 0, 2)
    return result
