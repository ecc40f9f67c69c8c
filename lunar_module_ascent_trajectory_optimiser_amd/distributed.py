"""Multi-GPU sharding of a batch of independent ascent NLPs.

The NLPs are independent, so the data path needs no collective: rank r of W solves problems
r, r+W, r+2W, ... (interleaved, so a sweep that is monotone in difficulty stays balanced) on its
own GPU.  The only exchange is the final gather of per-problem results to rank 0, one
torch.distributed collective (RCCL over xGMI when the backend is "nccl"; gloo on CPU in tests).
"""
from __future__ import annotations

import numpy as np

from .params import pack
from .solver import solve_batch


def shard_indices(n: int, rank: int, world: int) -> np.ndarray:
    """Problem indices owned by `rank` (interleaved)."""
    return np.arange(rank, n, world)


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def solve_sharded(params, nt: int = 200, tol: float = 1e-9, max_iter: int = 300, device: int | None = None,
                  gather_traj: bool = False, solver=solve_batch, **kw):
    """Solve this rank's shard of `params` (every rank passes the same full array) and gather the
    results on rank 0.  Returns on rank 0 a dict(tf, status, iters[, traj]) in the original problem
    order; on other ranks None.  Without an initialised process group it solves everything locally."""
    import torch
    P = pack(params)
    n = P.shape[0]
    dist = _dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
    if device is None:
        device = rank % max(1, torch.cuda.device_count()) if torch.cuda.is_available() else 0
    idx = shard_indices(n, rank, world)
    res = solver(P[idx], nt=nt, tol=tol, max_iter=max_iter, device=device, want_traj=gather_traj, **kw)
    m = (n + world - 1) // world                       # padded shard length, equal on every rank
    width = 3 + (10 * nt if gather_traj else 0)
    local = np.zeros((m, width))
    k = len(idx)
    local[:k, 0], local[:k, 1], local[:k, 2] = res.tf, res.status, res.iters
    if gather_traj:
        local[:k, 3:] = np.moveaxis(res.traj, 2, 0).reshape(k, -1)
    if not dist:
        gathered = [local]
    else:
        backend = dist.get_backend()
        dev = torch.device("cuda", device) if backend == "nccl" else torch.device("cpu")
        t = torch.from_numpy(local).to(dev)
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, bufs, dst=0)                    # the single collective of the whole job
        if rank != 0:
            return None
        gathered = [b.cpu().numpy() for b in bufs]
    out = np.zeros((n, width))
    for r, g in enumerate(gathered):
        ids = shard_indices(n, r, world)
        out[ids] = g[: len(ids)]
    result = dict(tf=out[:, 0], status=out[:, 1].astype(np.int32), iters=out[:, 2].astype(np.int32))
    if gather_traj:
        result["traj"] = np.moveaxis(out[:, 3:].reshape(n, 10, nt), 0, 2)
    return result
