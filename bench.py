#!/usr/bin/env python3
"""Headline benchmark: converged N=200 ascent NLPs per second (BASELINE.json metric).

A "step" = one pass of the hot path over one batch per GPU: `batch_per_gpu` independent ascent NLPs
solved from the built-in cold start to KKT error <= tol, parameters already resident in HBM, results
left in HBM, then the single result gather (tf, status, iters) to rank 0.
Workload at N=1: BASELINE.json configs[2] -- the 4096-NLP Isp x dry-mass sweep of the model the reference
declares, i.e. WITH its MV move penalty angledoubledot.DCOST = 1e-5 (Launch_Optimiser.py:99; --no-dcost times the
unpenalised NLP as the headline instead; the other variant is always reported beside it as a secondary line).
With N ranks (weak scaling) every rank solves that same sweep, so the N=1 and N=8 lines time the same per-GPU work;
--batch-per-gpu 32768 with --gpus 8 is config 4 itself (rank r = its contiguous shard of the 262 144-problem box).

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic HBM bytes of the solve kernel
(B_alg = B_io + I*B_iter per NLP, I = iterations actually taken) / its HIP-event time;
`cpu_baseline` = the plain-C oracle on the host cores for a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NT = 200
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # MI355X FP64 vector = matrix
APO_KM = np.linspace(70.0, 105.0, 8)


def algorithmic_bytes(levels, n_problems: int, nt: int) -> float:
    """B_io = params + full trajectory + (tf,status,iters); B_iter = 21 doubles read +
    21 written per node per interior-point iteration.  levels = [(nodes, total iterations on that grid), ...]:
    the iterations of the nested iteration's coarse grid are priced at the coarse grid's size."""
    b_io = 16 * 8 + 10 * nt * 8 + 16
    return float(n_problems * b_io + sum(it * 2 * 21 * 8 * n for n, it in levels))


def algorithmic_flops(levels) -> float:
    """SURVEY.md 8d work-optimal count: ~4e3 flop per node per iteration."""
    return float(sum(it * 4e3 * n for n, it in levels))


def nested_levels(nt: int) -> list:
    """The grids of the automatic nested iteration, finest first (include/ascent.h: ascent_opts.coarse_nodes)."""
    lv = [nt]
    while lv[-1] >= 40:
        c = max(14, (3 * lv[-1] + 5) // 10)
        if c > 17 and 1 <= (c - 1) % 16 <= 3:
            c -= (c - 1) % 16
        if c >= lv[-1]:
            break
        lv.append(c)
    return lv


def rank_params(batch: int, rank: int, world: int, dcost: float = 0.0) -> np.ndarray:
    import lunar_module_ascent_trajectory_optimiser_amd as A
    if batch == 32768 and world == 8:               # exact config-4 shard
        full = A.sweep_config4()
        P = np.ascontiguousarray(full[rank * batch:(rank + 1) * batch])
    else:
        n = int(round(batch ** 0.5))
        while batch % n:
            n -= 1
        P = A.sweep_isp_drymass(n, batch // n)      # 4096 -> the 64 x 64 grid of config 3, the same on every rank
    P[:, 15] = dcost                                # the MV's DCOST (Launch_Optimiser.py:99); used with move_penalty only
    return P


def source_sha16() -> str:
    """Fingerprint of the kernel sources: ties a PMC traffic measurement to the build it was made on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "lunar_module_ascent_trajectory_optimiser_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, fn), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "ascent.h"), "rb").read())
    return h.hexdigest()[:16]


def usable_cores() -> int:
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return int(os.environ.get("ASCENT_CPU_THREADS", n))


def cpu_baseline(params: np.ndarray, nt: int, tol: float, sample: int, move_penalty: bool = False):
    """Times the plain-C oracle (oracle/ascent_oracle.c, kind "port") on all host cores."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle
    c_oracle.build()
    cores = usable_cores()
    idx = np.linspace(0, len(params) - 1, sample).astype(int)
    S = np.ascontiguousarray(params[idx])
    chunks = np.array_split(np.arange(sample), cores * 4)
    c_oracle.solve_batch(S[:1], nt, 300, tol, move_penalty=move_penalty)      # load + warm
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:           # ctypes releases the GIL during the C call
        res = list(ex.map(lambda ch: c_oracle.solve_batch(S[ch], nt, 300, tol, move_penalty=move_penalty), [c for c in chunks if len(c)]))
    dt = time.perf_counter() - t0
    ok = sum(int((r["status"] == 0).sum()) for r in res)
    return dict(value=ok / dt, unit="NLPs/s", cores=cores, kind="port",
                sample=f"{sample} of the {len(params)} NLPs (every {len(params)//sample}th), same tol and cold start, "
                       f"{dt:.2f} s wall on {cores} threads; CPU restatement, not GEKKO/IPOPT"), idx, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=4096)
    ap.add_argument("--tol", type=float, default=1e-9)
    ap.add_argument("--cpu-sample", type=int, default=4096)
    ap.add_argument("--dcost", type=float, default=1e-5, help="weight of the MV move penalty (the reference's DCOST, Launch_Optimiser.py:99)")
    ap.add_argument("--no-dcost", action="store_true", help="headline on the unpenalised NLP (round 1/2's workload); the penalised one becomes the secondary line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-continuation", action="store_true", help="skip the secondary warm-start (continuation) measurement")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the secondary two-stream (overlapped batches) measurement")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the secondary lines for BASELINE configs 1, 4 (one shard) and 5")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a CPU rehearsal of the gather)")
    args = ap.parse_args()

    import torch
    import lunar_module_ascent_trajectory_optimiser_amd as A

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")      # where collectives run

    B = args.batch_per_gpu
    mp = not args.no_dcost and args.dcost > 0.0            # the headline's model: with the reference's move penalty
    P = rank_params(B, rank, world, args.dcost)
    P_t = torch.from_numpy(P).to(dev)                      # inputs resident in HBM before timing
    out = {}
    gathered = [torch.empty((B, 3), dtype=torch.float64, device=cdev) for _ in range(world)] if (dist and rank == 0) else None

    # The timed steps are enqueued on a stream of their own: with a caller stream and device pointers the library only enqueues
    # its launches (include/ascent.h; on the default stream it would wait for every solve), so the launches of step i+1 are
    # queued while step i runs.  The device time of every step is taken from HIP events recorded on that same stream around
    # the solve's launches and read after the closing synchronisation.
    main_stream = torch.cuda.Stream(dev)
    step_events = []

    def step(timed=False):
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_stream)
        A.solve_batch_torch(P_t, NT, tol=args.tol, want_traj=True, out=out, move_penalty=mp)
        if timed:
            e1.record(main_stream)
            step_events.append((e0, e1))
        if dist:                                           # the job's only collective: result gather
            pack = torch.stack([out["tf"], out["status"].double(), out["iters"].double()], dim=1).to(cdev)
            dist.gather(pack, gathered, dst=0)

    def barrier():
        if dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed_loop(fn, steps):
        """`steps` calls of fn enqueued on main_stream between two synchronisations: (wall ms per step, mean device ms per step)"""
        evs = []
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        with torch.cuda.stream(main_stream):
            for _ in range(steps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(main_stream); fn(); b.record(main_stream)
                evs.append((a, b))
        torch.cuda.synchronize(dev)
        wall = (time.perf_counter() - t1) / steps * 1e3
        return wall, float(np.mean([a.elapsed_time(b) for a, b in evs]))

    conv_steps = torch.zeros((), dtype=torch.int64, device=dev)   # converged NLPs summed over the timed steps (counted on the device)
    with torch.cuda.stream(main_stream):
        for _ in range(max(args.warmup, 1) if args.warmup else 0):
            step()
            conv_steps += (out["status"] == 0).sum()                  # (also warms up torch's own reduction kernel)
        if not args.warmup:
            conv_steps += torch.zeros((), dtype=torch.int64, device=dev)
        conv_steps.zero_()
    barrier()
    t0 = time.perf_counter()
    with torch.cuda.stream(main_stream):
        for _ in range(args.steps):
            step(timed=True)
            conv_steps += (out["status"] == 0).sum()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [e0.elapsed_time(e1) for e0, e1 in step_events]       # device time of each timed step's launches (all grid levels)
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    iters = out["iters"].cpu().numpy()
    status = out["status"].cpu().numpy()
    conv_local = int((status == 0).sum())
    conv_total, conv_all_steps = conv_local, int(conv_steps.item())
    if dist:
        t = torch.tensor([conv_local, conv_all_steps], dtype=torch.float64, device=cdev)
        dist.all_reduce(t)
        conv_total, conv_all_steps = int(t[0].item()), int(t[1].item())
    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        # iterations per grid level: `iters` counts all levels of the nested iteration; untimed solves of the coarser
        # grids alone (the same first legs, deterministic, at the coarse legs' tolerance) give the split
        lv = nested_levels(NT)
        cum = [iters.astype(np.int64)]
        for n in lv[1:]:
            sub = A.solve_batch_torch(P_t, n, tol=max(args.tol, 1e-3), want_traj=False, sync=True, move_penalty=mp)
            cum.append(sub["iters"].cpu().numpy().astype(np.int64))
        cum.append(np.zeros_like(cum[0]))
        per_level = [cum[i] - cum[i + 1] for i in range(len(lv))]
        levels = [(n, float(it.sum())) for n, it in zip(lv, per_level)]
        b_alg = algorithmic_bytes(levels, len(iters), NT)
        f_alg = algorithmic_flops(levels)
        achieved = b_alg / (k_ms * 1e-3) / 1e9
        # HBM bytes per solve from the PMC counters (profiles/traffic.json, made by scripts/traffic_from_pmc.py): only if that
        # file was measured on THIS build of the kernels (hash of the csrc sources), otherwise null
        def traffic_of(key):
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath)).get(key, {})
                if tj.get("source_sha16") == source_sha16():
                    return tj.get("hbm_bytes_per_launch")
            return None
        traffic = traffic_of("config3" if mp else "config3_nodcost") if B == 4096 else None

        def roofline_of(P_, nt_, res, ms, key=None, **kw):
            """roofline object of a secondary configuration: algorithmic bytes from the iterations taken on every grid level (untimed
            solves of the coarser grids alone give the split), device time of one solve, PMC traffic of profiles/traffic.json[key]"""
            lv_ = nested_levels(nt_)
            cum_ = [res.iters.astype(np.int64)]
            for n_ in lv_[1:]:
                cum_.append(A.solve_batch(P_, n_, tol=max(args.tol, 1e-3), want_traj=False, **kw).iters.astype(np.int64))
            cum_.append(np.zeros_like(cum_[0]))
            levels_ = [(n_, float((cum_[i] - cum_[i + 1]).sum())) for i, n_ in enumerate(lv_)]
            b_, f_ = algorithmic_bytes(levels_, len(res.iters), nt_), algorithmic_flops(levels_)
            ach = b_ / (ms * 1e-3) / 1e9
            return {"bound": "hbm", "binding": "fp64_issue", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic_of(key) if key else None, "kernel_ms": ms, "algorithmic_bytes_per_launch": b_,
                    "fp64_frac": f_ / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                    "iterations_mean_by_grid": {str(n_): it_ / len(res.iters) for n_, it_ in levels_}}
        line = {
            "metric": "solved ascent NLPs/sec (N=200 collocation nodes)",
            "value": conv_all_steps / elapsed,
            "unit": "NLPs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": ("BASELINE.json configs[2]: 4096-NLP Isp x dry-mass sweep (64x64, Isp 300-320 s, dry mass 2345-2545 kg)"
                             if B == 4096 else f"{B}-NLP sweep per GPU") + ", N=200 nodes, backward Euler (reference NODES=2), "
                            + (f"DCOST applied (the script's MV move penalty {args.dcost:g}, Launch_Optimiser.py:99, as an l1 term: the model the reference declares)"
                               if mp else "DCOST not applied (the unpenalised NLP; the script's MV move penalty, Launch_Optimiser.py:99, switched off)"),
                "move_penalty": bool(mp), "dcost": args.dcost if mp else 0.0,
                "batch_per_gpu": B, "global_batch": B * world, "n_nodes": NT, "tol": args.tol,
                "start": f"cold (built-in straight-line guess, mu0=0.1) on a {lv[-1]}-node grid, prolonged grid by grid ({' -> '.join(str(n) for n in lv[::-1])} nodes) "
                         "(nested iteration, part of the solver and of the timed step)",
                "parallelism": f"problem-sharded x{world}, gather only",
                "iterations_min_mean_max": [int(iters.min()), float(iters.mean()), int(iters.max())],
                "iterations_mean_by_grid": {str(n): float(it.mean()) for n, it in zip(lv, per_level)},
                "converged": conv_total, "of": B * world, "converged_over_all_timed_steps": conv_all_steps,
                "kernel_source_sha16": source_sha16(),
            },
            "roofline": {
                "bound": "hbm", "binding": "fp64_issue", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": (("p_solve (persistent kernel: one wavefront owns four NLPs for a whole grid level; one launch per grid level, "
                            f"{len(lv)} levels) + p_init / p_transfer / p_finish") if A.default_path(B, NT, move_penalty=mp) == "persist" else
                           "k_solve (fused, one lane per NLP), one launch per grid level" if A.default_path(B, NT, move_penalty=mp) == "fused" else
                           A.default_path(B, NT, move_penalty=mp)),
                "kernel_ms": k_ms, "algorithmic_bytes_per_launch": b_alg,
                "fp64_achieved_tflops": f_alg / (k_ms * 1e-3) / 1e12,
                "fp64_frac": f_alg / (k_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                "algorithmic_bytes_note": "SURVEY 8d's 21 doubles per node read + written per iteration; the penalised NLP carries 26 (lambda_u and the slack pair with its multipliers), not counted",
                "note": ("kernel_ms = device time of one whole solve (HIP events around all launches of the step; p_solve is 97 % of it: "
                         "profiles/); achieved = SURVEY 8d algorithmic bytes of the solve (iterations x nodes of every grid level x 2 x 21 "
                         "doubles + i/o) / that time. The kernel is FP64-issue bound, not HBM bound: one wavefront per SIMD issues one "
                         "instruction per 4 cycles, and the serial sweeps keep 10 of 16 lanes per NLP busy: see DESIGN.md section 5"),
            },
        }
        if world == 1 and not args.no_continuation:
            # Secondary, differently-defined number (never `value`): the same sweep under a continuation policy --
            # every NLP warm-started (primal-dual, mu0 = 1e-6) from the solution of the nominal Apollo-11 problem.
            nom = A.solve_batch(A.AscentParams(tf_ub=float(P[0, 14]), dcost=args.dcost), NT, tol=args.tol, want_blob=True, move_penalty=mp)
            g_t = torch.from_numpy(np.ascontiguousarray(np.repeat(nom.blob, B, axis=1))).to(dev)
            wout = {}
            A.solve_batch_torch(P_t, NT, tol=args.tol, guess_t=g_t, warm_start=2, mu_init=1e-6, want_traj=True, out=wout, sync=True, move_penalty=mp)
            wdt = timed_loop(lambda: A.solve_batch_torch(P_t, NT, tol=args.tol, guess_t=g_t, warm_start=2, mu_init=1e-6, want_traj=True, out=wout,
                                                         move_penalty=mp), args.steps)[0] * 1e-3
            wit = wout["iters"].cpu().numpy()
            wok = int((wout["status"].cpu().numpy() == 0).sum())
            line["continuation"] = {
                "value": wok / wdt, "unit": "NLPs/s", "ms_per_step": wdt * 1e3, "converged": wok,
                "iterations_min_mean_max": [int(wit.min()), float(wit.mean()), int(wit.max())],
                "max_abs_tf_diff_vs_cold": float((wout["tf"] - out["tf"]).abs().max().item()),
                "policy": "primal-dual warm start of every NLP from the nominal Apollo-11 solution, mu0=1e-6; not the headline value",
            }
        if world == 1 and args.dcost > 0.0:
            # Secondary (never `value`): the same steps on the OTHER variant of the model, so that both batch numbers stand side by side
            oout = {}
            A.solve_batch_torch(P_t, NT, tol=args.tol, want_traj=True, out=oout, move_penalty=not mp, sync=True)
            odt, okms = timed_loop(lambda: A.solve_batch_torch(P_t, NT, tol=args.tol, want_traj=True, out=oout, move_penalty=not mp), args.steps)
            odt *= 1e-3
            oit = oout["iters"].cpu().numpy()
            ook = int((oout["status"] == 0).sum().item())
            line["dcost_applied" if not mp else "dcost_not_applied"] = {
                "value": ook / odt, "unit": "NLPs/s", "ms_per_step": odt * 1e3, "kernel_ms": okms, "converged": ook, "of": B,
                "iterations_min_mean_max": [int(oit.min()), float(oit.mean()), int(oit.max())],
                "tf_shift_mean_s": float(((out["tf"] - oout["tf"]).mean() * P[0, 11]).item()) * (1.0 if mp else -1.0),
                "what": ("the same sweep WITH the script's MV DCOST as an l1 move penalty" if not mp else
                         "the same sweep WITHOUT the move penalty (rounds 1 and 2 quoted this variant as the headline)") + "; not the headline value",
            }
        if world == 1 and not args.no_pipelined:
            # Secondary (never `value`): the same K solves enqueued alternately on two streams.  The library keeps a workspace per
            # caller stream, so the wavefronts of one solve fill the SIMDs that the stragglers of the other leave idle (a level ends
            # with its slowest wavefront; at 4096 NLPs every SIMD holds exactly one).  Throughput of streamed batches, not the latency
            # of one solve.
            streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
            pouts = [{}, {}]
            for i in range(4):
                with torch.cuda.stream(streams[i % 2]):
                    A.solve_batch_torch(P_t, NT, tol=args.tol, want_traj=True, out=pouts[i % 2], move_penalty=mp)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for i in range(args.steps):
                with torch.cuda.stream(streams[i % 2]):
                    A.solve_batch_torch(P_t, NT, tol=args.tol, want_traj=True, out=pouts[i % 2], move_penalty=mp)
            torch.cuda.synchronize(dev)
            pdt = (time.perf_counter() - t1) / args.steps
            pok = min(int((po["status"] == 0).sum().item()) for po in pouts)
            line["pipelined_two_streams"] = {
                "value": pok / pdt, "unit": "NLPs/s", "ms_per_step": pdt * 1e3, "converged_per_step": pok,
                "max_abs_tf_diff_vs_single_stream": float(max((po["tf"] - out["tf"]).abs().max().item() for po in pouts)),
                "what": "the same steps enqueued alternately on two HIP streams (one workspace per stream): consecutive solves overlap on the device; not the headline value",
            }
        if world == 1 and not args.no_other_configs and B == 4096:
            # Secondary lines (never `value`): the other BASELINE.json configurations that fit one GPU, each timed over 3 solves
            def timed(fn, n=3):
                fn()
                ms = []
                for _ in range(n):
                    r = fn()
                    ms.append(r.kernel_ms)
                return r, float(np.mean(ms))
            oc = {}
            r1, ms1 = timed(lambda: A.solve_batch(A.AscentParams(), NT, tol=args.tol))
            oc["config1_single_nlp"] = {"ms_per_solve": ms1, "converged": int((r1.status == 0).sum()), "iterations": int(r1.iters[0]),
                                        "final_time_s": float(r1.final_time()[0]), "path": A.default_path(1, NT),
                                        "what": "the reference's own problem (Apollo 11, N=200, backward Euler), cold start"}
            r1t, ms1t = timed(lambda: A.solve_batch(A.AscentParams(), NT, tol=args.tol, scheme=1))
            oc["config1_single_nlp_trapezoid"] = {"ms_per_solve": ms1t, "converged": int((r1t.status == 0).sum()), "iterations": int(r1t.iters[0]),
                                                  "final_time_s": float(r1t.final_time()[0]), "path": A.default_path(1, NT, scheme=1),
                                                  "what": "the same problem with scheme 1 (BASELINE.json configs[1] says 'trapezoidal'; the reference's NODES=2 is backward Euler, SURVEY.md 8a1)"}
            r1d, ms1d = timed(lambda: A.solve_batch(A.AscentParams(dcost=1e-5), NT, tol=args.tol, move_penalty=True, max_iter=500))
            oc["config1_single_nlp_with_dcost"] = {"ms_per_solve": ms1d, "converged": int((r1d.status == 0).sum()), "iterations": int(r1d.iters[0]),
                                                   "final_time_s": float(r1d.final_time()[0]), "path": A.default_path(1, NT, move_penalty=True),
                                                   "what": "the same problem with the script's MV DCOST = 1e-5 applied as an l1 move penalty (Launch_Optimiser.py:99; ascent_opts.move_penalty = 1)"}
            S4 = np.ascontiguousarray(A.sweep_config4()[:32768])
            r4, ms4 = timed(lambda: A.solve_batch(S4, NT, tol=args.tol, want_traj=False))
            oc["config4_shard0"] = {"value": float((r4.status == 0).sum()) / (ms4 * 1e-3), "unit": "NLPs/s", "ms_per_solve": ms4,
                                    "converged": int((r4.status == 0).sum()), "of": 32768, "path": A.default_path(32768, NT),
                                    "iterations_min_mean_max": [int(r4.iters.min()), float(r4.iters.mean()), int(r4.iters.max())],
                                    "roofline": roofline_of(S4, NT, r4, ms4, "config4"),
                                    "what": "first contiguous 32768-NLP shard of the 262144-problem box (one GPU's share at 8 GPUs), DCOST not applied, no trajectories returned"}
            S4d = S4.copy(); S4d[:, 15] = args.dcost if args.dcost > 0 else 1e-5
            r4d, ms4d = timed(lambda: A.solve_batch(S4d, NT, tol=args.tol, want_traj=False, move_penalty=True))
            oc["config4_shard0_with_dcost"] = {"value": float((r4d.status == 0).sum()) / (ms4d * 1e-3), "unit": "NLPs/s", "ms_per_solve": ms4d,
                                               "converged": int((r4d.status == 0).sum()), "of": 32768, "path": A.default_path(32768, NT, move_penalty=True),
                                               "iterations_min_mean_max": [int(r4d.iters.min()), float(r4d.iters.mean()), int(r4d.iters.max())],
                                               "roofline": roofline_of(S4d, NT, r4d, ms4d, None, move_penalty=True),
                                               "what": "the same shard with the script's MV DCOST applied as an l1 move penalty"}
            Sh = P.copy()
            rh, msh = timed(lambda: A.solve_batch(Sh, NT, tol=args.tol, scheme=2, max_iter=500, want_traj=False))
            oc["config3_hermite_simpson"] = {"value": float((rh.status == 0).sum()) / (msh * 1e-3), "unit": "NLPs/s", "ms_per_solve": msh,
                                             "converged": int((rh.status == 0).sum()), "of": len(Sh), "path": A.default_path(len(Sh), NT, scheme=2),
                                             "iterations_min_mean_max": [int(rh.iters.min()), float(rh.iters.mean()), int(rh.iters.max())],
                                             "roofline": roofline_of(Sh, NT, rh, msh, "hs4096", scheme=2, max_iter=500),
                                             "what": "the config-3 sweep with Hermite-Simpson (scheme 2; persistent kernel h_solve of csrc/ascent_hs.hip: structured step Jacobians, one launch per grid level), DCOST not applied"}
            rhd, mshd = timed(lambda: A.solve_batch(Sh, NT, tol=args.tol, scheme=2, max_iter=500, want_traj=False, move_penalty=True), n=2)
            oc["config3_hermite_simpson_with_dcost"] = {"value": float((rhd.status == 0).sum()) / (mshd * 1e-3), "unit": "NLPs/s", "ms_per_solve": mshd,
                                                        "converged": int((rhd.status == 0).sum()), "of": len(Sh),
                                                        "path": A.default_path(len(Sh), NT, scheme=2, move_penalty=True),
                                                        "iterations_min_mean_max": [int(rhd.iters.min()), float(rhd.iters.mean()), int(rhd.iters.max())],
                                                        "what": "the same sweep with Hermite-Simpson AND the move penalty: the one combination the persistent kernels do not carry -- dense-block path (d_eval / d_newton, host-steered rounds), 8x slower than either alone"}
            r5, ms5 = timed(lambda: A.solve_batch(A.AscentParams(), 2000, tol=args.tol, scheme=2, terminal="ellipse", max_iter=500))
            o5 = r5.orbit()
            oc["config5_single_nlp"] = {"ms_per_solve": ms5, "converged": int((r5.status == 0).sum()), "iterations": int(r5.iters[0]),
                                        "final_time_s": float(r5.final_time()[0]), "path": A.default_path(1, 2000, scheme=2, terminal="ellipse"),
                                        "orbit_periapsis_apoapsis_alt_m": [float(o5["periapsis_alt"][0]), float(o5["apoapsis_alt"][0])],
                                        "roofline": roofline_of(A.AscentParams().as_row()[None], 2000, r5, ms5, "config5_one", scheme=2, terminal="ellipse", max_iter=500),
                                        "what": "N=2000 Hermite-Simpson, terminal condition of the (r_peri, r_apo) ellipse (terminal 1), Kepler coast on the device; angular-acceleration bound active"}
            r5f, ms5f = timed(lambda: A.solve_batch(A.AscentParams(), 2000, tol=args.tol, scheme=2, terminal="ellipse_free", max_iter=500), n=1)
            o5f = r5f.orbit()
            oc["config5_single_nlp_burnout_anywhere"] = {"ms_per_solve": ms5f, "converged": int((r5f.status == 0).sum()), "iterations": int(r5f.iters[0]),
                                                         "final_time_s": float(r5f.final_time()[0]), "burn_saved_vs_terminal1_ms": float((r5.final_time()[0] - r5f.final_time()[0]) * 1e3),
                                                         "orbit_periapsis_apoapsis_alt_m": [float(o5f["periapsis_alt"][0]), float(o5f["apoapsis_alt"][0])],
                                                         "path": A.default_path(1, 2000, scheme=2, terminal="ellipse_free"),
                                                         "roofline": roofline_of(A.AscentParams().as_row()[None], 2000, r5f, ms5f, None, scheme=2, terminal="ellipse_free", max_iter=500),
                                                         "what": "the same grid with terminal 2: burnout anywhere on the ellipse (angular momentum and energy), the burn--coast problem with the coast arc eliminated exactly; persistent Hermite-Simpson kernel, one NLP per wavefront"}
            S5 = A.sweep_isp_drymass(16, 16)
            r5b, ms5b = timed(lambda: A.solve_batch(S5, 2000, tol=args.tol, scheme=2, terminal="ellipse", max_iter=500, want_traj=False), n=2)
            oc["config5_batch256"] = {"value": float((r5b.status == 0).sum()) / (ms5b * 1e-3), "unit": "NLPs/s", "ms_per_solve": ms5b,
                                      "converged": int((r5b.status == 0).sum()), "of": 256, "path": A.default_path(256, 2000, scheme=2),
                                      "iterations_min_mean_max": [int(r5b.iters.min()), float(r5b.iters.mean()), int(r5b.iters.max())],
                                      "roofline": roofline_of(S5, 2000, r5b, ms5b, "config5", scheme=2, terminal="ellipse", max_iter=500),
                                      "what": "256 NLPs (16 x 16 Isp x dry-mass sweep) at N=2000 Hermite-Simpson, terminal 1: config 5 as a batch (persistent Hermite-Simpson kernel, one NLP per wavefront)"}
            r5c, ms5c = timed(lambda: A.solve_batch(S5, 2000, tol=args.tol, scheme=2, terminal="ellipse_free", max_iter=500, want_traj=False), n=2)
            oc["config5_batch256_burnout_anywhere"] = {"value": float((r5c.status == 0).sum()) / (ms5c * 1e-3), "unit": "NLPs/s", "ms_per_solve": ms5c,
                                                       "converged": int((r5c.status == 0).sum()), "of": 256, "path": A.default_path(256, 2000, scheme=2, terminal="ellipse_free"),
                                                       "iterations_min_mean_max": [int(r5c.iters.min()), float(r5c.iters.mean()), int(r5c.iters.max())],
                                                       "roofline": roofline_of(S5, 2000, r5c, ms5c, "config5_free", scheme=2, terminal="ellipse_free", max_iter=500),
                                                       "what": "the same batch with terminal 2"}
            if not args.no_cpu_baseline:
                # BASELINE.json configs[1] beside ONE host core: the C restatement of the same algorithm on the same single problem
                # (median of 5 solves, one thread; a stated baseline like cpu_baseline below, never the target)
                from oracle import c_oracle
                c_oracle.build()
                def one_core(**kw):
                    row = A.AscentParams(dcost=1e-5).as_row()[None]
                    c_oracle.solve_batch(row, NT, 500, args.tol, **kw)
                    ts = []
                    for _ in range(5):
                        t0 = time.perf_counter(); c_oracle.solve_batch(row, NT, 500, args.tol, **kw); ts.append(time.perf_counter() - t0)
                    return float(np.median(ts)) * 1e3
                for key, kw in (("config1_single_nlp", {}), ("config1_single_nlp_trapezoid", {"scheme": 1}), ("config1_single_nlp_with_dcost", {"move_penalty": True})):
                    oc[key]["cpu_one_core_ms"] = one_core(**kw)
                    oc[key]["cpu_one_core_what"] = "the plain-C restatement (oracle/) of the same algorithm, same problem, tolerance and cold start, one host thread"
                c_oracle.set_scheme(0)
            line["other_configs"] = oc
        if world == 1 and not args.no_cpu_baseline:
            cb, idx, res = cpu_baseline(P, NT, args.tol, min(args.cpu_sample, B), move_penalty=mp)
            tf_cpu = np.concatenate([r["tf"] for r in res])
            # the baseline doubles as a live parity check of the timed run
            order = np.concatenate([c for c in np.array_split(np.arange(len(idx)), cb["cores"] * 4) if len(c)])
            tf_gpu = out["tf"].cpu().numpy()[idx[order]]
            cb["max_rel_tf_diff_vs_gpu"] = float(np.max(np.abs(tf_cpu - tf_gpu) / tf_cpu))
            line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
