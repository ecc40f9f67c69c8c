"""Developer tool: profiles/traffic.json from rocprofv3 output.

    python scripts/traffic_from_pmc.py <dir of --pmc FETCH_SIZE run> <dir of --pmc WRITE_SIZE run> <dir of --kernel-trace --stats run> <solves in each run> [key]

FETCH_SIZE / WRITE_SIZE are in KB.  gfx950 counts a 128-byte read request as 64 bytes, so FETCH_SIZE of wide coalesced
streaming reads is doubled (MI355X_MICROARCH.md, HBM section) -- the guide calls other access widths uncalibrated and asks for a
calibration on a known byte count in the kernel's own pattern.  The kernels of the persistent path (p_*) read rows of 16 nodes x
8 B per NLP (one 128-byte line per 16 lanes): p_transfer to the 200-node grid reads exactly 21 rows x 64 nodes x 8 B + 70 scalars
+ 128 B of parameters per NLP = 46.8 MB at batch 4096 and FETCH_SIZE reports 45.4 MB UNdoubled (p_init: 0.52 MB of parameters,
0.6 MB reported), so their factor is 1; WRITE_SIZE is exact on both (145.4 MB / 65.6 MB against 145.4 / 65.9 known).  Kernels
reading [row][batch] blobs 512 B per wavefront-instruction (q_*, k_*) keep the factor 2 that q_init's known traffic confirmed."""
import csv, glob, json, os, re, sys

fetch_dir, write_dir, stats_dir, n_solves = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
key = sys.argv[5] if len(sys.argv) > 5 else "config3"      # a PROF_WORKLOAD of scripts/prof_solve.py
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_sha16  # noqa: E402  (the stamp bench.py checks before it quotes this file)


def short(name):
    m = re.search(r"\b(q_\w+|k_\w+|p_\w+|h_solve|d_\w+|bt_\w+|pc_\w+)", name)
    return m.group(1) if m else name[:24]


def counters(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    tot, per, n = 0.0, {}, {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        v = float(r["Counter_Value"]) * 1024.0
        k = short(r["Kernel_Name"])
        tot += v
        per[k] = max(per.get(k, 0.0), v)     # the dispatch with all lanes active
        n[k] = n.get(k, 0) + 1
    return tot, per, n


def fetch_factor(kernel):
    return 1.0 if kernel.startswith(("p_", "h_")) else 2.0      # (h_solve: the layout and the 8-byte node-contiguous accesses of p_solve)


def counters_per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    tot = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = short(r["Kernel_Name"])
            tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"]) * 1024.0
    return tot


_, rd_max, _ = counters(fetch_dir, "FETCH_SIZE")
wr, wr_max, _ = counters(write_dir, "WRITE_SIZE")
rd_tot = counters_per_kernel(fetch_dir, "FETCH_SIZE")
rd = sum(fetch_factor(k) * v for k, v in rd_tot.items())
rd_max = {k: fetch_factor(k) * v for k, v in rd_max.items()}
stats = {}
f = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)[0]
for r in csv.DictReader(open(f)):
    k = short(r["Name"])
    if not k.startswith(("q_", "k_", "p_", "h_", "d_", "bt_", "pc_")):
        continue
    stats[k] = {"calls_per_solve": int(r["Calls"]) / n_solves, "avg_us": float(r["AverageNs"]) / 1e3,
                "ms_per_solve": float(r["TotalDurationNs"]) / 1e6 / n_solves, "percent": float(r["Percentage"])}
out = {
    "source_sha16": source_sha16(),
    "workload": f"scripts/prof_solve.py PROF_WORKLOAD={key}, default dispatch, cold start, tol 1e-9",
    "hbm_bytes_per_launch": (rd + wr) / n_solves, "read_bytes": rd / n_solves, "write_bytes": wr / n_solves,
    "definition": "one launch = one bench step = one whole solve of the batch (all kernels of all rounds)",
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KB*1024; FETCH_SIZE x fetch_factor: 1 for the p_* "
              "kernels (16 nodes x 8 B = one 128-B line per 16 lanes; calibrated on p_transfer: 46.8 MB known, 45.4 MB reported undoubled; "
              "p_init: 0.52 MB known, 0.6 reported), 2 for kernels streaming 512 B per wavefront-instruction (gfx950 counts 128-B requests as "
              "64 B, MI355X_MICROARCH.md HBM section; confirmed on q_init in round 1); WRITE_SIZE exact (p_transfer 145.4 MB, p_finish 65.6 MB "
              "against 145.4 / 65.9 known)",
    "fetch_factor": {k: fetch_factor(k) for k in sorted(rd_tot)},
    "per_dispatch_all_lanes_active_MB": {k: {"read": rd_max.get(k, 0) / 1e6, "write": wr_max.get(k, 0) / 1e6} for k in sorted(set(rd_max) | set(wr_max))},
    "kernel_stats": stats, "sum_kernel_ms_per_solve": sum(v["ms_per_solve"] for v in stats.values()),
}
path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(path)) if os.path.exists(path) else {}
tj[key] = out
json.dump(tj, open(path, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "read_bytes", "write_bytes", "sum_kernel_ms_per_solve")}, indent=1))
for k, v in out["per_dispatch_all_lanes_active_MB"].items():
    print(f"{k:18s} read {v['read']:9.1f} MB  write {v['write']:9.1f} MB   {stats.get(k, {}).get('avg_us', 0):8.1f} us avg")
