"""Developer tool: profiles/traffic.json from rocprofv3 output.

    python scripts/traffic_from_pmc.py <dir of --pmc FETCH_SIZE run> <dir of --pmc WRITE_SIZE run> <dir of --kernel-trace --stats run> <solves in each run> [key]

FETCH_SIZE / WRITE_SIZE are in KB; gfx950 counts a 128-byte read request as 64 bytes, so FETCH_SIZE is doubled
(MI355X_MICROARCH.md, HBM section); the correction is checked on q_init, whose traffic is known exactly."""
import csv, glob, json, os, re, sys

fetch_dir, write_dir, stats_dir, n_solves = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
key = sys.argv[5] if len(sys.argv) > 5 else "batch4096"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_sha16  # noqa: E402  (the stamp bench.py checks before it quotes this file)


def short(name):
    m = re.search(r"(q_\w+|k_\w+)", name)
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


rd, rd_max, _ = counters(fetch_dir, "FETCH_SIZE")
wr, wr_max, _ = counters(write_dir, "WRITE_SIZE")
rd, rd_max = 2.0 * rd, {k: 2.0 * v for k, v in rd_max.items()}
stats = {}
f = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)[0]
for r in csv.DictReader(open(f)):
    k = short(r["Name"])
    if not k.startswith(("q_", "k_")):
        continue
    stats[k] = {"calls_per_solve": int(r["Calls"]) / n_solves, "avg_us": float(r["AverageNs"]) / 1e3,
                "ms_per_solve": float(r["TotalDurationNs"]) / 1e6 / n_solves, "percent": float(r["Percentage"])}
out = {
    "source_sha16": source_sha16(),
    "workload": "BASELINE.json configs[2], split pipeline with 16-lane sweeps, cold start, tol 1e-9",
    "hbm_bytes_per_launch": (rd + wr) / n_solves, "read_bytes": rd / n_solves, "write_bytes": wr / n_solves,
    "definition": "one launch = one bench step = one whole solve of the batch (all kernels of all rounds)",
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KB*1024; FETCH_SIZE doubled (gfx950 counts "
              "128-B requests as 64 B, MI355X_MICROARCH.md HBM section); q_init (known: reads 16 doubles per NLP, writes 21+21 rows "
              "per step + scalars) is the calibration",
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
