"""Developer tool: how much the kernels of a rocprofv3 kernel trace overlap in time (sum of durations vs union)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
ev.sort()
t0 = ev[0][0]
tot = sum(e - s for s, e, _, _ in ev)
union, cur_s, cur_e = 0, ev[0][0], ev[0][1]
for s, e, _, _ in ev[1:]:
    if s > cur_e:
        union += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f"kernels {len(ev)}  sum of durations {tot/1e6:.2f} ms  union {union/1e6:.2f} ms  span {(ev[-1][1]-t0)/1e6:.2f} ms  queues {sorted(set(q for *_, q in ev))}")
if len(sys.argv) > 2:
    import re
    for s, e, n, q in ev[int(sys.argv[2]):int(sys.argv[2]) + int(sys.argv[3])]:
        m = re.search(r"(q_\w+)", n)
        print(f"q{q} {(s-t0)/1e3:10.1f} {(e-t0)/1e3:10.1f} {(e-s)/1e3:8.1f} us  {m.group(1) if m else n[:20]}")
