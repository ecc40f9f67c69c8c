"""Developer tool: summary of the SQ counters scripts/sq_counters.sh collected: per p_solve / h_solve dispatch of the LARGEST grid level
(the one with the most SQ_WAVE_CYCLES), averaged over the solves -- instruction mix, issue / wait / stall shares per wavefront.
(SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles: x4 for shader cycles, MI355X_MICROARCH.md)"""
import csv, glob, os, sys
tot = {}
for d in sys.argv[1:3]:
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if "p_solve" not in r["Kernel_Name"] and "h_solve" not in r["Kernel_Name"]:      # (h_solve: the Hermite-Simpson kernel)
            continue
        per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    # the fine-level dispatches: those with the largest value of the pass's first cycle-like counter
    key = "SQ_WAVE_CYCLES" if any("SQ_WAVE_CYCLES" in v for v in per.values()) else "SQ_WAIT_ANY"
    big = max(v[key] for v in per.values())
    fine = [v for v in per.values() if v[key] > 0.6 * big]
    for k in fine[0]:
        tot[k] = sum(v[k] for v in fine) / len(fine)
    print(f"{d}: {len(per)} p_solve / h_solve dispatches, {len(fine)} of the finest grid level")
for k in sorted(tot):
    print(f"  {k:22s} {tot[k]:.4g}")
w = tot["SQ_WAVES"]
cyc = 4 * tot["SQ_WAVE_CYCLES"] / w
ins = tot["SQ_INSTS_VALU"] + tot["SQ_INSTS_SALU"] + tot["SQ_INSTS_LDS"] + tot["SQ_INSTS_VMEM_RD"] + tot["SQ_INSTS_VMEM_WR"] + tot.get("SQ_INSTS_SMEM", 0)
print(f"per wavefront: {cyc / 1e6:.2f} M cycles; issuing {100 * tot['SQ_ACTIVE_INST_ANY'] / tot['SQ_WAVE_CYCLES']:.0f} %, parked on s_waitcnt "
      f"{100 * tot['SQ_WAIT_ANY'] / tot['SQ_WAVE_CYCLES']:.0f} %, issue stalls {100 * tot['SQ_WAIT_INST_ANY'] / tot['SQ_WAVE_CYCLES']:.0f} %")
print(f"instructions per wavefront: {ins / w / 1e3:.0f} k (VALU {100 * tot['SQ_INSTS_VALU'] / ins:.0f} %, LDS {100 * tot['SQ_INSTS_LDS'] / ins:.0f} %, scalar "
      f"{100 * tot['SQ_INSTS_SALU'] / ins:.0f} %, global loads {100 * tot['SQ_INSTS_VMEM_RD'] / ins:.1f} %, global stores {100 * tot['SQ_INSTS_VMEM_WR'] / ins:.1f} %)")
print(f"VALU issue: {4 * tot['SQ_ACTIVE_INST_VALU'] / tot['SQ_INSTS_VALU']:.2f} cycles per instruction")
