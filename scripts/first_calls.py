import csv, sys, glob, re
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
seen={}
for r in rows:
    m=re.search(r"(q_\w+)", r["Kernel_Name"])
    if not m: continue
    k=m.group(1)
    seen.setdefault(k,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in seen.items(): print(k, "first calls (us):", [round(x,1) for x in v[:3]])
