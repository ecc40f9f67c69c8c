import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)): print(r["Name"][:60].ljust(60), r["Calls"], r["AverageNs"], r["Percentage"])
