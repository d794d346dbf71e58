import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms", tot / 1e6)
for r in rows[:14]:
    print(f"{float(r['TotalDurationNs'])/1e6:8.1f} ms {r['Calls']:>6} calls  {r['Name'][:100]}")
