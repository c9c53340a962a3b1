import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
for r in rows[:n]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):6.2f}% calls={r['Calls']:>6} avg={float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:100]}")
