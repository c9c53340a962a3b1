#!/bin/bash
# rocprofv3 kernel stats of one spectrum's eigensolve at D = 3072 and D = 12288 -> gpurun_out/<tag>_{3072,12288}/ + summary on stdout
TAG=${1:-sbr}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for D in 3072 12288; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_$D -o sbr -- python3 $R/scripts/sbr_prof.py $D > $R/gpurun_out/${TAG}_$D.log 2>&1 || echo "profile $D failed"
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/${TAG}_$D/sbr_kernel_stats.csv")))
tot = 0
print("D = $D (per spectrum)")
for r in rows[:18]:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0][:34]
    tot += int(r['TotalDurationNs']) / 2e6
    print(f"  {n:36s} calls {int(r['Calls'])//2:5d} total {int(r['TotalDurationNs'])/2e6:8.2f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us")
print("  sum", round(tot, 2))
PY
done
