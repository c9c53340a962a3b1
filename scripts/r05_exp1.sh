#!/bin/bash
# round 5, GPU call 1b: the whole GPU suite (no -x), then the workgroup-order experiment for the dominant kernel
set -uo pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python -m pytest tests -m gpu -q > $O/r05_gputests_b.log 2>&1; echo "pytest rc=$?" >> $O/r05_gputests_b.log; tail -4 $O/r05_gputests_b.log
echo "== default order"; python scripts/wino43h_probe.py time > $O/r05_ngroup_default.txt 2>&1; tail -13 $O/r05_ngroup_default.txt
echo "== IDIFF_WINO_NGROUP=4"; IDIFF_WINO_NGROUP=4 python scripts/wino43h_probe.py time > $O/r05_ngroup_4.txt 2>&1; tail -13 $O/r05_ngroup_4.txt
echo "== IDIFF_WINO_NGROUP=1"; IDIFF_WINO_NGROUP=1 python scripts/wino43h_probe.py time > $O/r05_ngroup_1.txt 2>&1; tail -13 $O/r05_ngroup_1.txt
cd /tmp && export TMPDIR=/tmp
export IDIFF_WINO_NGROUP=4
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_r05_ng4/fetch43h -- python3 $R/scripts/wino_shapes.py 2240 43h > $O/prof_r05_ng4_shapes_fetch.log 2>&1 || echo "fetch pass failed"
unset IDIFF_WINO_NGROUP
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/prof_r05_ng4/fetch43h/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "winograd43h_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 1024 * 2 / 1e6))
rows.sort()
keys = [l.split()[1] for l in open("$O/prof_r05_ng4_shapes_fetch.log") if l.startswith("KEY")]
print("NGROUP=4 fetch MB per launch (second launch of each shape):")
# the forward itself comes first; the isolated launches are the last 2 * len(keys) dispatches
iso = rows[-2 * len(keys):]
for i, k in enumerate(keys):
    print(f"  {k:24s} {iso[2 * i + 1][1]:9.1f} MB")
PY
