#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r03_sbr_pmc2; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/p1 -- python3 $R/scripts/sbr_prof.py 8192 > $O/p1.log 2>&1 || echo "pass failed"
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$O/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for name in ("trailing_y_kernel", "trailing_update_strip_kernel"):
            if name in r["Kernel_Name"]:
                acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
dur = collections.defaultdict(float)
for f in glob.glob("$O/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for name in acc:
            if name in r["Kernel_Name"]:
                dur[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for name in acc:
    print(name, f"{dur[name]/1e3:.1f} ms total (profiled)")
    for c, v in sorted(acc[name].items()): print(f"   {c:30s} {v:.4g}")
PY
