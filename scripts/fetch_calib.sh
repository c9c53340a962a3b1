#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_fetch_calib; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -- $R/scripts/fetch_calib > $O/run.log 2>&1 || echo "calib failed"
python3 - <<PY | tee $R/gpurun_out/r05_fetch_calibration.txt
import csv, glob
rows = []
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], float(r["Counter_Value"])))
rows.sort()
asked = [256.0, 256.0, 128.0, 32.0]
what = ["16 B per lane, streaming (the guide's calibrated case)", "4 B per lane, 64-byte chunks, every chunk (whole 128-byte lines)",
        "4 B per lane, even chunks only (half of every 128-byte line asked for)", "4 B per lane, one 64-byte chunk of every 512 bytes"]
print("FETCH_SIZE (KB as reported) against the bytes the kernel asked for; factor = asked / reported:")
for (d, name, v), a, w in zip(rows[-4:], asked, what):
    print(f"  {name:10s} asked {a:6.1f} MiB  reported {v / 1024:7.1f} MiB  factor {a * 1024 / v:5.2f}   {w}")
PY
