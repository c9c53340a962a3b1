#!/bin/bash
# SQ-level PMC passes over scripts/wino_one.py (run on the GPU box): pmc_sq.sh TAG H Cin Cout B [split]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/scripts/wino_one.py "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "winograd" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:36s} per launch {sum(v)/max(1,len(v)):.4g}  (n={len(v)})")
PY
