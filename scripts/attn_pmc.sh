#!/bin/bash
# PMC passes over the one-launch attention kernel (one counter set per pass, kernel trace only)
set -uo pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_attn; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$n -- python3 $R/scripts/attn_one.py > $O/pmc_$n.log 2>&1 || echo "pmc pass $n failed"
done
python3 - <<PY | tee $R/gpurun_out/r05_attn_pmc.txt
import csv, glob, collections
print("attention256_kernel<256>, B = 2240, per launch (rocprofv3 --pmc, one pass per counter set):")
for f in sorted(glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "attention256" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"  {k:32s} {sum(v)/max(1,len(v)):.4g}")
for f in sorted(glob.glob("$O/pmc_SQ_VALU*/**/*kernel_trace.csv", recursive=True)):
    d = [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'attention256' in r['Kernel_Name']]
    print("  kernel duration us (that pass):", [round(x) for x in d])
PY
