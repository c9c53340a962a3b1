#!/bin/bash
# PMC passes over winograd43h_kernel on one shape: bash scripts/w43h_pmc.sh [H Cin Cout]   -> gpurun_out/w43h_pmc.txt
set -uo pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/w43h_pmc; mkdir -p $O
H=${1:-16}; CI=${2:-256}; CO=${3:-256}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$n -- python3 $R/scripts/wino_one.py $H $CI $CO 2240 f43h > $O/pmc_$n.log 2>&1 || echo "pmc pass $n failed"
done
python3 - <<PY > $R/gpurun_out/w43h_pmc.txt
import csv, glob, collections
print("winograd43h_kernel, conv [2240,$H,$H,$CI]->$CO, per launch (rocprofv3 --pmc):")
for f in sorted(glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "winograd43h_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"  {k:32s} {sum(v)/max(1,len(v)):.4g}")
for f in sorted(glob.glob("$O/pmc_SQ_VALU*/**/*kernel_trace.csv", recursive=True)):
    d = [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'winograd43h_kernel' in r['Kernel_Name']]
    print("  kernel duration us (that pass):", [round(x) for x in d])
PY
cat $R/gpurun_out/w43h_pmc.txt
