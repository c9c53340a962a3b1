#!/bin/bash
# PMC passes over the split-precision igemm_pipe_kernel on one shape: bash scripts/igemm_pmc.sh [M N K]   -> gpurun_out/igemm_pmc.txt
set -uo pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/igemm_pmc; mkdir -p $O
M=${1:-573440}; N=${2:-256}; K=${3:-256}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$n -- python3 $R/scripts/gemm_one.py $M $N $K > $O/pmc_$n.log 2>&1 || echo "pmc pass $n failed"
done
python3 - <<PY > $R/gpurun_out/igemm_pmc.txt
import csv, glob, collections
print("igemm_pipe_kernel<128,128,2,2,false,false,true>, GEMM [$M x $K] x [$N x $K]^T, per launch (rocprofv3 --pmc):")
for f in sorted(glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "igemm_pipe_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"  {k:32s} {sum(v)/max(1,len(v)):.4g}")
for f in sorted(glob.glob("$O/pmc_SQ_VALU*/**/*kernel_trace.csv", recursive=True)):
    d = [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'igemm_pipe_kernel' in r['Kernel_Name']]
    print("  kernel duration us (that pass):", [round(x) for x in d])
PY
cat $R/gpurun_out/igemm_pmc.txt
