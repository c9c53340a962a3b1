#!/bin/bash
# Second part of r05_l1_exp.sh: does the time follow the L1's request count?  Timing-only builds without the patch requests that repeat a neighbouring tile's.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/l1exp2
mkdir -p $OUT
cd $R
python scripts/wino43h_ab.py "" "-DIDIFF_W43H_DIAG_NO_COL45" "-DIDIFF_W43H_DIAG_NO_COL45 -DIDIFF_W43H_DIAG_NO_ROW45" "-DIDIFF_W43H_COLORDER=1 -DIDIFF_W43H_COLS_PER_PART=2" \
   "-DIDIFF_W43H_COLORDER=1" "" 2>&1 | grep -v "amdgpu.ids\|DIAGNOSTIC" | tee $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
n=0
for flags in "-DIDIFF_W43H_COLORDER=1" "-DIDIFF_W43H_DIAG_NO_COL45" "-DIDIFF_W43H_DIAG_NO_COL45 -DIDIFF_W43H_DIAG_NO_ROW45" "-DIDIFF_W43H_COLORDER=1 -DIDIFF_W43H_COLS_PER_PART=2"; do
  n=$((n+1))
  IDIFF_VARIANT=pmc$n IDIFF_VARIANT_FLAGS="$flags" IDIFF_SCRATCH_LIMIT=100000 bash $R/id-diff_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build $n failed"; exit 1; }
  export IDIFF_LIB_VARIANT=pmc$n
  for shape in "16 256 256" "32 128 128"; do
    tag=$(echo $shape | tr ' ' '_')
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/v${n}_${tag} -- python3 $R/scripts/wino_one.py $shape 2240 f43h > $OUT/v${n}_${tag}.log 2>&1 || echo "variant $n $shape failed"
  done
  unset IDIFF_LIB_VARIANT
  rm -f $R/id-diff_amd/csrc/libidiff_hip.pmc$n.so
  echo "variant $n ('$flags') done"
done
python3 - <<PY | tee $OUT/pmc.txt
import csv, glob, collections, os
for d in sorted(glob.glob("$OUT/v*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "winograd43h_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{os.path.basename(d.rstrip('/')):24s} {k:32s} per launch {sum(v)/max(1,len(v)):.4g}  (n={len(v)})")
PY
