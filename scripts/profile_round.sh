#!/bin/bash
# Everything profiles/ holds for one round, in one GPU-box call:  bash scripts/profile_round.sh r01
#   1. FETCH_SIZE / WRITE_SIZE passes over scripts/wino_shapes.py  -> profiles/<tag>_wino_traffic.json
#   2. PMC passes over one dominant-kernel shape (MFMA busy, waits, LDS)      -> gpurun_out/<tag>_pmc.log
#   3. rocprofv3 --kernel-trace --stats of the default bench (no cpu baseline) -> <tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json
#   4. the plain default bench with cpu_baseline                             -> <tag>_bench.json
set -uo pipefail
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "[1] traffic passes"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/scripts/wino_shapes.py 2240 > $O/shapes_fetch.log 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/scripts/wino_shapes.py 2240 > $O/shapes_write.log 2>&1 || echo "write pass failed"
python3 $R/scripts/parse_wino_traffic.py $O/shapes_fetch.log $O/fetch $O/write $O/${TAG}_wino_traffic.json > $O/traffic_summary.txt 2>&1 || echo "traffic parse failed"
cat $O/traffic_summary.txt
# bench.py only quotes a traffic table that carries the hash of the winograd.hip in the tree: put the fresh one where it looks
[ -s $O/${TAG}_wino_traffic.json ] && cp $O/${TAG}_wino_traffic.json $R/profiles/${TAG}_wino_traffic.json
echo "[1b] traffic passes, F(4x4,3x3) kernel"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch43 -- python3 $R/scripts/wino_shapes.py 2240 43 > $O/shapes43_fetch.log 2>&1 || echo "fetch43 pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write43 -- python3 $R/scripts/wino_shapes.py 2240 43 > $O/shapes43_write.log 2>&1 || echo "write43 pass failed"
python3 $R/scripts/parse_wino_traffic.py $O/shapes43_fetch.log $O/fetch43 $O/write43 $O/${TAG}_wino43_traffic.json 43 > $O/traffic43_summary.txt 2>&1 || echo "traffic43 parse failed"
cat $O/traffic43_summary.txt
[ -s $O/${TAG}_wino43_traffic.json ] && cp $O/${TAG}_wino43_traffic.json $R/profiles/${TAG}_wino43_traffic.json
echo "[1c] traffic passes, F(4x4,3x3) kernel on fp16 pairs"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch43h -- python3 $R/scripts/wino_shapes.py 2240 43h > $O/shapes43h_fetch.log 2>&1 || echo "fetch43h pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write43h -- python3 $R/scripts/wino_shapes.py 2240 43h > $O/shapes43h_write.log 2>&1 || echo "write43h pass failed"
python3 $R/scripts/parse_wino_traffic.py $O/shapes43h_fetch.log $O/fetch43h $O/write43h $O/${TAG}_wino43h_traffic.json 43h > $O/traffic43h_summary.txt 2>&1 || echo "traffic43h parse failed"
cat $O/traffic43h_summary.txt
[ -s $O/${TAG}_wino43h_traffic.json ] && cp $O/${TAG}_wino43h_traffic.json $R/profiles/${TAG}_wino43h_traffic.json
echo "[1d] traffic passes, row-wise F(4,3) kernel on fp16 pairs"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch1d -- python3 $R/scripts/wino_shapes.py 2240 1d > $O/shapes1d_fetch.log 2>&1 || echo "fetch1d pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write1d -- python3 $R/scripts/wino_shapes.py 2240 1d > $O/shapes1d_write.log 2>&1 || echo "write1d pass failed"
python3 $R/scripts/parse_wino_traffic.py $O/shapes1d_fetch.log $O/fetch1d $O/write1d $O/${TAG}_wino1d_traffic.json 1d > $O/traffic1d_summary.txt 2>&1 || echo "traffic1d parse failed"
cat $O/traffic1d_summary.txt
[ -s $O/${TAG}_wino1d_traffic.json ] && cp $O/${TAG}_wino1d_traffic.json $R/profiles/${TAG}_wino1d_traffic.json
echo "[2] PMC of the dominant kernel (16x16 256->256)"
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$n -- python3 $R/scripts/wino_one.py 16 256 256 2240 w1d > $O/pmc_$n.log 2>&1 || echo "pmc pass $n failed"
done
python3 - <<PY > $O/${TAG}_pmc_winograd.txt
import csv, glob, collections
print("wino1d_kernel (F(4,3) along the rows on fp16 pairs), conv [2240,16,16,256]->256, per launch (rocprofv3 --pmc, one pass per counter set):")
for f in sorted(glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "wino1d_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"  {k:32s} {sum(v)/max(1,len(v)):.4g}")
for f in sorted(glob.glob("$O/pmc_SQ_VALU*/**/*kernel_trace.csv", recursive=True)):
    d = [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'wino1d_kernel' in r['Kernel_Name']]
    print("  kernel duration us (that pass):", [round(x) for x in d])
PY
cat $O/${TAG}_pmc_winograd.txt
echo "[3] kernel stats of the bench"
cd $R
( cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/${TAG}_bench_under_rocprof.json 2> $O/bench_under_rocprof.err ) || echo "stats run failed"
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/${TAG}_kernel_stats.csv && python3 scripts/stats_top.py $O/stats 30 > $O/${TAG}_kernel_stats_top.txt
cat $O/${TAG}_kernel_stats_top.txt | head -12
echo "[4] plain bench"
timeout -k 10 600 python3 bench.py --steps 3 --warmup 1 > $O/${TAG}_bench.json 2> $O/bench.err || echo "bench failed"
cut -c1-300 $O/${TAG}_bench.json
echo "[5] eigensolver kernels (two-stage) at D = 3072 and 12288"
bash $R/scripts/sbr_prof.sh ${TAG}_sbr > $O/${TAG}_eigensolver_kernels.txt 2>&1 || echo "sbr profile failed"
for D in 3072 12288; do cp $R/gpurun_out/${TAG}_sbr_$D/sbr_kernel_stats.csv $O/${TAG}_eigensolver_kernel_stats_$D.csv 2>/dev/null; done
head -24 $O/${TAG}_eigensolver_kernels.txt
echo "[6] memory-bound op kernels at SURVEY 8(d) cfg-3 shapes"
timeout -k 10 300 python3 $R/scripts/ops_probe.py > $O/${TAG}_membound_kernels.log 2>&1 || echo "ops probe failed"
tail -25 $O/${TAG}_membound_kernels.log
