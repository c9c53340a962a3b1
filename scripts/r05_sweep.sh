#!/bin/bash
# launch-set size and concurrent-set sweep of the headline loop on ONE box (bench.py, 3 points each, no probe / extras / CPU baseline)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_sweep.txt; : > $O
for args in "--inflight 2240" "--inflight 1120" "--inflight 1536" "--inflight 4480" "--inflight 2240 --concurrent-sets 2" "--inflight 2240"; do
  python bench.py --steps 3 --warmup 1 --no-probe --no-extras --no-cpu-baseline $args 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('%-44s %8.0f evals/s  %7.1f ms per point' % ('$args', d['value'], d['ms_per_step']))" | tee -a $O
done
