#!/bin/bash
# round 5: full GPU suite, then the bench with and without the one-launch attention on the same box
set -uo pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python -m pytest tests -m gpu -q > $O/r05_gputests_c.log 2>&1; echo "pytest rc=$?" >> $O/r05_gputests_c.log; tail -4 $O/r05_gputests_c.log
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/r05_bench_attn_on.json 2> $O/r05_bench_attn_on.err; python - <<PY
import json; d = json.load(open("$O/r05_bench_attn_on.json")); print("fused attention ON :", round(d["value"]), "evals/s", round(d["ms_per_step"], 1), "ms/point", d["id_estimates"])
PY
IDIFF_NO_FUSED_ATTN=1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/r05_bench_attn_off.json 2> $O/r05_bench_attn_off.err; python - <<PY
import json; d = json.load(open("$O/r05_bench_attn_off.json")); print("fused attention OFF:", round(d["value"]), "evals/s", round(d["ms_per_step"], 1), "ms/point", d["id_estimates"])
PY
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-probe > $O/r05_bench_attn_on2.json 2>/dev/null; python - <<PY
import json; d = json.load(open("$O/r05_bench_attn_on2.json")); print("fused attention ON (again):", round(d["value"]), "evals/s", round(d["ms_per_step"], 1), "ms/point")
PY
