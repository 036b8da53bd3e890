#!/bin/bash
# full GPU test suite + a few bench variants.  usage: tools/r03_suite.sh <tag>
set -u
TAG=${1:-r03b}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -q -x -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep "TEACHER-FORCED\|S24 free run\|passed\|failed\|Error\|MUJOCO-PRESENCE" $OUT/pytest_gpu.log | tail -20
for v in "" "--no-gather"; do
  timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-second-window $v > $OUT/bench_s24$v.json 2> $OUT/bench_s24$v.err; python - <<PY
import json
try:
    r=json.loads(open("$OUT/bench_s24$v.json").read().strip().splitlines()[-1]); print("s24 $v", r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"])
except Exception as e: print("fail", e)
PY
done
for c in c3 c5; do for v in "" "--no-gather"; do
  timeout 600 python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline --no-second-window $v > $OUT/bench_$c$v.json 2> $OUT/bench_$c$v.err; python - <<PY
import json
try:
    r=json.loads(open("$OUT/bench_$c$v.json").read().strip().splitlines()[-1]); print("$c $v", r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"])
except Exception as e: print("fail", e)
PY
done; done
