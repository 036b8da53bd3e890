#!/bin/bash
# Y-form dense solver (Y = J L^-1): dense on / off parity tests, robot fixtures, C4 bench
set -u
TAG=${1:-r03y}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_round3.py -m gpu -q -x > $OUT/pytest_r3.log 2>&1; echo "pytest r3 rc=$?" | tee -a $OUT/pytest_r3.log
tail -15 $OUT/pytest_r3.log
timeout 1500 python -m pytest tests -m gpu -q -x -k "robot_models or c4 or pr2 or many_body or sensors or sub_wave" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -8 $OUT/pytest.log
for i in 1 2; do
  timeout 600 python bench.py --config c4 --steps 100 --warmup 20 --no-cpu-baseline --no-second-window > $OUT/bench_c4_$i.json 2> $OUT/bench_c4_$i.err
  python - <<PY
import json
try:
    r=json.loads(open("$OUT/bench_c4_$i.json").read().strip().splitlines()[-1]); print("c4", r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["config"]["mean_nefc"], r["config"]["max_nefc"], r["config"]["mean_solver_iter"])
except Exception as e: print("fail", e); print(open("$OUT/bench_c4_$i.err").read()[-1500:])
PY
done
