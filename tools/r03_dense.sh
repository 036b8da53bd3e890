#!/bin/bash
# dense solver (dense_pgs.h): parity of the robot fixtures in the global-pools layout, C4 bench with and without it
set -u
TAG=${1:-r03c}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x -k "robot_models or c4 or pr2 or many_body or sensors or sub_wave" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
for d in 1 0; do
  MJH_DENSE=$d timeout 600 python bench.py --config c4 --steps 100 --warmup 20 --no-cpu-baseline --no-second-window > $OUT/bench_c4_dense$d.json 2> $OUT/bench_c4_dense$d.err
  python - <<PY
import json
try:
    r=json.loads(open("$OUT/bench_c4_dense$d.json").read().strip().splitlines()[-1]); print("c4 dense=$d", r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["config"]["mean_nefc"], r["config"]["max_nefc"], r["config"]["mean_solver_iter"])
except Exception as e: print("fail", e); print(open("$OUT/bench_c4_dense$d.err").read()[-1500:])
PY
done
