#!/bin/bash
# window sweep: S24 teacher-forced parity, then the bench line with and without it
set -u
TAG=${1:-r04d}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -x -q -s -k "s24" > $OUT/pytest_tf.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_tf.log
grep -E "TEACHER|bitwise|passed|failed|Error|assert" $OUT/pytest_tf.log | cut -c1-600 | tail -20
for w in 1 0; do
  MJH_WINDOW=$w timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --no-second-window --steps 100 > $OUT/bench_s24_win$w.json 2> $OUT/bench_s24_win$w.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/bench_s24_win$w.json").read().strip().splitlines()[-1])
    print("S24 window $w:", round(r["value"] / 1e6, 3), "M env-steps/s  ms/step", round(r["ms_per_step"], 4), "sweeps", round(r["config"]["mean_solver_iter"], 1), "ncon", round(r["config"]["mean_ncon"], 1), "nefc", round(r["config"]["mean_nefc"], 1), "overflow", r["config"]["overflow_envs"], "reset", r["config"]["reset_envs"])
except Exception as ex:
    print("S24 window $w: FAILED", ex); print(open("$OUT/bench_s24_win$w.err").read()[-1500:])
PY
done
