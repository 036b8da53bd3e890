#!/bin/bash
# round 4, first GPU session: teacher-forced parity (default = row order, bit-identity of the list schedule), then S24 under the three schedules
set -u
TAG=${1:-r04a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -x -q -s -k "s24 or c2" > $OUT/pytest_tf.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_tf.log
grep -E "TEACHER|bitwise|passed|failed|Error|assert" $OUT/pytest_tf.log | tail -40
for s in 1 0 2; do
  timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --no-second-window --pgs-schedule $s --steps 100 > $OUT/bench_s24_sched$s.json 2> $OUT/bench_s24_sched$s.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/bench_s24_sched$s.json").read().strip().splitlines()[-1])
    print("S24 schedule $s:", round(r["value"] / 1e6, 3), "M env-steps/s  ms/step", round(r["ms_per_step"], 4), "sweeps", round(r["config"]["mean_solver_iter"], 1), "ncon", round(r["config"]["mean_ncon"], 1), r["config"]["pgs_order"], "|", r["config"]["pgs_schedule"])
except Exception as ex:
    print("S24 schedule $s: FAILED", ex); print(open("$OUT/bench_s24_sched$s.err").read()[-1500:])
PY
done
for s in 1 0; do
  timeout 400 python bench.py --config c2 --no-cpu-baseline --no-second-window --pgs-schedule $s --steps 30 --warmup 5 > $OUT/bench_c2_sched$s.json 2> $OUT/bench_c2_sched$s.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/bench_c2_sched$s.json").read().strip().splitlines()[-1])
    print("C2 schedule $s:", round(r["value"] / 1e6, 4), "M env-steps/s  ms/step", round(r["ms_per_step"], 4), "sweeps", round(r["config"]["mean_solver_iter"], 1), "ncon", round(r["config"]["mean_ncon"], 1))
except Exception as ex:
    print("C2 schedule $s: FAILED", ex); print(open("$OUT/bench_c2_sched$s.err").read()[-1500:])
PY
done
