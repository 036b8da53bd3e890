#!/bin/bash
# in-kernel step loop: bitwise test, C3 / C5 with 1 / 3 / 8 steps per launch (with and without the 60 Hz publish), parity prints
set -u
TAG=${1:-r04c}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -s -k "loop or independent" > $OUT/pytest_loop.log 2>&1; echo "pytest loop rc=$?"
grep -E "BOXBOX|passed|failed|Error|assert|differ" $OUT/pytest_loop.log | cut -c1-600 | tail -12
run() {
  local label=$1 cfg=$2; shift 2
  timeout 300 python bench.py --config $cfg --no-extra-configs --no-cpu-baseline --no-second-window --steps 300 "$@" > $OUT/b_$label.json 2> $OUT/b_$label.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_$label.json").read().strip().splitlines()[-1])
    print("$label:", round(r["value"] / 1e6, 3), "M  ms/step", round(r["ms_per_step"], 4), "spl", r["config"]["steps_per_launch"], "cohorts", r["config"]["cohorts"], "kernel_ms", round(r["roofline"]["kernel_ms"], 4))
except Exception as ex:
    print("$label: FAILED", ex); print(open("$OUT/b_$label.err").read()[-800:])
PY
}
for cfg in c3 c5; do
  for spl in 1 3; do run ${cfg}_spl$spl $cfg --steps-per-launch $spl; done
  for spl in 1 8; do run ${cfg}_nogather_spl$spl $cfg --steps-per-launch $spl --no-gather; done
  run ${cfg}_spl3_c2 $cfg --steps-per-launch 3 --cohorts 2
  run ${cfg}_spl3_c4 $cfg --steps-per-launch 3 --cohorts 4
done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q -s -k "trajectory_parity or golden_fixture or sampled_envs" > $OUT/pytest_parity.log 2>&1; echo "pytest parity rc=$?"
grep -E "S24 free run|S24 golden|sampled|passed|failed|Error|assert" $OUT/pytest_parity.log | cut -c1-400 | tail -20
