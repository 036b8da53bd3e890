#!/bin/bash
set -u
TAG=${1:-r04q}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_teacher_forced.py tests/test_gpu_parity.py -m gpu -x -q -k "window or s24 or cohort or split or 32_row" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log | cut -c1-300
run() {
  local label=$1 cfg=$2; shift 2
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --config $cfg --no-extra-configs --no-cpu-baseline --no-second-window --steps 100 "$@" > $OUT/b_$label.json 2> $OUT/b_$label.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_$label.json").read().strip().splitlines()[-1])
    print("$label:", round(r["value"] / 1e6, 3), "M  ms/step", round(r["ms_per_step"], 4), "sweeps", round(r["config"]["mean_solver_iter"], 2), "cohorts", r["config"]["cohorts"], "overflow", r["config"]["overflow_envs"], "kernel_ms", round(r["roofline"]["kernel_ms"], 4))
except Exception as ex:
    print("$label: FAILED", ex); print(open("$OUT/b_$label.err").read()[-500:])
PY
}
run s24 s24 A=1 --
run s24_again s24 A=1 --
run s24_nl3 s24 MJH_WN_NL=3 --
run s24_fullldsA s24 MJH_WINDOW_SLIM_LDS=0 --
run s24_c2 s24 A=1 -- --cohorts 2
run s24_c4 s24 A=1 -- --cohorts 4
run s24d s24d A=1 --
