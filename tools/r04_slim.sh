#!/bin/bash
# S24: the assemble-only instance beside the window kernel (registers: 128 + NW windows), variants
set -u
TAG=${1:-r04i}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_teacher_forced.py -m gpu -x -q -k "window or s24" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log | cut -c1-300
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --config s24 --no-extra-configs --no-cpu-baseline --no-second-window --steps 100 "$@" > $OUT/b_$label.json 2> $OUT/b_$label.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_$label.json").read().strip().splitlines()[-1])
    print("$label:", round(r["value"] / 1e6, 3), "M  ms/step", round(r["ms_per_step"], 4), "sweeps", round(r["config"]["mean_solver_iter"], 1), "nefc", round(r["config"]["mean_nefc"], 1), "cohorts", r["config"]["cohorts"], "overflow", r["config"]["overflow_envs"], "kernel_ms", round(r["roofline"]["kernel_ms"], 4))
except Exception as ex:
    print("$label: FAILED", ex); print(open("$OUT/b_$label.err").read()[-500:])
PY
}
run base MJH_WINDOW_SLIM=0 --
run slim_nw8_nl3 A=1 --
run slim_nw8_nl0 MJH_WN_NL=0 --
for c in 2 3 4; do
  run slim_nw5_nl1_c$c MJH_WN_NW=5 MJH_WN_NL=1 -- --cohorts $c
  run slim_nw6_nl1_c$c MJH_WN_NW=6 MJH_WN_NL=1 -- --cohorts $c
  run slim_nw5_nl3_c$c MJH_WN_NW=5 MJH_WN_NL=3 -- --cohorts $c
done
run slim_nw8_nl0_c4 MJH_WN_NL=0 -- --cohorts 4
run slim_nw8_nl0_c2 MJH_WN_NL=0 -- --cohorts 2
