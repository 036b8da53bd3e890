#!/bin/bash
# S24 / S24D: cohort count x LDS tier of the window kernel (MJH_WN_NL), fused kernel for comparison (MJH_WINDOW=0); stage profile
set -u
TAG=${1:-r04b}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -s -k "independent" > $OUT/pytest_boxbox.log 2>&1; echo "pytest rc=$?"
grep -E "BOXBOX|passed|failed|Error|assert" $OUT/pytest_boxbox.log | cut -c1-1500 | tail -12
run() {  # label, config, env assignments..., then bench args after --
  local label=$1 cfg=$2; shift 2
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --config $cfg --no-extra-configs --no-cpu-baseline --no-second-window --steps 100 "$@" > $OUT/b_$label.json 2> $OUT/b_$label.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_$label.json").read().strip().splitlines()[-1])
    print("$label:", round(r["value"] / 1e6, 3), "M  ms/step", round(r["ms_per_step"], 4), "sweeps", round(r["config"]["mean_solver_iter"], 1), "nefc", round(r["config"]["mean_nefc"], 1), "max", r["config"]["max_nefc"], "overflow", r["config"]["overflow_envs"], "kernel_ms", round(r["roofline"]["kernel_ms"], 4))
except Exception as ex:
    print("$label: FAILED", ex); print(open("$OUT/b_$label.err").read()[-800:])
PY
}
for c in 2 3 4 6; do
  run s24_c${c}_nl0 s24 MJH_WN_NL=0 -- --cohorts $c
  run s24_c${c}_nl3 s24 MJH_WN_NL=3 -- --cohorts $c
done
for c in 2 3 4; do
  run s24d_c${c}_nl3 s24d MJH_WN_NL=3 -- --cohorts $c
  run s24d_c${c}_nl1 s24d MJH_WN_NL=1 -- --cohorts $c
done
run s24d_fused_c2 s24d MJH_WINDOW=0 -- --cohorts 2
run s24d_fused_c3 s24d MJH_WINDOW=0 -- --cohorts 3
run s24d_cap80 s24d MJH_WN_NL=3 -- --cohorts 3 --maxcon 80
python tools/stage_profile.py 4096 400 > $OUT/stage_s24.txt 2>&1; cat $OUT/stage_s24.txt
