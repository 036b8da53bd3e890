#!/bin/bash
# C3 / C5: environments per wavefront with the step loop
set -u
TAG=${1:-r04h}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
run() {
  local label=$1 cfg=$2; shift 2
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --config $cfg --no-extra-configs --no-cpu-baseline --no-second-window --steps 300 "$@" > $OUT/b_$label.json 2> $OUT/b_$label.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_$label.json").read().strip().splitlines()[-1])
    print("$label:", round(r["value"] / 1e6, 3), "M  ms/step", round(r["ms_per_step"], 4), "spl", r["config"]["steps_per_launch"], "cohorts", r["config"]["cohorts"], "pack", r["config"]["envs_per_wavefront"], "lds", r["config"]["lds_bytes_per_env"], "kernel_ms", round(r["roofline"]["kernel_ms"], 4))
except Exception as ex:
    print("$label: FAILED", ex); print(open("$OUT/b_$label.err").read()[-500:])
PY
}
for pk in 2 4 8; do
  run c3_p$pk c3 A=1 -- --pack $pk
  run c3_p${pk}_lds c3 MJH_FORCE_BIG=0 -- --pack $pk
done
run c3_p8_lds_c2 c3 MJH_FORCE_BIG=0 -- --pack 8 --cohorts 2
for pk in 1 2 4; do
  run c5_p$pk c5 A=1 -- --pack $pk
done
run c5_p2_lds c5 MJH_FORCE_BIG=0 -- --pack 2
run c5_p4_lds c5 MJH_FORCE_BIG=0 -- --pack 4
run c5_p1_c3 c5 A=1 -- --pack 1 --cohorts 3
run c5_p2_c3 c5 A=1 -- --pack 2 --cohorts 3
