#!/bin/bash
# C2: the solve launch at three waves per SIMD (168 VGPRs, spills) against two (211 VGPRs)
set -u
TAG=${1:-r04t}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for lib in mujoco_sim_amd build_exp; do
 for c in 3 6; do
  MJHIP_LIB=$ROOT/$lib/libmjhip.so timeout 400 python bench.py --config c2 --cohorts $c --no-extra-configs --no-cpu-baseline --no-second-window --steps 40 --warmup 5 > $OUT/b_${lib}_$c.json 2> $OUT/b_${lib}_$c.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_${lib}_$c.json").read().strip().splitlines()[-1])
    print("$lib cohorts $c:", round(r["value"] / 1e6, 4), "M  ms/step", round(r["ms_per_step"], 4), "sweeps", round(r["config"]["mean_solver_iter"], 1), "nefc", round(r["config"]["mean_nefc"], 1), "kernel_ms", round(r["roofline"]["kernel_ms"], 4))
except Exception as ex:
    print("$lib: FAILED", ex); print(open("$OUT/b_${lib}_$c.err").read()[-500:])
PY
 done
done
