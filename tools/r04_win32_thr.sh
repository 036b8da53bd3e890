#!/bin/bash
set -u
TAG=${1:-r04m}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for thr in 104 96 92 88 80; do
 for c in 3; do
  MJH_WINDOW32=$thr timeout 300 python bench.py --config s24 --cohorts $c --no-extra-configs --no-cpu-baseline --no-second-window --steps 100 > $OUT/b_${thr}_${c}.json 2> $OUT/b_${thr}_${c}.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_${thr}_${c}.json").read().strip().splitlines()[-1])
    print("threshold $thr cohorts $c:", round(r["value"] / 1e6, 3), "M  ms/step", round(r["ms_per_step"], 4), "kernel_ms", round(r["roofline"]["kernel_ms"], 4))
except Exception as ex:
    print("FAILED", ex)
PY
 done
done
