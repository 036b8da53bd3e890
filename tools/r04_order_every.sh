#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04oe; mkdir -p $OUT
cd $ROOT
for cfg in s24 c2 c4; do
 for oe in 8 16 32 64; do
  MJH_ORDER_EVERY=$oe timeout 400 python bench.py --config $cfg --no-extra-configs --no-cpu-baseline --no-second-window --steps $([ $cfg = s24 ] && echo 200 || echo 60) > $OUT/b_${cfg}_$oe.json 2> $OUT/b_${cfg}_$oe.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_${cfg}_$oe.json").read().strip().splitlines()[-1])
    print("$cfg order every $oe:", round(r["value"] / 1e6, 4), "M  ms/step", round(r["ms_per_step"], 4), "kernel_ms", round(r["roofline"]["kernel_ms"], 4))
except Exception as ex:
    print("FAILED", ex)
PY
 done
done
