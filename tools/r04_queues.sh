#!/bin/bash
# more cohorts than hardware queues?  GPU_MAX_HW_QUEUES x cohorts on S24
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04z; mkdir -p $OUT
cd $ROOT
for q in 4 8 16; do
 for c in 3 4 6 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --config s24 --cohorts $c --no-extra-configs --no-cpu-baseline --no-second-window --steps 100 > $OUT/b_${q}_$c.json 2> $OUT/b_${q}_$c.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_${q}_$c.json").read().strip().splitlines()[-1])
    print("hw queues $q cohorts $c:", round(r["value"] / 1e6, 3), "M  ms/step", round(r["ms_per_step"], 4), "kernel_ms", round(r["roofline"]["kernel_ms"], 4))
except Exception as ex:
    print("FAILED", ex)
PY
 done
done
