#!/bin/bash
# S24 bench line for 1..4 env cohorts (separate HIP streams).  usage: tools/cohort_sweep.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/${1:-coh}; mkdir -p $OUT; cd $ROOT
for c in 1 2 3 4; do
  timeout 300 python bench.py --cohorts $c --cpu-seconds 0.5 > $OUT/c$c.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("$OUT/c$c.json").read().strip().splitlines()[-1])
print("cohorts $c: %.3f M  ms %.3f  kernel_ms %.3f" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["kernel_ms"]))
PY
done
