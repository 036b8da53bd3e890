#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04e; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --config s24 --steps 60 --warmup 10 --no-cpu-baseline --no-second-window --no-extra-configs $*"
rocprofv3 --kernel-trace --stats -d /tmp/bc -o trace -- $BENCH > $OUT/bench.json 2> $OUT/trace.log
python - <<PY
import json
r = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("S24:", round(r["value"] / 1e6, 3), "M env-steps/s  ms/step", round(r["ms_per_step"], 4))
PY
python $ROOT/tools/kstats.py /tmp/bc 360 2 > $OUT/kstats.txt 2>&1; cat $OUT/kstats.txt
