#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04y; mkdir -p $OUT
cd $ROOT
for thr in 112 104 96 88 80 72; do
  MJH_WINDOW32=$thr timeout 300 python bench.py --config s24 --no-extra-configs --no-cpu-baseline --no-second-window --steps 100 > $OUT/b_$thr.json 2> $OUT/b_$thr.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_$thr.json").read().strip().splitlines()[-1])
    print("threshold $thr:", round(r["value"] / 1e6, 3), "M  ms/step", round(r["ms_per_step"], 4), "kernel_ms", round(r["roofline"]["kernel_ms"], 4))
except Exception as ex:
    print("FAILED", ex)
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/tr -o t -- python $ROOT/bench.py --config s24 --steps 60 --warmup 10 --no-cpu-baseline --no-second-window --no-extra-configs > $OUT/b_trace.json 2> $OUT/trace.err
python $ROOT/tools/kstats.py /tmp/tr 240 2 2>&1 | grep -E "pos [01]|sequence period|mean gap" | head -8
