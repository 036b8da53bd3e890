#!/bin/bash
# S24D through the window chain at contact capacities 64 / 80 / 96 / 128: throughput, overflow, assemble LDS
set -u
TAG=${1:-r05c}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
for mc in ${CAPS:-64 80 96 128}; do
  python $ROOT/bench.py --config s24d --maxcon $mc --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline --no-second-window --no-extra-configs > $OUT/b_$mc.json 2> $OUT/b_$mc.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_$mc.json").read().strip().splitlines()[-1])
    c = r["config"]
    print("maxcon $mc:", round(r["value"] / 1e6, 3), "M  ms/step", round(r["ms_per_step"], 4), "kernel_ms", round(r["roofline"]["kernel_ms"], 4), "nefc", round(c["mean_nefc"], 1), "max", c["max_nefc"], "max ncon", c["max_ncon"], "overflow", c["overflow_envs"], "lds", c["lds_bytes_per_env"])
except Exception as ex:
    print("FAILED", ex); print(open("$OUT/b_$mc.err").read()[-800:])
PY
done
