#!/bin/bash
# S24D: what the contact capacity costs (kernel trace per variant)
set -u
TAG=${1:-r04g}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "64 1" "72 1" "80 1" "80 0" "96 1"; do
  set -- $v; mc=$1; win=$2
  rm -rf /tmp/tr_${mc}_${win}
  MJH_WINDOW=$win rocprofv3 --kernel-trace --stats -d /tmp/tr_${mc}_$win -o t -- python $ROOT/bench.py --config s24d --maxcon $mc --steps 40 --warmup 10 --no-cpu-baseline --no-second-window --no-extra-configs > $OUT/b_${mc}_$win.json 2> $OUT/b_${mc}_$win.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_${mc}_$win.json").read().strip().splitlines()[-1])
    print("maxcon $mc window $win:", round(r["value"] / 1e6, 3), "M  ms/step", round(r["ms_per_step"], 4), "nefc", round(r["config"]["mean_nefc"], 1), "max", r["config"]["max_nefc"], "max ncon", r["config"]["max_ncon"], "overflow", r["config"]["overflow_envs"], "lds", r["config"]["lds_bytes_per_env"])
except Exception as ex:
    print("FAILED", ex); print(open("$OUT/b_${mc}_$win.err").read()[-600:])
PY
  python $ROOT/tools/kstats.py /tmp/tr_${mc}_$win 120 2 2>&1 | grep -E "pos [01]|sequence period" | head -4
done
