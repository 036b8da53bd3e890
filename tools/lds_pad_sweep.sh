#!/bin/bash
# How much does the number of resident envs per CU (LDS per env) matter?  MJH_LDS_PAD adds unused LDS.  usage: tools/lds_pad_sweep.sh <config> "<pads>"
CFG=${1:-c4}; PADS=${2:-"0 3000 6000"}
cd ${GRAFT_REPO_ROOT:-/root/repo}
for p in $PADS; do
  MJH_LDS_PAD=$p timeout 600 python bench.py --config $CFG --steps 100 --warmup 20 --no-cpu-baseline --no-second-window --no-extra-configs 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFG pad $p lds', r['config']['lds_bytes_per_env'], 'value', round(r['value']), 'ms', round(r['ms_per_step'], 3))"
done
