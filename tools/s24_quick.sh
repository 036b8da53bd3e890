#!/bin/bash
# one S24 bench line (no extras): value, ms/step, event-timed chain   usage: [ENV=..] tools/s24_quick.sh <label> [bench args]
L=${1:-s24}; shift
python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --no-extra-configs --no-cpu-baseline --no-second-window "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value']/1e6,3), 'M  ms/step', round(d['ms_per_step'],4), 'chain_ms', round(d['roofline']['kernel_ms'],4))"
