#!/bin/bash
# C3 / C5 (small models): env-steps/s against instances per wavefront and cohorts.  usage (GPU box): tools/c3_sweep.sh "<packs>" "<cohorts>" [c3|c5]
CFG=${3:-c3}
for p in ${1:-1 2 4}; do for c in ${2:-1 2 3}; do
python bench.py --config $CFG --pack $p --cohorts $c --steps 100 --no-cpu-baseline --no-second-window --no-gather 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pack $p cohorts $c  %.2f M  ms %.4f kernel_ms %.4f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
