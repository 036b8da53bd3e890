#!/bin/bash
# round 6: the assemble-only launch at 80 / 96 registers (a wave that fits on a SIMD beside a window wavefront) against the 128-register build
set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06b
VARIANTS=${VARIANTS:-"w6 w5"}
for r in 1 2; do
  tools/s24_quick.sh s24_head
  for v in $VARIANTS; do MJHIP_LIB=build_exp/$v/libmjhip.so tools/s24_quick.sh s24_$v; done
  tools/s24_quick.sh s24d_head --config s24d --steps 200
  for v in $VARIANTS; do MJHIP_LIB=build_exp/$v/libmjhip.so tools/s24_quick.sh s24d_$v --config s24d --steps 200; done
done
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in head $VARIANTS; do
  L=""; [ $v != head ] && L="MJHIP_LIB=$R/build_exp/$v/libmjhip.so"
  for c in s24 s24d; do
    rm -rf /tmp/tr_${v}_$c
    env $L rocprofv3 --kernel-trace --stats -d /tmp/tr_${v}_$c -o t -- python $R/bench.py --config $c --steps 100 --warmup 20 --no-cpu-baseline --no-second-window --no-extra-configs > /dev/null 2>&1
    f=$(find /tmp/tr_${v}_$c -name "*kernel_stats.csv" | head -1)
    echo "== $v $c"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:4]:
    print("   ", r["Name"][:60], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1))
PY
  done
done
