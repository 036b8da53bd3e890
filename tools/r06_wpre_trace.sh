#!/bin/bash
set -u; cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in head ${VARIANTS:-w6}; do
  L="X=1"; [ $v != head ] && L="MJHIP_LIB=$R/build_exp/$v/libmjhip.so"
  for c in s24 s24d; do
    rm -rf /tmp/tr_${v}_$c
    env $L rocprofv3 --kernel-trace --stats -d /tmp/tr_${v}_$c -o t -- python $R/bench.py --config $c --steps 100 --warmup 20 --no-cpu-baseline --no-second-window --no-extra-configs > /dev/null 2>&1
    f=$(find /tmp/tr_${v}_$c -name "*kernel_stats.csv" | head -1)
    echo "== $v $c"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:3]:
    print("   ", r["Name"][:60], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1))
PY
  done
done
