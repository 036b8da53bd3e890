#!/bin/bash
# SQ issue / LDS counters of the step kernel (two rocprofv3 --pmc passes): tools/pmc_sq.sh <tag>
set -u
TAG=${1:-sq}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 50 --warmup 400 --no-cpu-baseline --cohorts 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS -d $OUT/p1 -o p1 -- $BENCH > $OUT/b1.json 2> $OUT/p1.log
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_IFETCH -d $OUT/p2 -o p2 -- $BENCH > $OUT/b2.json 2> $OUT/p2.log
python - <<PY
import glob, sqlite3
for p in ("p1", "p2"):
    f = glob.glob("$OUT/%s/*.db" % p)
    if not f: print("no db for", p); continue
    con = sqlite3.connect(f[0])
    names = [r[0] for r in con.execute("select distinct counter_name from counters_collection")]
    for n in sorted(names):
        rows = con.execute("select value from counters_collection where kernel_name like '%mjh_step_kernel%' and counter_name=? order by start", (n,)).fetchall()
        last = [r[0] for r in rows[-50:]]
        print("%-24s %.4e per launch   %.4e per wave" % (n, sum(last)/len(last), sum(last)/len(last)/4096))
PY
