#!/bin/bash
# SQ counters of the C2 kernels (mjh_solve_kernel and the assemble / integrate launches): where do a wave's cycles go?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
RAW=/tmp/c2sq; rm -rf $RAW; mkdir -p $RAW $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export MJH_SOLVE_WAVES=${2:-1}
BENCH="python $ROOT/bench.py --config ${1:-c2} --steps 30 --warmup 10 --no-cpu-baseline --no-second-window"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $RAW/p1 -o p1 -- $BENCH > $RAW/b1.json 2> $RAW/p1.log
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY -d $RAW/p2 -o p2 -- $BENCH > $RAW/b2.json 2> $RAW/p2.log
python - <<PY > $ROOT/gpurun_out/${1:-c2}_sq.txt
import glob, sqlite3
for p in ("p1", "p2"):
    f = glob.glob("$RAW/%s/**/*.db" % p, recursive=True)
    if not f: print("no db for", p); continue
    con = sqlite3.connect(f[0])
    names = sorted(r[0] for r in con.execute("select distinct counter_name from counters_collection"))
    for kern in ("mjh_solve_kernel", "mjh_step_kernel"):
        for n in names:
            rows = con.execute("select value from counters_collection where kernel_name like ? and counter_name=? order by start", ("%" + kern + "%", n)).fetchall()
            last = [r[0] for r in rows[-60:]]
            if last: print("%-18s %-26s %.4e per launch   %.4e per env" % (kern, n, sum(last)/len(last), sum(last)/len(last)/2048))
PY
cat $ROOT/gpurun_out/${1:-c2}_sq.txt
