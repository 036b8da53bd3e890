#!/bin/bash
# S24 in the many-body layout (MJH_FORCE_BIG=1: assemble -> mjh_solve_kernel -> integrate): what the chain's launches cost
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04c; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MJH_FORCE_BIG=1
BENCH="python $ROOT/bench.py --config s24 --steps 60 --warmup 10 --no-cpu-baseline --no-second-window --no-extra-configs"
rocprofv3 --kernel-trace --stats -d /tmp/bc -o trace -- $BENCH > $OUT/bench.json 2> $OUT/trace.log
tail -c 600 $OUT/bench.json
python $ROOT/tools/kstats.py /tmp/bc 270 3 > $OUT/kstats.txt 2>&1; cat $OUT/kstats.txt

