#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + HBM PMC counters for the bench command.
# usage: tools/profile_gpu.sh <tag>    -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 100 --warmup 400 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- $BENCH > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- $BENCH > $OUT/bench_write.json 2> $OUT/write.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq -o sq -- $BENCH > $OUT/bench_sq.json 2> $OUT/sq.log
find $OUT -name "*.csv" | head -40
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
tail -2 $OUT/bench_trace.json
