#!/bin/bash
# HBM traffic of the bench command (PMC passes only): tools/traffic_gpu.sh <tag>  -> gpurun_out/prof_<tag>/pmc_{fetch,write}
set -u
TAG=${1:-t}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 100 --warmup 400 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- $BENCH > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- $BENCH > $OUT/bench_write.json 2> $OUT/write.log
tail -c 300 $OUT/bench_write.json
