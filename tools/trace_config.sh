#!/bin/bash
# kernel-trace only (no PMC passes) of one bench config with extra bench args.  usage: tools/trace_config.sh <config> <tag> "<extra args>"
set -u
CFG=${1:-c4}; TAG=${2:-t}; EXTRA=${3:-}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
RAW=/tmp/trace_${TAG}_${CFG}
OUT=$ROOT/gpurun_out/profiles
rm -rf $RAW; mkdir -p $RAW $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --config $CFG --steps 100 --warmup 20 --no-cpu-baseline --no-second-window --no-extra-configs $EXTRA"
rocprofv3 --kernel-trace --stats -d $RAW/trace -o trace -- $BENCH > $RAW/bench_trace.json 2> $RAW/trace.log
python $ROOT/tools/summarize_config.py $RAW $OUT $TAG $CFG | head -24
rm -rf $RAW
