#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + HBM PMC passes (+ SQ pass) of one bench.py config, summarised
# ON the box (the raw .db files exceed what gpurun copies back).   usage: tools/profile_config.sh <config> <tag> [steps] [sq]
#   -> gpurun_out/profiles/<tag>_<config>_{summary.md,kernel_stats.csv,bench.json,traffic.json}
set -u
CFG=${1:-s24}; TAG=${2:-r02}; STEPS=${3:-100}; SQ=${4:-0}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
RAW=/tmp/prof_${TAG}_${CFG}
OUT=$ROOT/gpurun_out/profiles
rm -rf $RAW; mkdir -p $RAW $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --config $CFG --steps $STEPS --warmup 20 --no-cpu-baseline --no-second-window --no-extra-configs"
rocprofv3 --kernel-trace --stats -d $RAW/trace -o trace -- $BENCH > $RAW/bench_trace.json 2> $RAW/trace.log
rocprofv3 --pmc FETCH_SIZE -d $RAW/pmc_fetch -o fetch -- $BENCH > $RAW/bench_fetch.json 2> $RAW/fetch.log
rocprofv3 --pmc WRITE_SIZE -d $RAW/pmc_write -o write -- $BENCH > $RAW/bench_write.json 2> $RAW/write.log
if [ "$SQ" = "1" ]; then
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT -d $RAW/pmc_sq -o sq -- $BENCH > $RAW/bench_sq.json 2> $RAW/sq.log
  rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $RAW/pmc_sq2 -o sq2 -- $BENCH > $RAW/bench_sq2.json 2> $RAW/sq2.log
  tail -2 $RAW/sq2.log
fi
python $ROOT/tools/summarize_config.py $RAW $OUT $TAG $CFG
rm -rf $RAW
