#!/bin/bash
# short GPU session: the S24 parity tests, then the S24 bench line.  usage: tools/quick_gpu.sh <tag> [pytest -k expression]
set -u
TAG=${1:-q}
KEXP=${2:-s24}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXP" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -30 $OUT/pytest.log
timeout 600 python bench.py --cpu-seconds 2 > $OUT/bench_s24.json 2> $OUT/bench_s24.err; tail -c 2500 $OUT/bench_s24.json; tail -3 $OUT/bench_s24.err
