#!/bin/bash
# split API through the window chain: tests, literal loop
set -u
TAG=${1:-r04u}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -s -k "split_api_hands" > $OUT/pytest_split.log 2>&1; echo "pytest split rc=$?"; grep -E "SPLIT-HANDOVER|passed|failed|Error|assert" $OUT/pytest_split.log | cut -c1-400 | tail -8
timeout 1800 python -m pytest tests -m gpu -x -q -k "split or literal or window or s24 or cohort or host or bench" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log | cut -c1-300
for ho in 1 0; do
  MJH_SPLIT_HANDOVER=$ho timeout 300 python tools/literal_loop.py > $OUT/literal_$ho.txt 2>&1; echo "handover $ho:"; tail -4 $OUT/literal_$ho.txt | cut -c1-300
done
