#!/bin/bash
# round-2 GPU session: tests, issue-rate micro-benchmark, bench lines of every config.  usage: tools/r02_run.sh <tag>
set -u
TAG=${1:-r02a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 120 tools/valu_issue_bench > $OUT/valu_issue.txt 2>&1; cat $OUT/valu_issue.txt
timeout 600 python bench.py > $OUT/bench_s24.json 2> $OUT/bench_s24.err; tail -c 3000 $OUT/bench_s24.json
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_s24_driver.json 2> $OUT/bench_s24_driver.err; tail -c 1500 $OUT/bench_s24_driver.json
for c in c2 c3 c4 c5; do
  timeout 900 python bench.py --config $c --steps 100 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; echo "== $c rc=$?"; tail -c 2500 $OUT/bench_$c.json; tail -3 $OUT/bench_$c.err
done
