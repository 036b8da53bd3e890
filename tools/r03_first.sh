#!/bin/bash
# round 3, first GPU session: teacher-forced parity tests, the driver-style bench line (with the other configs + literal loop), the group host
set -u
TAG=${1:-r03a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
(nproc; python -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | head -20; rocm-smi --showid | head -20) > $OUT/host.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -s > $OUT/pytest_tf.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_tf.log
grep "TEACHER-FORCED\|passed\|failed\|Error\|assert" $OUT/pytest_tf.log | tail -30
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"; tail -c 6000 $OUT/bench_driver.json; tail -3 $OUT/bench_driver.err
timeout 600 python bench.py --gpus 2 --host group --group-devices 0,0 --steps 60 --warmup 5 --envs-per-gpu 2048 > $OUT/bench_group.json 2> $OUT/bench_group.err; echo "group rc=$?"; tail -c 2500 $OUT/bench_group.json; tail -3 $OUT/bench_group.err
timeout 600 python bench.py --gpus 1 --host group --group-devices 0 --steps 60 --warmup 5 > $OUT/bench_group1.json 2> $OUT/bench_group1.err; echo "group1 rc=$?"; tail -c 1500 $OUT/bench_group1.json; tail -3 $OUT/bench_group1.err
