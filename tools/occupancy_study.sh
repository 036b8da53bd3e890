#!/bin/bash
# S24 step time against the number of resident waves (one cohort, so a launch is one batch of waves): tells a per-wave issue bound
# (time flat up to the slot count, then steps) from a SIMD / LDS throughput bound (time proportional to the wave count).
# usage (GPU box): tools/occupancy_study.sh > gpurun_out/occupancy_study.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for n in 256 512 1024 1536 2048 2304 3072 4096 6144 8192; do
  python bench.py --envs-per-gpu $n --cohorts 1 --steps 100 --no-cpu-baseline --no-second-window --no-gather 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('envs %5d  ms/step %.4f  kernel_ms %.4f  env-steps/s %.3f M  mean_ncon %.2f' % (d['config']['envs_per_gpu'], d['ms_per_step'], r['kernel_ms'], d['value']/1e6, d['config']['mean_ncon']))"
done
