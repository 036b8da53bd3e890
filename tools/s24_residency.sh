#!/bin/bash
# S24 residency experiment (HISTORY.md Round 3): the default build (235 VGPRs: 2 waves / SIMD, records in registers) against the
# A/B build -DPP_NRC=1 -DPP_NSU=4 -DMJH_STEP_WAVES=3 (168 VGPRs: 3 waves / SIMD), each at contact capacities 40 / 28 / 24 / 20
# (LDS per env -> workgroups per CU).   usage: tools/s24_residency.sh <tag>
set -u
TAG=${1:-res}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for lib in default exp; do
  for mc in 40 28 24 20; do
    for envs in 4096 16384; do
      if [ $lib = exp ]; then export MJHIP_LIB=$ROOT/mujoco_sim_amd/build_exp/libmjhip.so; else unset MJHIP_LIB; fi
      python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs --no-second-window --maxcon $mc --envs-per-gpu $envs 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=r['config']
print('$lib maxcon $mc envs $envs: %.3f M env-steps/s  %.4f ms/step  kernel %.4f ms  lds %d B  overflow %d  mean_ncon %.1f' % (r['value']/1e6, r['ms_per_step'], r['roofline']['kernel_ms'], c['lds_bytes_per_env'], c['overflow_envs'], c['mean_ncon']))" | tee -a $OUT/residency.txt
    done
  done
done
