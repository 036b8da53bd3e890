#!/bin/bash
# S24 throughput against resident workgroups per CU: the contact capacity sets the LDS footprint per env (344 B per contact).
# Environments that overflow a small capacity drop contacts (flagged) — this is an occupancy experiment, not a physics run.
# usage: tools/occupancy_sweep.sh "<envs list>" "<maxcon list>" "<cohort list>"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for ne in ${1:-4096}; do for mc in ${2:-16 20 24 28 32 40 48}; do for co in ${3:-2}; do
  python bench.py --maxcon $mc --envs-per-gpu $ne --cohorts $co --steps 100 --no-cpu-baseline --no-second-window 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d['config']; lds = c['lds_bytes_per_env']; gran = -(-lds // 1280) * 1280
print('envs %5d cohorts %d maxcon %3d  lds %6d B (%2d WG/CU by LDS)  %.3f M env-steps/s  kernel %.3f ms  overflow envs %d  mean ncon %.1f' % ($ne, $co, $mc, lds, 163840 // gran, d['value'] / 1e6, d['roofline']['kernel_ms'], c['overflow_envs'], c['mean_ncon']))"
done; done; done
