# round 6: is C2's solve launch bound by the stream of its operand pools from HBM?  env-steps/s against the number of envs (working set 95 KB per env
# against the 256 MB Infinity Cache) and cohorts
cd ${GRAFT_REPO_ROOT:-/root/repo}
q() { L=$1; shift; python bench.py --config c2 --steps 500 --warmup 5 --no-extra-configs --no-cpu-baseline --no-second-window "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value']/1e6,4), 'M  ms/step', round(d['ms_per_step'],4), 'chain_ms', round(d['roofline']['kernel_ms'],4), 'ncon', round(d['config']['mean_ncon'],1))"; }
for n in 512 1024 2048 3072 4096; do q c2_envs$n --envs-per-gpu $n; done
for c in 2 8; do q c2_envs2048_c$c --envs-per-gpu 2048 --cohorts $c; done
q c2_c8 --cohorts 8
