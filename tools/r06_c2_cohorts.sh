cd ${GRAFT_REPO_ROOT:-/root/repo}
q() { L=$1; shift; python bench.py --config c2 --steps 500 --warmup 5 --no-extra-configs --no-cpu-baseline --no-second-window "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value']/1e6,4), 'M  ms/step', round(d['ms_per_step'],4), 'chain_ms', round(d['roofline']['kernel_ms'],4), 'ncon', round(d['config']['mean_ncon'],1))"; }
for c in 1 2 3 4 6; do q c2_c$c --cohorts $c; done
