cd ${GRAFT_REPO_ROOT:-/root/repo}
q() { L=$1; shift; python bench.py --config c4 --steps 300 --warmup 5 --no-extra-configs --no-cpu-baseline --no-second-window "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value']/1e6,4), 'M  ms/step', round(d['ms_per_step'],4), 'chain_ms', round(d['roofline']['kernel_ms'],4), 'sweeps', round(d['config']['mean_solver_iter'],3))"; }
q c4_default
for c in 2 4 5 6; do q c4_c$c --cohorts $c; done
for mi in 16 24 48; do MJH_DENSE_MIN_ITER=$mi q c4_minit$mi; done
for oe in 16 64; do MJH_ORDER_EVERY=$oe q c4_oe$oe; done
q c4_default2
