# C4 same-call A/B of library variants: tools/r06_c4_ab.sh <variant name under build_exp> [more variants]   (default library first and last)
cd ${GRAFT_REPO_ROOT:-/root/repo}
q() { L=$1; shift; python bench.py --config c4 --steps 300 --warmup 5 --no-extra-configs --no-cpu-baseline --no-second-window "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value']/1e6,4), 'M  ms/step', round(d['ms_per_step'],4), 'chain_ms', round(d['roofline']['kernel_ms'],4), 'sweeps', round(d['config']['mean_solver_iter'],3))"; }
unset MJHIP_LIB; q default
for v in "$@"; do export MJHIP_LIB=$PWD/build_exp/$v/libmjhip.so; q $v; q $v; done
unset MJHIP_LIB; q default
