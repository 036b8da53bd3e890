# whole-library A/B: the in-tree build against build_exp/<name>/libmjhip.so — state hashes (S24, S24D), then S24 / S24D / C4 / C3 / C5 throughput in one call
# usage: tools/r06_lib_ab.sh <name>
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=${1:-prev}
for lib in cur $V; do
  if [ $lib = cur ]; then unset MJHIP_LIB; else export MJHIP_LIB=$PWD/build_exp/$lib/libmjhip.so; fi
  for c in s24 s24d c4 c3 c5 c2; do python tools/state_hash.py $c $([ $c = c2 ] && echo 256 || echo 1024) $([ $c = c2 ] && echo 150 || echo 300) 2>/dev/null | grep STATEHASH | sed "s/^/$lib /"; done
done
q() { L=$1; shift; python bench.py --no-extra-configs --no-cpu-baseline --no-second-window "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value']/1e6,4), 'M  ms/step', round(d['ms_per_step'],4), 'chain_ms', round(d['roofline']['kernel_ms'],4), 'sweeps', round(d['config']['mean_solver_iter'],3))"; }
for rep in 1 2; do
  for lib in cur $V; do
    if [ $lib = cur ]; then unset MJHIP_LIB; else export MJHIP_LIB=$PWD/build_exp/$lib/libmjhip.so; fi
    q "$lib s24" --steps 100 --warmup 20
    q "$lib s24d" --config s24d --steps 200 --warmup 20
    q "$lib c4" --config c4 --steps 300 --warmup 5
    q "$lib c3" --config c3 --steps 300 --warmup 5
    q "$lib c5" --config c5 --steps 300 --warmup 5
    q "$lib c2" --config c2 --steps 100 --warmup 5
  done
done
