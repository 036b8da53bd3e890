# S24 / S24D same-call A/B of window-unit variants (build_exp/<name>): state hashes first, then throughput      usage: tools/r06_win_ab.sh <variant> [<variant> ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  for c in s24 s24d; do MJHIP_LIB=$PWD/build_exp/$v/libmjhip.so python tools/state_hash.py $c 1024 300 2>/dev/null | grep STATEHASH | sed "s/^/$v /"; done
done
for rep in 1 2; do
  for v in "$@"; do
    MJHIP_LIB=$PWD/build_exp/$v/libmjhip.so bash tools/s24_quick.sh "$v s24" --steps 100 --warmup 20
    MJHIP_LIB=$PWD/build_exp/$v/libmjhip.so bash tools/s24_quick.sh "$v s24d" --config s24d --steps 200 --warmup 20
  done
done
