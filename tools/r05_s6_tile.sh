# window kernel: tile builds with the dof loop outermost (no compiler s_nop between multiply-adds into the same accumulator): bitwise A/B, throughput
set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
H=build_exp/head/libmjhip.so
{
for c in s24 s24d; do
python tools/state_hash.py $c 1024 450 2>&1 | grep STATEHASH
MJHIP_LIB=$H python tools/state_hash.py $c 1024 450 2>&1 | grep STATEHASH
done
for r in 1 2 3; do
tools/s24_quick.sh s24_new
MJHIP_LIB=$H tools/s24_quick.sh s24_head
done
for r in 1 2; do
tools/s24_quick.sh s24d_new --config s24d
MJHIP_LIB=$H tools/s24_quick.sh s24d_head --config s24d
done
} > gpurun_out/r05s6/tile.log 2>&1
cat gpurun_out/r05s6/tile.log
