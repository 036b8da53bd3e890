# window kernel (XL instance): tiles of the windows from WN_TILE_LDS0 on in LDS as well — A/B against the committed build (cross tiles only)
set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
H=build_exp/head/libmjhip.so
{
python tools/state_hash.py s24 1024 450 2>&1 | grep STATEHASH
MJHIP_LIB=$H python tools/state_hash.py s24 1024 450 2>&1 | grep STATEHASH
for r in 1 2 3; do
tools/s24_quick.sh s24_new
MJHIP_LIB=$H tools/s24_quick.sh s24_head
done
for v in t0 t4; do [ -f build_exp/$v/libmjhip.so ] && MJHIP_LIB=build_exp/$v/libmjhip.so tools/s24_quick.sh s24_$v; done
tools/s24_quick.sh s24d_new --config s24d
} > gpurun_out/r05s6/tilelds.log 2>&1
cat gpurun_out/r05s6/tilelds.log
