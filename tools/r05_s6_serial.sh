# short trees: a subtree on one lane, body after body (FK, RNE forward) instead of level by level: A/B against the level form (MJH_TREE_SERIAL=0) and the build before
set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
H=build_exp/head/libmjhip.so
{
for c in c3 c5 c4; do
python tools/state_hash.py $c 512 60 2>&1 | grep STATEHASH
MJH_TREE_SERIAL=0 python tools/state_hash.py $c 512 60 2>&1 | grep STATEHASH
MJHIP_LIB=$H python tools/state_hash.py $c 512 60 2>&1 | grep STATEHASH
done
python tools/robot_err.py tiago 2>&1 | grep -v amdgpu | tail -2
MJHIP_LIB=$H python tools/robot_err.py tiago 2>&1 | grep -v amdgpu | tail -2
for r in 1 2; do
for c in c3 c5; do
tools/s24_quick.sh ${c}_serial --config $c
MJH_TREE_SERIAL=0 tools/s24_quick.sh ${c}_level --config $c
MJHIP_LIB=$H tools/s24_quick.sh ${c}_head --config $c
done
done
python tools/c3_stage_profile.py 2>&1 | grep -v amdgpu | sed -n 1,16p
} > gpurun_out/r05s6/serial.log 2>&1
cat gpurun_out/r05s6/serial.log
