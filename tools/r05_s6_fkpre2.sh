set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
H=build_exp/head/libmjhip.so
{
for c in c3 c5 c4; do
python tools/state_hash.py $c 512 60 2>&1 | grep STATEHASH
MJHIP_LIB=$H python tools/state_hash.py $c 512 60 2>&1 | grep STATEHASH
done
python tools/robot_err.py tiago pr2 2>&1 | grep -v amdgpu | tail -4
MJHIP_LIB=$H python tools/robot_err.py tiago pr2 2>&1 | grep -v amdgpu | tail -4
for c in c3 c5; do
tools/s24_quick.sh ${c}_new --config $c
MJHIP_LIB=$H tools/s24_quick.sh ${c}_head --config $c
done
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
} > gpurun_out/r05s6/fkpre2.log 2>&1
cat gpurun_out/r05s6/fkpre2.log
