# articulated models of at most 64 bodies: the level loops' table entries fetched once in front of the loops (FK, RNE): bitwise A/B, throughput
set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
H=build_exp/head/libmjhip.so
{
for c in c3 c5; do
python tools/state_hash.py $c 1024 60 2>&1 | grep STATEHASH
MJHIP_LIB=$H python tools/state_hash.py $c 1024 60 2>&1 | grep STATEHASH
done
for r in 1 2; do
for c in c3 c5; do
tools/s24_quick.sh ${c}_new --config $c
MJHIP_LIB=$H tools/s24_quick.sh ${c}_head --config $c
done
done
tools/s24_quick.sh c4_new --config c4
MJHIP_LIB=$H tools/s24_quick.sh c4_head --config c4
tools/s24_quick.sh s24_new
MJHIP_LIB=$H tools/s24_quick.sh s24_head
python tools/c3_stage_profile.py 2>&1 | grep -v amdgpu | sed -n 2,5p
python tools/c3_stage_profile.py 2>&1 | grep -v amdgpu | grep "vel stage"
timeout 1500 python -m pytest tests -m gpu -x -q -k "tree or fuzz or fixture or robot or loop or pendulum or arm or c3 or c5 or c1 or kat" 2>&1 | tail -5
} > gpurun_out/r05s6/fkpre.log 2>&1
cat gpurun_out/r05s6/fkpre.log
