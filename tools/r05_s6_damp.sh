# C4: the factor of M + h D formed by a wavefront of the dense build kernel instead of the integrate launch: bitwise A/B, throughput, tests
set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
{
python tools/state_hash.py c4 512 150 2>&1 | grep -v amdgpu.ids | tail -2
MJHIP_LIB=build_exp/head/libmjhip.so python tools/state_hash.py c4 512 150 2>&1 | grep -v amdgpu.ids | tail -2
for r in 1 2; do
tools/s24_quick.sh c4_new --config c4
MJHIP_LIB=build_exp/head/libmjhip.so tools/s24_quick.sh c4_head --config c4
done
timeout 300 python tools/c4_stage_profile.py 2>&1 | tail -12
timeout 1500 python -m pytest tests -m gpu -x -q -k "dense or robot or c4 or pr2 or fixture or tiago or hsr or ridgeback or armar or damp" 2>&1 | tail -6
} > gpurun_out/r05s6/damp.log 2>&1
cat gpurun_out/r05s6/damp.log
