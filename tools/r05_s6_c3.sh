set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
{
for r in 1 2; do
tools/s24_quick.sh c3_new --config c3
MJHIP_LIB=build_exp/head/libmjhip.so tools/s24_quick.sh c3_head --config c3
done
tools/s24_quick.sh c3_new_c2 --config c3 --cohorts 2
tools/s24_quick.sh c3_new_c4 --config c3 --cohorts 4
tools/s24_quick.sh c5_new --config c5
MJHIP_LIB=build_exp/head/libmjhip.so tools/s24_quick.sh c5_head --config c5
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "c3 or arm or pendulum or c1 or loop" 2>&1 | tail -5
} > gpurun_out/r05s6/c3.log 2>&1
cat gpurun_out/r05s6/c3.log
