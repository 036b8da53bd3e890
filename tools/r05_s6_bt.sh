# C4: threads per env of the dense build kernel (512 default): its waves fill the SIMDs (5 per SIMD at 86 registers) while it runs
set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
{
for r in 1 2; do
tools/s24_quick.sh c4_bt512 --config c4
MJHIP_LIB=build_exp/bt256/libmjhip.so tools/s24_quick.sh c4_bt256 --config c4
MJHIP_LIB=build_exp/bt1024/libmjhip.so tools/s24_quick.sh c4_bt1024 --config c4
done
tools/s24_quick.sh c4_c2 --config c4 --cohorts 2
tools/s24_quick.sh c4_c4 --config c4 --cohorts 4
MJH_DENSE_MIN_ITER=16 tools/s24_quick.sh c4_minit16 --config c4
MJH_DENSE_MIN_ITER=64 tools/s24_quick.sh c4_minit64 --config c4
} > gpurun_out/r05s6/bt.log 2>&1
cat gpurun_out/r05s6/bt.log
