set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
{
for r in 1 2; do
tools/s24_quick.sh c2_qd3 --config c2
MJHIP_LIB=build_exp/qd2/libmjhip.so tools/s24_quick.sh c2_qd2 --config c2
done
MJHIP_LIB=build_exp/qd2/libmjhip.so tools/s24_quick.sh c2_qd2_c2 --config c2 --cohorts 2
MJHIP_LIB=build_exp/qd2/libmjhip.so tools/s24_quick.sh c2_qd2_c4 --config c2 --cohorts 4
tools/s24_quick.sh c2_qd3_c2 --config c2 --cohorts 2
tools/s24_quick.sh c2_qd3_c4 --config c2 --cohorts 4
} > gpurun_out/r05s6/qd.log 2>&1
cat gpurun_out/r05s6/qd.log
