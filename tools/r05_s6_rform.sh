set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
{
for r in 1 2; do
tools/s24_quick.sh c4_rform --config c4
MJHIP_LIB=build_exp/head/libmjhip.so tools/s24_quick.sh c4_head --config c4
done
timeout 1500 python -m pytest tests -m gpu -x -q -k "dense or robot or c4 or pr2 or fixture or tiago or hsr or ridgeback or armar" 2>&1 | tail -6
} > gpurun_out/r05s6/rform.log 2>&1
cat gpurun_out/r05s6/rform.log
