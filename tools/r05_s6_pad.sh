set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
{
tools/s24_quick.sh c4 --config c4
MJH_LDS_PAD=1600 tools/s24_quick.sh c4_pad1600 --config c4
MJH_LDS_PAD=4000 tools/s24_quick.sh c4_pad4000 --config c4
tools/s24_quick.sh c2 --config c2
MJH_LDS_PAD=2600 tools/s24_quick.sh c2_pad2600 --config c2
tools/s24_quick.sh c5 --config c5
MJH_LDS_PAD=2000 tools/s24_quick.sh c5_pad2000 --config c5
tools/s24_quick.sh c3 --config c3
MJH_LDS_PAD=2000 tools/s24_quick.sh c3_pad2000 --config c3
} > gpurun_out/r05s6/pad.log 2>&1
cat gpurun_out/r05s6/pad.log
