set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
{
tools/s24_quick.sh c5 --config c5
tools/s24_quick.sh c5_p2 --config c5 --pack 2
tools/s24_quick.sh c5_p2_c3 --config c5 --pack 2 --cohorts 3
tools/s24_quick.sh c5_p2_c1 --config c5 --pack 2 --cohorts 1
tools/s24_quick.sh c5_c3 --config c5 --cohorts 3
tools/s24_quick.sh c5_c4 --config c5 --cohorts 4
tools/s24_quick.sh c5_p4 --config c5 --pack 4
} > gpurun_out/r05s6/c5.log 2>&1
cat gpurun_out/r05s6/c5.log
