set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
{
for r in 1 2 3; do
tools/s24_quick.sh c3_c3 --config c3 --cohorts 3
tools/s24_quick.sh c3_c2 --config c3 --cohorts 2
done
for r in 1 2; do
tools/s24_quick.sh c2_c3 --config c2 --cohorts 3
tools/s24_quick.sh c2_c4 --config c2 --cohorts 4
done
} > gpurun_out/r05s6/coh.log 2>&1
cat gpurun_out/r05s6/coh.log
