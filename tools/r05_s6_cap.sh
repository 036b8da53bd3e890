set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
{
for mc in 96 94 85 76 70; do tools/s24_quick.sh s24d_mc$mc --config s24d --maxcon $mc; done
tools/s24_quick.sh s24d_mc96 --config s24d --maxcon 96
} > gpurun_out/r05s6/cap.log 2>&1
cat gpurun_out/r05s6/cap.log
