# S24: what LDS held by the window wavefronts costs (MJH_WN_NL forces the tier's allocation: 12 KB per window per wavefront, unused by S24's 16-row form) — the price side of "tiles / J^ records in LDS"
set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
{
for nl in 0 1 2 3; do MJH_WN_NL=$nl tools/s24_quick.sh s24_nl$nl; done
tools/s24_quick.sh s24_default
} > gpurun_out/r05s6/wnlds.log 2>&1
cat gpurun_out/r05s6/wnlds.log
