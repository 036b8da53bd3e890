# S24: what the assemble launch takes with the chip to itself (one cohort: nothing overlaps), and how that depends on its LDS
set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
{
echo "== one cohort"; bash tools/trace_config.sh s24 r05x1 "--cohorts 1" 2>&1 | grep -i "mean\|Summed" | cut -c1-220
echo "== one cohort, +18 KB of LDS per assemble workgroup"; MJH_WPRE_LDS_PAD=18000 bash tools/trace_config.sh s24 r05x2 "--cohorts 1" 2>&1 | grep -i "mean\|Summed" | cut -c1-220
echo "== one cohort, 2048 envs"; bash tools/trace_config.sh s24 r05x3 "--cohorts 1 --envs-per-gpu 2048" 2>&1 | grep -i "mean\|Summed" | cut -c1-220
echo "== one cohort, 1024 envs"; bash tools/trace_config.sh s24 r05x4 "--cohorts 1 --envs-per-gpu 1024" 2>&1 | grep -i "mean\|Summed" | cut -c1-220
echo "== three cohorts"; bash tools/trace_config.sh s24 r05x5 "" 2>&1 | grep -i "mean\|Summed" | cut -c1-220
} > gpurun_out/r05s6/alone.log 2>&1
cat gpurun_out/r05s6/alone.log
