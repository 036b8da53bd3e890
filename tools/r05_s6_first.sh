# round 5, session 6: sanity of the rebuilt library + S24D knobs + C4 stage clocks
set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
{
tools/s24_quick.sh s24
tools/s24_quick.sh s24d --config s24d
for nl in 0 1 2; do MJH_WN_NL=$nl tools/s24_quick.sh s24d_nl$nl --config s24d; done
tools/s24_quick.sh s24d_c2 --config s24d --cohorts 2
tools/s24_quick.sh s24d_c4 --config s24d --cohorts 4
tools/s24_quick.sh c4 --config c4
tools/s24_quick.sh c2 --config c2
timeout 300 python tools/c4_stage_profile.py 2>&1 | tail -40
} > gpurun_out/r05s6/first.log 2>&1
cat gpurun_out/r05s6/first.log
