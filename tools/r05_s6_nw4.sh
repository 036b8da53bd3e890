# S24 with a window kernel an assemble wave fits beside: 4 register windows (16-row form) + 2 (64-row form) = 348 registers, windows 5 / 6 from the LDS tier
set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
V=build_exp/nw4/libmjhip.so
{
tools/s24_quick.sh s24_head
for nl in 2 3; do MJHIP_LIB=$V MJH_WN_NL=$nl tools/s24_quick.sh s24_nw4_nl$nl; done
MJHIP_LIB=$V tools/s24_quick.sh s24_nw4_default
for w in 64 80; do MJHIP_LIB=$V MJH_WN_NL=2 MJH_WINDOW64=$w tools/s24_quick.sh s24_nw4_nl2_w64_$w; done
MJHIP_LIB=$V MJH_WN_NL=2 tools/s24_quick.sh s24_nw4_nl2_c4 --cohorts 4
MJHIP_LIB=$V MJH_WN_NL=2 tools/s24_quick.sh s24_nw4_nl2_c2 --cohorts 2
python tools/state_hash.py s24 1024 300
MJHIP_LIB=$V MJH_WN_NL=2 python tools/state_hash.py s24 1024 300
MJHIP_LIB=$V tools/s24_quick.sh s24d_nw4 --config s24d
} > gpurun_out/r05s6/nw4.log 2>&1
cat gpurun_out/r05s6/nw4.log
