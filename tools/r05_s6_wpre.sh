# round 5, session 6: the assemble-only launch of window-only models without the schedule / X-extension LDS (S24D: 25.3 -> 19.9 KB)
set -u; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05s6
{
python - <<'PY'
import sys; sys.path.insert(0, ".")
import mujoco_sim_amd as ms
from mujoco_sim_amd import capi
L = capi.load()
m = ms.scene("s24pen", 0.175, 96)
print("s24d lds", L.mjh_query_lds_bytes(m.ptr), "assemble-only", L.mjh_query_lds_bytes_assemble(m.ptr))
PY
echo "--- bitwise: new build against HEAD's (s24d, s24)"
python tools/state_hash.py s24d 1024 450
MJHIP_LIB=build_exp/head/libmjhip.so python tools/state_hash.py s24d 1024 450
python tools/state_hash.py s24 1024 300
MJHIP_LIB=build_exp/head/libmjhip.so python tools/state_hash.py s24 1024 300
echo "--- throughput"
for r in 1 2; do
tools/s24_quick.sh s24d_new --config s24d
MJHIP_LIB=build_exp/head/libmjhip.so tools/s24_quick.sh s24d_head --config s24d
done
MJH_WPRE_SLIM2=0 tools/s24_quick.sh s24d_slim2off --config s24d
MJH_WPRE_LDS_PAD=5400 tools/s24_quick.sh s24d_pad5400 --config s24d
MJH_WPRE_LDS_PAD=12000 tools/s24_quick.sh s24d_pad12000 --config s24d
tools/s24_quick.sh s24d_new_c3 --config s24d --cohorts 3
tools/s24_quick.sh s24_new
MJHIP_LIB=build_exp/head/libmjhip.so tools/s24_quick.sh s24_head
MJH_WPRE_LDS_PAD=6000 tools/s24_quick.sh s24_pad6000
MJH_ORDER_EVERY=8 tools/s24_quick.sh s24d_oe8 --config s24d
MJH_ORDER_EVERY=16 tools/s24_quick.sh s24d_oe16 --config s24d
} > gpurun_out/r05s6/wpre.log 2>&1
cat gpurun_out/r05s6/wpre.log
