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
echo "--- bitwise: new build against HEAD's (s24d)"
python tools/state_hash.py s24d 1024 450
MJHIP_LIB=build_exp/head/libmjhip.so python tools/state_hash.py s24d 1024 450
echo "--- throughput"
for r in 1 2; do
tools/s24_quick.sh s24d_new --config s24d
MJHIP_LIB=build_exp/head/libmjhip.so tools/s24_quick.sh s24d_head --config s24d
done
tools/s24_quick.sh s24d_new_c3 --config s24d --cohorts 3
for w in 176 192 224; do MJH_WINDOW64=$w tools/s24_quick.sh s24d_w64_$w --config s24d; done
MJH_WPRE_LDS_PAD=1024 tools/s24_quick.sh s24d_pad1024 --config s24d
tools/s24_quick.sh s24_new
timeout 600 python -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -5
} > gpurun_out/r05s6/wpre2.log 2>&1
cat gpurun_out/r05s6/wpre2.log
