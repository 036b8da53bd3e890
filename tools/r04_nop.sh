#!/bin/bash
# the row chain without its two wait states (-DMJH_NO_ROW_NOP, build_exp/libmjhip.so): do the results change, what does it buy
set -u
TAG=${1:-r04k}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
cat > /tmp/traj.py <<'PY'
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import mujoco_sim_amd as ms
m = ms.scene("s24"); e = ms.Engine(m, 2048); e.load_s24(); e.set_cohorts(3)
out = []
for k in range(6):
    e.step(100); t, q, v, w = e.get_state(); out.append(np.concatenate([q, v, w], axis=1)); st = e.get_stats()
np.save(sys.argv[2], np.array(out)); print("sweeps", st[:, 2].mean(), "nefc", st[:, 1].mean())
PY
python /tmp/traj.py $ROOT $OUT/traj_nop.npy
MJHIP_LIB=$ROOT/build_exp/libmjhip.so python /tmp/traj.py $ROOT $OUT/traj_nonop.npy
python - <<PY
import numpy as np
a = np.load("$OUT/traj_nop.npy"); b = np.load("$OUT/traj_nonop.npy")
for k in range(len(a)):
    print("after", 100 * (k + 1), "steps: bitwise equal", np.array_equal(a[k], b[k]), "max |diff|", float(np.abs(a[k] - b[k]).max()), "finite", bool(np.isfinite(b[k]).all()))
PY
MJHIP_LIB=$ROOT/build_exp/libmjhip.so timeout 600 python -m pytest tests/test_gpu_teacher_forced.py tests/test_gpu_round4.py -m gpu -x -q -k "s24 or window" > $OUT/pytest_nonop.log 2>&1; echo "pytest (no nop) rc=$?"; tail -3 $OUT/pytest_nonop.log | cut -c1-300
for lib in mujoco_sim_amd build_exp; do
  MJHIP_LIB=$ROOT/$lib/libmjhip.so timeout 300 python bench.py --config s24 --no-extra-configs --no-cpu-baseline --no-second-window --steps 100 > $OUT/b_$lib.json 2> $OUT/b_$lib.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_$lib.json").read().strip().splitlines()[-1])
    print("$lib:", round(r["value"] / 1e6, 3), "M  ms/step", round(r["ms_per_step"], 4), "sweeps", round(r["config"]["mean_solver_iter"], 2), "nefc", round(r["config"]["mean_nefc"], 2), "kernel_ms", round(r["roofline"]["kernel_ms"], 4))
except Exception as ex:
    print("$lib: FAILED", ex); print(open("$OUT/b_$lib.err").read()[-500:])
PY
done
