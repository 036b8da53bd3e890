"""C3 / C5: is the step loop bound by the host's per-step calls?  mjh_step(n) issues n x cohorts launches from C++ in one call.
python tools/c3_hostbound.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import mujoco_sim_amd as ms
base = ms.scene("arm7", 1); m = base.replicate(4)
rows = 2048
e = ms.Engine(m, rows); e.set_controlled_dofs(np.ones(m.nv, dtype=np.int32)); e.set_pd_controller(200.0, 50.0); e.set_cohorts(3)
lo, hi = base.array("jnt_range").reshape(-1, 2).T
e.set_pd_target(np.random.default_rng(0).uniform(lo, hi, size=(rows * 4, base.nv)).reshape(rows, -1))
e.step(200, True); e.synchronize()
for chunk in (1, 2, 5, 20, 100):
    n = 400
    t0 = time.perf_counter()
    for _ in range(n // chunk):
        e.step(chunk, True)
    e.synchronize()
    dt = time.perf_counter() - t0
    print(f"mjh_step({chunk}) per call: {rows * 4 * n / dt / 1e6:.1f} M env-steps/s, {dt / n * 1e3:.4f} ms per step")
