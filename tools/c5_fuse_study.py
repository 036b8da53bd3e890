"""Launch-bound configs: ms per step of the C5 scene (4096 envs) against steps per call, hardware-queue count, launch timing and
the per-env initial spin — what the bench line's 0.11 ms per step is made of.   python tools/c5_fuse_study.py"""
import sys, os, time
if len(sys.argv) > 1 and sys.argv[1] == "q8":
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mujoco_sim_amd as ms
from mujoco_sim_amd.tables import load_model_tables
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m, z = load_model_tables(os.path.join(ROOT, "tests", "golden", "robot_c5_pendulum_bowl_mesh.npz"))
nenv = 4096
for spin in ("same", "per-env"):
    e = ms.Engine(m, nenv); e.set_controlled_dofs(z["controlled"].astype(np.int32))
    rng = np.random.default_rng(0)
    v = np.tile(z["qvel0"], (nenv, 1)) * (rng.uniform(0.5, 1.5, size=(nenv, 1)) if spin == "per-env" else 1.0)
    e.set_state(qvel=v)
    e.step(100, True); e.synchronize()
    for timing in (0, 1):
        e.set_launch_timing(timing)
        t0 = time.perf_counter()
        for _ in range(300): e.step(1, True)
        e.synchronize(); dt = time.perf_counter() - t0
        km = e.get_launch_timing() if timing else (0, 0)
        st = e.get_stats()
        print("queues", os.environ.get("GPU_MAX_HW_QUEUES", "default"), "spin", spin, "timing", timing, "ms/step %.4f" % (dt / 300 * 1e3), "kernel_ms %.4f" % km[0], "mean ncon %.2f" % st[:, 0].mean(), "iters %.1f" % st[:, 2].mean())
    e.close()
