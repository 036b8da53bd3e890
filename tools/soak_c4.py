"""C4 soak (PR2 + world + object pool, dense row-space solver, computed-torque wrapper + mj_inverse every step, 1/16 of the envs get one object
spawned and one destroyed every 100 steps, as bench.py runs it): 2048 envs for a long time — contact / row maxima, capacity flags, resets,
non-finite state or inverse forces.   python tools/soak_c4.py [nenv] [steps]"""
import sys, os, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import mujoco_sim_amd as ms
import bench
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
args = types.SimpleNamespace(envs_per_gpu=nenv, pack=0, maxcon=0, pen_half=0.0)
w = bench.WORKLOADS["c4"](ms, args, 0, 0, None)
w.eng.set_cohorts(3)
e = w.eng
done, mx, sticky, resets = 0, [0, 0], 0, 0
for mark in sorted({200, 1000, 2000, 5000, 10000, 20000, steps}):
    if mark > steps: break
    w.step(mark - done, True); done = mark
    st = e.get_stats(); t, q, v, a = e.get_state()
    fin = bool(np.isfinite(q).all() and np.isfinite(v).all() and np.isfinite(a).all())
    finv = bool(np.isfinite(e.get_field("qfrc_inverse")).all())
    mx = [max(mx[0], int(st[:, 0].max())), max(mx[1], int(st[:, 1].max()))]
    print("step %6d: ncon mean %.1f max %d  nefc mean %.1f max %d  sweeps mean %.1f max %d  objects alive per env %.2f  capacity-flagged envs %d  reset envs %d  finite state %s inverse %s  robot base height %.3f .. %.3f" %
          (mark, st[:, 0].mean(), st[:, 0].max(), st[:, 1].mean(), st[:, 1].max(), st[:, 2].mean(), st[:, 2].max(), w.active.sum() / nenv,
           ((st[:, 3] & 3) != 0).sum(), ((st[:, 3] & 4) != 0).sum(), fin, finv, q[:, 2].min(), q[:, 2].max()), flush=True)
print("SOAK c4", nenv, "envs", done, "steps: max ncon", mx[0], "max rows", mx[1], "churns", getattr(w, "service_calls", 0))
