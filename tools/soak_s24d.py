"""S24D soak (capacity 96, the window chain with its assemble-only launch): 4096 envs for a long time, mj_inverse on in every second stretch —
contact / row maxima, sticky capacity flags, resets, non-finite state or inverse forces.   python tools/soak_s24d.py [nenv] [steps]"""
import sys, os, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (bench imports it)
import mujoco_sim_amd as ms
import bench
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
args = types.SimpleNamespace(envs_per_gpu=nenv, pack=0, maxcon=0, pen_half=0.0)
w = bench.WORKLOADS["s24d"](ms, args, 0, 0, None)
e = w.eng
done, k, mx = 0, 0, [0, 0]
for mark in sorted({400, 1000, 2000, 5000, 10000, 20000, 50000, steps}):
    if mark > steps: break
    inv = bool(k & 1); k += 1
    e.step(mark - done, inv); done = mark
    st = e.get_stats(); t, q, v, a = e.get_state()
    fin = bool(np.isfinite(q).all() and np.isfinite(v).all() and np.isfinite(a).all())
    finv = bool(np.isfinite(e.get_field("qfrc_inverse")).all()) if inv else True
    mx = [max(mx[0], int(st[:, 0].max())), max(mx[1], int(st[:, 1].max()))]
    print("step %6d%s: ncon mean %.1f max %d  nefc mean %.1f max %d  sweeps mean %.1f  capacity-flagged envs %d  reset envs %d  finite state %s inverse %s" %
          (mark, " (inverse)" if inv else "", st[:, 0].mean(), st[:, 0].max(), st[:, 1].mean(), st[:, 1].max(), st[:, 2].mean(),
           ((st[:, 3] & 3) != 0).sum(), ((st[:, 3] & 4) != 0).sum(), fin, finv), flush=True)
print("SOAK s24d", nenv, "envs", done, "steps: max ncon", mx[0], "max rows", mx[1])
