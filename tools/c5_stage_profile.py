"""Per-stage shader-clock breakdown of C5 as bench.py runs it (pendulum + bowl of 37 mesh geoms, per-env spin, 4096 envs):
python tools/c5_stage_profile.py [steps before the stamps]"""
import sys, os, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import mujoco_sim_amd as ms
from mujoco_sim_amd import capi
ap = argparse.Namespace(envs_per_gpu=0, pack=0, maxcon=0, extra_steps=20, timing_stride=5, pen_half=0.0, cohorts=-1, steps_per_launch=0, no_gather=False)
w = bench.WORKLOADS["c5"](ms, ap, 0, 0, torch.cuda.current_stream().cuda_stream)
w.step(int(sys.argv[1]) if len(sys.argv) > 1 else 120, True); w.eng.synchronize()
e = w.eng; m = w.model
names = ["", "load state", "FK + geoms", "COM/cdof/CRBA", "factor", "collision", "row headers", "J rows + params", "B, schedule",
         "vel stage (RNE, aref)", "controller/inverse", "smooth acc", "warmstart + A_c + AR", "PGS sweeps", "checkAcc + integrate", "store"]
out = np.zeros(16)
for rep in range(2):
    capi.load().mjh_debug_stage_cycles(e.h, 1, capi.dptr(out))
st = e.get_stats()
print("c5 nv", m.nv, "nbody", m.c.nbody, "ngeom", m.c.ngeom, "lds", e.lds_bytes, "mean ncon %.2f nefc %.2f iter %.2f" % (st[:, 0].mean(), st[:, 1].mean(), st[:, 2].mean()))
prev = 0
for k in range(1, 16):
    if out[k] == 0: continue
    print(f"{k:2d} {names[k]:24s} +{out[k]-prev:10.0f} ticks   cum {out[k]:10.0f}")
    prev = out[k]
