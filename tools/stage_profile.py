"""Per-stage shader-clock breakdown of one fused step (debug tool): python tools/stage_profile.py [nenv] [settle]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
import mujoco_sim_amd as ms
from mujoco_sim_amd import capi
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 400
m = ms.scene("s24"); e = ms.Engine(m, nenv); e.load_s24()
e.step(settle); e.synchronize()
names = ["", "load state", "FK + geoms", "COM/cdof/CRBA", "factor", "collision", "row headers", "J rows + params", "B, A_c, schedule",
         "vel stage (RNE, aref)", "controller/inverse", "smooth acc", "warmstart + AR", "PGS sweeps", "checkAcc + integrate", "store"]
out = np.zeros(16)
for rep in range(3):
    capi.load().mjh_debug_stage_cycles(e.h, 0, capi.dptr(out))
st = e.get_stats()
print("nenv", nenv, "mean ncon %.1f nefc %.1f iter %.1f" % (st[:,0].mean(), st[:,1].mean(), st[:,2].mean()))
prev = 0
for k in range(1, 16):
    if out[k] == 0: continue
    print(f"{k:2d} {names[k]:18s} +{out[k]-prev:12.0f} ticks   cum {out[k]:12.0f}")
    prev = out[k]
