"""Per-stage shader-clock breakdown of config C3 (7-hinge arm, four arms per wavefront, in-engine PD, mj_inverse): python tools/c3_stage_profile.py [rows] [pack]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mujoco_sim_amd as ms
from mujoco_sim_amd import capi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 683
pack = int(sys.argv[2]) if len(sys.argv) > 2 else 4
base = ms.scene("arm7", 1); m = base.replicate(pack) if pack > 1 else base
e = ms.Engine(m, rows); e.set_controlled_dofs(np.ones(m.nv, dtype=np.int32)); e.set_pd_controller(200.0, 50.0)
lo, hi = base.array("jnt_range").reshape(-1, 2).T
e.set_pd_target(np.random.default_rng(0).uniform(lo, hi, size=(rows * pack, base.nv)).reshape(rows, -1))
e.step(200, True); e.synchronize()
names = ["", "load state", "FK + geoms", "COM/cdof/CRBA", "factor", "collision", "row headers", "J rows + params", "B, schedule",
         "vel stage (RNE, aref)", "controller/inverse", "smooth acc", "warmstart + A_c + AR", "PGS sweeps", "checkAcc + integrate", "store"]
out = np.zeros(16)
for rep in range(3):
    capi.load().mjh_debug_stage_cycles(e.h, 1, capi.dptr(out))
print("rows", rows, "pack", pack, "nv", m.nv, "lds", e.lds_bytes)
prev = 0
for k in range(1, 16):
    if out[k] == 0: continue
    print(f"{k:2d} {names[k]:24s} +{out[k]-prev:10.0f} ticks   cum {out[k]:10.0f}")
    prev = out[k]
