"""Per-stage shader-clock breakdown for config C2 (64 boxes): python tools/c2_stage_profile.py [nenv]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mujoco_sim_amd as ms
from mujoco_sim_amd import capi
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = ms.scene("boxpile", 64); m.c.maxcon = 600; m.c.maxefc = 2400
e = ms.Engine(m, nenv)
e.load_tables(ms.boxes_randomize(m, 0, nenv, jitter=0.01))      # the D3-exact pile (random sizes / orientations), as bench.py --config c2
e.step(200); e.synchronize()
names = ["", "load state", "FK + geoms", "COM/cdof/CRBA", "factor", "collision", "row headers", "J rows + params", "B, schedule",
         "vel stage (RNE, aref)", "controller/inverse", "smooth acc", "warmstart + A_c + AR", "PGS sweeps", "checkAcc + integrate", "store"]
def show(mode, title):
    out = np.zeros(16)
    for rep in range(2):
        capi.load().mjh_debug_stage_cycles(e.h, mode, capi.dptr(out))
    print(title)
    prev = 0
    for k in range(1, 16):
        if out[k] == 0: continue
        print(f"{k:2d} {names[k]:24s} +{out[k]-prev:12.0f} ticks   cum {out[k]:12.0f}")
        prev = out[k]

st = e.get_stats()
print("nenv", nenv, "mean ncon %.1f nefc %.1f iter %.1f" % (st[:,0].mean(), st[:,1].mean(), st[:,2].mean()))
show(0, "fused kernel (one launch)")
show(2, "launch chain: assemble launch")
show(4, "launch chain: integrate launch")
