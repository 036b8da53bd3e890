"""Per-stage shader-clock breakdown of config C4 as bench.py runs it (PR2 + world, object pool EMPTY): python tools/c4_stage_profile.py [nenv]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import mujoco_sim_amd as ms
from mujoco_sim_amd import capi
from mujoco_sim_amd.tables import load_model_tables
from robot_common import robot_command
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 683
m, z = load_model_tables(os.path.join(ROOT, "tests", "golden", "robot_c4_pr2_world_objects_mesh.npz"))
e = ms.Engine(m, nenv); e.set_controlled_dofs(z["controlled"].astype(np.int32))
lib = capi.load()
for b in range(m.c.nbody):
    if lib.mjh_id2name(m.ptr, 0, b).decode().startswith("object_"):
        e.set_slot_active(b, False)
for k in range(1, 120):
    if k % 10 == 1:
        e.set_cmd(ddq=np.tile(robot_command(m, k), (nenv, 1)))
    e.step(1, True)
names = ["", "load state", "FK + geoms", "COM/cdof/CRBA", "factor", "collision", "row headers", "J rows + params", "B, schedule",
         "vel stage (RNE, aref)", "controller/inverse", "smooth acc", "warmstart + A_c + AR", "PGS sweeps", "checkAcc + integrate", "store"]
def show(mode, title):
    out = np.zeros(16)
    for rep in range(2):
        lib.mjh_debug_stage_cycles(e.h, mode, capi.dptr(out))
    print(title)
    prev = 0
    for k in range(1, 16):
        if out[k] == 0: continue
        print(f"{k:2d} {names[k]:24s} +{out[k]-prev:10.0f} ticks   cum {out[k]:10.0f}")
        prev = out[k]

st = e.get_stats()
print("nenv", nenv, "nv", m.nv, "lds", e.lds_bytes, "mean ncon %.1f nefc %.1f iter %.1f" % (st[:, 0].mean(), st[:, 1].mean(), st[:, 2].mean()))
show(1, "fused kernel (one launch)")
show(1 | 2, "launch chain: assemble launch (dense cohort)")
show(1 | 4, "launch chain: integrate launch")
