"""C4 as SURVEY.md §8-d D3 states it: PR2 (nv 49, 6 joint equalities, its 37 mesh geoms as convex hulls) standing on the
reference's floor, 2048 envs, computed-torque wrapper + mj_inverse every step, and the spawn / destroy services exercised at
run time: every 100 steps 1/16 of the envs get one object slot spawned (dropped from 2 m beside the robot, as
test/test_spawn_and_destroy.py does from 5 m) and one destroyed.      python tools/c4_bench.py [nenv] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import mujoco_sim_amd as ms
from mujoco_sim_amd.tables import load_model_tables
from robot_common import robot_command

nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
m, z = load_model_tables(os.path.join(ROOT, "tests", "golden", "robot_c4_pr2_world_objects_mesh.npz"))
names = [ms.capi.load().mjh_id2name(m.ptr, 0, b).decode() for b in range(m.c.nbody)]
slots = [b for b, n in enumerate(names) if n.startswith("object_")]
e = ms.Engine(m, nenv)
e.set_controlled_dofs(z["controlled"].astype(np.int32))
for b in slots:
    e.set_slot_active(b, False)                      # the pool starts empty
active = np.zeros((nenv, len(slots)), dtype=bool)
rng = np.random.default_rng(0)


def churn():
    envs = rng.choice(nenv, nenv // 16, replace=False)
    for i in envs:
        off = np.nonzero(~active[i])[0]; on = np.nonzero(active[i])[0]
        if len(on) > 2:
            k = int(rng.choice(on)); e.set_slot_active(slots[k], False, env0=int(i), n=1); active[i, k] = False
        if len(off):
            k = int(rng.choice(off)); a = rng.uniform(-np.pi, np.pi); r = rng.uniform(0.8, 1.5)
            e.set_slot_active(slots[k], True, env0=int(i), n=1)
            e.set_body_pose(int(i), slots[k], [r * np.sin(a), r * np.cos(a), 2.0], [1, 0, 0, 0], [0, 0, 0, 0, 0, 0])
            active[i, k] = True


for k in range(1, 201):                              # settle + fill the pools a little
    if k % 25 == 0:
        churn()
    e.set_cmd(ddq=np.tile(robot_command(m, k), (nenv, 1))); e.step(1, True)
e.synchronize()
t0 = time.perf_counter(); tchurn = 0.0
for k in range(steps):
    if k % 100 == 0:
        e.synchronize(); t1 = time.perf_counter(); churn(); tchurn += time.perf_counter() - t1
    e.step(1, True)
e.synchronize(); dt = time.perf_counter() - t0
st = e.get_stats()
print("C4 pr2 + world + spawn/destroy: nv %d, nenv %d, lds %d B/env, %.3f ms/step (%.0f env-steps/s) incl. %.1f ms of service calls per churn; "
      "objects alive per env %.1f, mean ncon %.1f nefc %.1f sweeps %.1f, flagged %d"
      % (m.nv, nenv, e.lds_bytes, dt / steps * 1e3, nenv * steps / dt, tchurn / max(1, steps // 100) * 1e3, active.sum(1).mean(), st[:, 0].mean(), st[:, 1].mean(), st[:, 2].mean(), int((st[:, 3] != 0).sum())))
e.close()
