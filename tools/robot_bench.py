"""Throughput of the robot configs (table fixtures of the reference's assets): python tools/robot_bench.py <name> [nenv] [steps]
   name: pr2_world | pr2_world_mesh | hsrb4s_world | ridgeback_panda | tiago | c5_pendulum_bowl_mesh ...
   (C3: ridgeback_panda at 8192 envs, C4: pr2_world_mesh at 2048, C5: c5_pendulum_bowl_mesh at 4096 per GPU)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import mujoco_sim_amd as ms
from mujoco_sim_amd.tables import load_model_tables
from robot_common import robot_command
name = sys.argv[1]; nenv = int(sys.argv[2]) if len(sys.argv) > 2 else 2048; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
m, z = load_model_tables(os.path.join(ROOT, "tests", "golden", f"robot_{name}.npz"))
e = ms.Engine(m, nenv); e.set_controlled_dofs(z["controlled"].astype(np.int32))
if "qvel0" in z and np.any(z["qvel0"]):          # C5 (pendulum world + bowl): per-env spin so that the envs differ
    rng = np.random.default_rng(0)
    e.set_state(qvel=z["qvel0"][None, :] * rng.uniform(0.5, 1.5, size=(nenv, 1)))
for k in range(1, 101):                       # settle under the commanded accelerations (one command upload per step, all envs)
    e.set_cmd(ddq=np.tile(robot_command(m, k), (nenv, 1))); e.step(1, True)
e.synchronize()
cmd = np.tile(robot_command(m, 101), (nenv, 1))
t0 = time.perf_counter()
for k in range(steps):
    e.step(1, True)                           # mj_inverse every step, as MjHWInterface::read does
e.synchronize(); dt = time.perf_counter() - t0
st = e.get_stats()
print("%s: nv %d, nenv %d, lds %d B/env, %.3f ms/step, %.0f env-steps/s (with mj_inverse); mean ncon %.1f nefc %.1f sweeps %.1f, flagged %d" %
      (name, m.nv, nenv, e.lds_bytes, dt / steps * 1e3, nenv * steps / dt, st[:, 0].mean(), st[:, 1].mean(), st[:, 2].mean(), int((st[:, 3] != 0).sum())))
