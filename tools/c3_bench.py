"""C3 as SURVEY.md §8-d D3 states it: 7 hinge joints (Panda chain, limits), fixed base, gravcomp = 1, computed-torque
wrapper on, 8192 envs, joint-space PD ddq = Kp (q* - q) - Kd qd (Kp 200, Kd 50: model/ontology/box/box.yaml:8) with
targets re-drawn every 200 steps.     python tools/c3_bench.py [nenv] [steps]
Two figures: (a) the PD law evaluated on the host every step from mjh_get_joint_state (the reference's ros_control
hand-off, all envs), (b) the device-only rate with the command held between hand-offs every 10 steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_sim_amd as ms

nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
m = ms.scene("arm7", 1)
e = ms.Engine(m, nenv)
e.set_controlled_dofs(np.ones(m.nv, dtype=np.int32))
rng = np.random.default_rng(0)
lo, hi = m.array("jnt_range").reshape(-1, 2).T
target = rng.uniform(lo, hi, size=(nenv, m.nv))
Kp, Kd = 200.0, 50.0


def pd():
    _, q, v, _ = e.get_state()
    return Kp * (target - q) - Kd * v


for mode, every in (("host PD every step", 1), ("command held 10 steps", 10)):
    e.reset()
    for k in range(200):                                  # settle towards the first targets
        e.set_cmd(ddq=pd()); e.step(1, True)
    e.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        if k % 200 == 0:
            target = rng.uniform(lo, hi, size=(nenv, m.nv))
        if k % every == 0:
            e.set_cmd(ddq=pd())
            e.step(every, True)
    e.synchronize(); dt = time.perf_counter() - t0
    _, q, v, _ = e.get_state(); st = e.get_stats()
    print("C3 arm7 (%s): nenv %d, %.3f ms/step, %.0f env-steps/s (with mj_inverse); |q - q*| mean %.3f, limit rows mean %.2f, flagged %d"
          % (mode, nenv, dt / steps * 1e3, nenv * steps / dt, np.abs(q - target).mean(), st[:, 1].mean(), int((st[:, 3] != 0).sum())))
e.close()
