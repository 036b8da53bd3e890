"""Rate of the LITERAL reference loop (mj_main.cpp:82-112) against the engine: per step  step1 -> read (mj_inverse +
joint state of env 0 to the host) -> write (command of env 0 from the host) -> step2, i.e. two launches, two small
PCIe transfers and two host synchronisations per step.  python tools/literal_loop.py [nenv] [steps] [split]
(`split`: the scene is settled through the split API as well — a host that never calls the fused mjh_step)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mujoco_sim_amd as ms
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
m = ms.scene("s24"); e = ms.Engine(m, nenv); e.load_s24(); e.set_cohorts(3)
if len(sys.argv) > 3 and sys.argv[3] == "split":
    for k in range(400): e.step1(); e.inverse(); e.step2()
else: e.step(400)
e.synchronize()
cmd = np.zeros((1, m.nv))
t0 = time.perf_counter()
for k in range(steps):
    e.step1()
    e.inverse()
    q, v, f = e.get_joint_state(0, 1)      # MjHWInterface::read of env 0 (host sync)
    e.set_cmd(ddq=cmd, dq=None, env0=0)    # MjHWInterface::write
    e.step2()
e.synchronize()
dt = time.perf_counter() - t0
print("literal loop: nenv %d, %.3f ms/step, %.0f env-steps/s (2 launches + 1 host sync per step)" % (nenv, dt / steps * 1e3, nenv * steps / dt))
t0 = time.perf_counter()
for k in range(steps):
    e.step(1, True)
    q, v, f = e.get_joint_state(0, 1)
    e.set_cmd(ddq=cmd, dq=None, env0=0)
e.synchronize()
dt = time.perf_counter() - t0
print("fused step + per-step read/write of env 0: %.3f ms/step, %.0f env-steps/s" % (dt / steps * 1e3, nenv * steps / dt))
