"""Host cost of the per-step getters / setters of the reference's loop on an IDLE engine (nothing queued): what MjHWInterface::read /
write cost beside the step itself.  python tools/getter_latency.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mujoco_sim_amd as ms
m = ms.scene("s24"); e = ms.Engine(m, 4096); e.load_s24(); e.set_cohorts(3); e.step(50); e.synchronize()
cmd = np.zeros((1, m.nv))
def t(f, n=300):
    f(); e.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("get_joint_state(0, 1)  %.1f us" % t(lambda: e.get_joint_state(0, 1)))
print("get_joint_state(0, 16) %.1f us" % t(lambda: e.get_joint_state(0, 16)))
print("set_cmd(env 0)         %.1f us" % t(lambda: e.set_cmd(ddq=cmd, dq=None, env0=0)))
print("synchronize            %.1f us" % t(lambda: e.synchronize()))
