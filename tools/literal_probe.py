"""Debug probe: the reference's literal loop and the fused step with a per-step read / write of env 0, S24 at 4096 envs: python tools/literal_probe.py [cohorts]"""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import mujoco_sim_amd as ms
m = ms.scene("s24"); e = ms.Engine(m, 4096); e.load_s24(); e.set_cohorts(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
e.step(400); e.synchronize()
cmd = np.zeros((1, e.nv))
def fused(n, rd=True, wr=True):
    for _ in range(n):
        e.step(1, True)
        if rd: e.get_joint_state(0, 1)
        if wr: e.set_cmd(ddq=cmd, dq=None, env0=0)
def literal(n):
    for _ in range(n):
        e.step1(); e.inverse(); e.get_joint_state(0, 1); e.set_cmd(ddq=cmd, dq=None, env0=0); e.step2()
for name, f in (("literal", literal), ("fused rd+wr", lambda n: fused(n)), ("fused rd", lambda n: fused(n, True, False)), ("fused none", lambda n: fused(n, False, False)), ("literal", literal), ("fused rd+wr", lambda n: fused(n))):
    f(10); e.synchronize(); t0 = time.perf_counter(); f(200); e.synchronize(); dt = time.perf_counter() - t0
    print(name, "%.3f ms/step  %.2f M" % (dt / 200 * 1e3, 4096 * 200 / dt / 1e6))
