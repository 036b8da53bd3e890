import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch
torch.cuda.set_device(0)
import mujoco_sim_amd as ms
own = torch.cuda.Stream() if "own" in (sys.argv[1] if len(sys.argv) > 1 else "") else None
st = own.cuda_stream if own is not None else torch.cuda.current_stream().cuda_stream
print("torch stream handle", st)
m = ms.scene("s24"); e = ms.Engine(m, 4096, device=0, stream=st); e.load_s24(); e.set_cohorts(3)
pub = torch.empty(4096 * e.state_stride, dtype=torch.float32, device="cuda")
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
mode = sys.argv[1] if len(sys.argv) > 1 else ""
if "timing" in mode:
    e.set_launch_timing(5); fused(50, False, False); e.synchronize(); print(e.get_launch_timing()); e.set_launch_timing(False)
if "export" in mode:
    for _ in range(20): e.step(1, True); e.export_state_device(pub.data_ptr())
    e.synchronize()
for name, f in (("literal", literal), ("fused rd+wr", lambda n: fused(n)), ("fused none", lambda n: fused(n, False, False))):
    f(10); e.synchronize(); t0 = time.perf_counter(); f(200); e.synchronize(); dt = time.perf_counter() - t0
    print(name, "%.3f ms/step  %.2f M" % (dt / 200 * 1e3, 4096 * 200 / dt / 1e6))
