"""Debug probe: host-side cost of each call of the fused-step + read / write loop, with and without PyTorch in the process:
python tools/literal_probe_calls.py [torch]"""
import sys, time, numpy as np, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.set_device(0); x = torch.empty(1 << 20, device="cuda")
import mujoco_sim_amd as ms
m = ms.scene("s24"); e = ms.Engine(m, 4096); e.load_s24(); e.set_cohorts(3)
e.step(400); e.synchronize()
cmd = np.zeros((1, e.nv))
T = np.zeros(3); N = 300
for it in range(N + 20):
    a = time.perf_counter(); e.step(1, True)
    b = time.perf_counter(); e.get_joint_state(0, 1)
    c = time.perf_counter(); e.set_cmd(ddq=cmd, dq=None, env0=0)
    d = time.perf_counter()
    if it >= 20: T += (b - a, c - b, d - c)
e.synchronize()
print(sys.argv[1:], "per call [us]: step %.1f  get_joint_state %.1f  set_cmd %.1f   sum %.1f" % (*(T / N * 1e6), T.sum() / N * 1e6))
