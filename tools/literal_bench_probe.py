"""The literal loop's slow runs hold a few waits of 5-60 ms in mjh_get_joint_state (the median stays at 0.36 ms): where they fall.
python tools/literal_bench_probe.py"""
import sys, os, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
import mujoco_sim_amd as ms
ap = argparse.Namespace(envs_per_gpu=0, pack=0, maxcon=0, extra_steps=20, timing_stride=5, pen_half=0.0, cohorts=-1, steps_per_launch=0, no_gather=False)
stream = torch.cuda.current_stream().cuda_stream
tp0 = time.perf_counter()
w = bench.WORKLOADS["s24"](ms, ap, 0, 0, stream); w.eng.set_cohorts(3); w.step(int(os.environ.get("SETTLE", "400")), False); w.eng.synchronize()
e = w.eng; cmd = np.zeros((1, e.nv))
import gc
if os.environ.get("GC_OFF"): gc.collect(); gc.disable(); print("cycle collector off")
n = 3000; W = np.zeros(n); T = np.zeros(n)
e.synchronize(); t00 = time.perf_counter()
print(f"literal loop starts {t00 - tp0:.3f} s after the engine was built")
for i in range(n):
    e.step1(); e.inverse(); t1 = time.perf_counter(); e.get_joint_state(0, 1); t2 = time.perf_counter(); W[i] = t2 - t1; T[i] = t2 - t00; e.set_cmd(ddq=cmd, dq=None, env0=0); e.step2()
e.synchronize(); el = time.perf_counter() - t00
print(f"{n} steps: {w.nenv * n / el / 1e6:5.2f} M; median wait {np.median(W)*1e6:.0f} us; waits over 1 ms:")
for i in np.nonzero(W > 1e-3)[0]: print(f"   step {i:5d} at {T[i]*1e3:8.1f} ms: {W[i]*1e3:.2f} ms")
print("without them:", f"{w.nenv * n / (el - W[W > 1e-3].sum() + (W > 1e-3).sum() * np.median(W)) / 1e6:.2f} M")
