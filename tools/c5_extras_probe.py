"""C5 as the driver line's extra (bench.short_config_line) against the main path's sequence, in one process: where do 12 % go?
python tools/c5_extras_probe.py"""
import sys, os, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import mujoco_sim_amd as ms
ap = argparse.Namespace(envs_per_gpu=0, pack=0, maxcon=0, extra_steps=20, timing_stride=5, pen_half=0.0, cohorts=-1, steps_per_launch=0, no_gather=False)
stream = torch.cuda.current_stream().cuda_stream

def mainlike(settle, warm, steps, timing=5, sync="torch"):
    w = bench.WORKLOADS["c5"](ms, ap, 0, 0, stream); eng = w.eng
    pub = torch.empty(w.rows * eng.state_stride, dtype=torch.float32, device="cuda")
    def run(n):
        while n > 0:
            k = min(n, 3 - w.step_count % 3); w.step(k, w.inverse); n -= k
            if w.step_count % 3 == 0: eng.export_state_device(pub.data_ptr())
    run(settle); run(warm); torch.cuda.synchronize()
    if timing: eng.set_launch_timing(timing)
    t0 = time.perf_counter(); run(steps)
    torch.cuda.synchronize() if sync == "torch" else eng.synchronize()
    el = time.perf_counter() - t0
    eng.close()
    return w.nenv * steps / el / 1e6

for settle, warm, steps in ((100, 20, 300), (100, 20, 358), (100, 25, 300), (100, 26, 300), (100, 20, 600), (99, 21, 300)):
    print(f"main-like {settle} + {warm} + {steps}:", round(mainlike(settle, warm, steps), 2))
r = bench.short_config_line(ms, ap, "c5", 0, stream); print("extras function:", round(r["value"] / 1e6, 2), "steps", r["steps"])
