"""Round 5: per-env contact / row / sweep statistics of S24 and S24D in the timed regime (what the window forms have to be sized for).
python tools/r05_hist.py [nenv]  -> gpurun_out/r05_hist_<config>[_mcNN].npz (ncon, nefc, niter, flags: [reps, nenv])"""
import sys, os, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
import mujoco_sim_amd as ms
import bench
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
for name, mc, steps in (("s24", 0, 1000), ("s24d", 0, 1000), ("s24d", 96, 1000)):
    args = types.SimpleNamespace(envs_per_gpu=nenv, pack=0, maxcon=mc, pen_half=0.0)
    w = bench.WORKLOADS[name](ms, args, 0, 0, None)
    e = w.eng
    e.step(w.settle_steps); e.synchronize()
    rec = []
    for s in range(0, steps, 50):
        e.step(50); e.synchronize()
        rec.append(e.get_stats().copy())
    r = np.stack(rec)
    tag = f"{name}" + (f"_mc{mc}" if mc else "")
    np.savez_compressed(os.path.join(out, f"r05_hist_{tag}.npz"), ncon=r[:, :, 0], nefc=r[:, :, 1], niter=r[:, :, 2], flags=r[:, :, 3] & 0xff)
    nc, nr, it, fl = r[:, :, 0], r[:, :, 1], r[:, :, 2], r[:, :, 3] & 0xff
    print(f"{tag}: window {e.window_solver()} lds {e.lds_bytes}; ncon mean {nc.mean():.1f} max {nc.max()}; rows mean {nr.mean():.1f} p50 {np.median(nr):.0f} p90 {np.quantile(nr, .9):.0f} p99 {np.quantile(nr, .99):.0f} max {nr.max()}; "
          f"sweeps mean {it.mean():.1f}, at cap {100 * (it >= 100).mean():.1f} %; flagged envs (any step) {int(((fl & 3) != 0).any(0).sum())}")
    print("   rows histogram (bins of 16):", np.bincount((nr.ravel() + 15) // 16, minlength=20))
    print("   per-env max ncon over the run: hist from 56:", np.bincount(nc.max(0), minlength=100)[56:])
    e.close()
