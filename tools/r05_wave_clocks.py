"""Round 5 probe (library built with -DWN_PROF_CLK: tools/window_variant.sh prof -DWN_PROF_CLK): clocks of the window kernel's wavefront per env, by form.
MJHIP_LIB=build_exp/prof/libmjhip.so python tools/r05_wave_clocks.py [s24d|s24] [cohorts]"""
import sys, os, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
import mujoco_sim_amd as ms
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "s24d"
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 3
args = types.SimpleNamespace(envs_per_gpu=4096, pack=0, maxcon=0, pen_half=0.0)
w = bench.WORKLOADS[name](ms, args, 0, 0, None)
e = w.eng; e.set_cohorts(nc)
e.step(w.settle_steps + 40); e.synchronize()
w64 = int(os.environ.get("MJH_WINDOW64", "208")); w32 = 96
for rep in range(3):
    e.step(1); e.synchronize()
    st = e.get_stats()
    rows, it, clk = st[:, 1], st[:, 2], st[:, 0] * 32.0
    us = clk / 100.0        # s_memtime ticks at 100 MHz
    if rep < 2: continue
    print(f"{name}: per-env clocks of its wavefront (s_memtime ticks x 32 granularity; 100 MHz -> us): max {us.max():.0f} us")
    for lo, hi, form in ((1, 64, "16-row"), (65, 96, "16-row"), (97, 128, "32-row"), (129, 160, "16-row + tiers"), (161, 192, "16-row + tiers"), (193, w64, "16-row + tiers"), (w64 + 1, 256, "64-row"), (257, 320, "64-row"), (321, 400, "16-row + tiers")):
        m = (rows >= lo) & (rows <= hi)
        if m.any():
            print(f"  rows {lo:3d}-{hi:3d} ({form:15s}): {int(m.sum()):5d} envs, sweeps mean {it[m].mean():5.1f}, wave time mean {us[m].mean():7.1f} max {us[m].max():7.1f} us; at the sweep cap: mean {us[m & (it >= 100)].mean() if (m & (it >= 100)).any() else 0:7.1f} us")
