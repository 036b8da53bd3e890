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
w64 = int(os.environ.get("MJH_WINDOW64", "192" if name == "s24d" else "96")); w32 = int(os.environ.get("MJH_WINDOW32", "0" if name == "s24d" else "96"))
for rep in range(3):
    e.step(1); e.synchronize()
    st = e.get_stats()
    rows, it, clk = st[:, 1], st[:, 2], st[:, 0] * 32.0
    us = clk / 2400.0       # s_memtime counts shader clocks here (2.4 GHz: a 258 us kernel reads 620 k ticks)
    if rep < 2: continue
    print(f"{name}: per-env clocks of its wavefront (s_memtime ticks, 32-tick granularity, 2.4 GHz -> us): max {us.max():.0f} us")
    def form_of(lo):          # the form an env of lo.. rows takes under the thresholds in force (engine.hip: S.win32 / S.win64; step_kernel.h: wh[4])
        if w64 and lo > w64: return "64-row"
        if w32 and lo > w32 and lo <= 128: return "32-row"
        return "16-row" if lo <= 96 else "16-row + tiers"
    edges = [(1, 32), (33, 64), (65, 80), (81, 96), (97, 112), (113, 128), (129, 144), (145, 160), (161, 176), (177, 192), (193, 208), (209, 256), (257, 320)]
    classes = [(lo, hi, form_of(lo)) for lo, hi in edges]
    simd_us = 0.0
    for lo, hi, form in classes:
        m = (rows >= lo) & (rows <= hi)
        if m.any():
            simd_us += us[m].sum() / {"16": 4, "32": 2, "64": 1}[form[:2]]
            print(f"  rows {lo:3d}-{hi:3d} ({form:15s}): {int(m.sum()):5d} envs, sweeps mean {it[m].mean():5.1f}, wave time mean {us[m].mean():7.1f} max {us[m].max():7.1f} us; at the sweep cap: mean {us[m & (it >= 100)].mean() if (m & (it >= 100)).any() else 0:7.1f} us")
    print(f"  window wavefronts hold {simd_us / 1e3:.1f} SIMD-ms per step of all cohorts: {simd_us / 1024:.0f} us of every one of the chip's 1024 SIMDs (one such wavefront per SIMD: 424 registers)")
