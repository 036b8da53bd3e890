"""C4 as bench.py runs it: distribution of the solver sweeps per env, and how persistent the set of slow envs is (is a launch order / cohort
assignment renewed every 32 steps still right 32 steps later?).   python tools/r06_c4_sweeps.py [nenv]"""
import sys, os, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_sim_amd as ms
import bench
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
args = types.SimpleNamespace(envs_per_gpu=nenv, pack=0, maxcon=0, pen_half=0.0)
w = bench.WORKLOADS["c4"](ms, args, 0, 0, None)
w.eng.set_cohorts(3)
w.step(w.settle_steps, w.inverse)
prev = None
for rep in range(8):
    w.step(32, w.inverse)
    st = w.eng.get_stats()
    it = st[:, 2]; rows = st[:, 1]
    heavy = set(np.nonzero(it >= 32)[0].tolist())
    top = set(np.argsort(-(it * (rows + 24)))[:64].tolist())
    line = f"step {w.step_count}: sweeps mean {it.mean():.1f} q50/q90/q99/max {np.quantile(it, .5):.0f}/{np.quantile(it, .9):.0f}/{np.quantile(it, .99):.0f}/{it.max()}  envs >= 32 sweeps: {len(heavy)}  at cap: {(it >= 100).sum()}  rows of the envs >= 32: max {rows[list(heavy)].max() if heavy else 0}  rows>128: {(rows > 128).sum()}"
    if prev is not None:
        line += f"  | of the envs >= 32 sweeps now, {len(heavy & prev[0])} were so 32 steps ago; top-64 overlap {len(top & prev[1])}"
    print(line)
    prev = (heavy, top)
