"""Round 5 probe: how much of the 16-row form's wavefront time is lanes waiting for the slowest of the wave's four envs?
Cost model of an env: pairs(windows) x sweeps.  Grouping by the PREVIOUS step's (windows, sweeps) — the best the launch order can know — against
the perfect grouping (this step's own) and a random one.   python tools/r05_group_waste.py [s24|s24d] [lag]"""
import sys, os, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import mujoco_sim_amd as ms
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "s24"
lag = int(sys.argv[2]) if len(sys.argv) > 2 else 1
args = types.SimpleNamespace(envs_per_gpu=4096, pack=0, maxcon=0, pen_half=0.0)
w = bench.WORKLOADS[name](ms, args, 0, 0, None)
e = w.eng
e.step(w.settle_steps + 40); e.synchronize()
def snap():
    st = e.get_stats()
    return st[:, 1].astype(np.int64), st[:, 2].astype(np.int64)
r0, i0 = snap()
e.step(lag); e.synchronize()
r1, i1 = snap()
m = (r1 <= 96) & (r0 <= 96)          # the 16-row form's envs
nw0, nw1 = (r0[m] + 15) // 16, (r1[m] + 15) // 16
it0, it1 = i0[m], i1[m]
cost1 = ((nw1 + 1) // 2) * it1      # pair-sweeps
def simd(order):
    n = len(order) // 4 * 4
    nwg = nw1[order][:n].reshape(-1, 4).max(1); itg = it1[order][:n].reshape(-1, 4).max(1)
    return float((((nwg + 1) // 2) * itg).sum())
own = float(cost1.sum()) / 4
print(f"{name}: {int(m.sum())} envs in the 16-row form; sweeps mean {it1.mean():.1f}, at the cap {float((it1 >= 100).mean()) * 100:.0f}%; corr(sweeps now, sweeps {lag} step(s) ago) = {np.corrcoef(it0, it1)[0, 1]:.3f}")
print(f"  pair-sweeps of wavefronts: own work / 4 = {own:.0f} (1.00);  perfect grouping {simd(np.lexsort((-it1, -nw1))) / own:.3f};  by the values {lag} step(s) ago {simd(np.lexsort((-it0, -nw0))) / own:.3f};"
      f"  by windows only {simd(np.argsort(-nw1, kind='stable')) / own:.3f};  random {simd(np.random.default_rng(0).permutation(len(nw1))) / own:.3f}")
for lo, hi in [(0, 30), (30, 60), (60, 99), (100, 100)]:
    k = (it0 >= lo) & (it0 <= hi)
    if k.any(): print(f"  sweeps {lag} step(s) ago in [{lo},{hi}]: {int(k.sum())} envs, now mean {it1[k].mean():.1f}, std {it1[k].std():.1f}, at the cap {float((it1[k] >= 100).mean()) * 100:.0f}%")
