"""Contact-count statistics of S24 over a rollout (capacity planning): python tools/ncon_hist.py [nenv] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mujoco_sim_amd as ms
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1400
m = ms.scene("s24"); e = ms.Engine(m, nenv); e.load_s24()
print("capacity maxcon", m.maxcon, "maxefc", m.maxefc, "lds", e.lds_bytes)
gmax = np.zeros(nenv, dtype=int); gmaxr = np.zeros(nenv, dtype=int)
for s in range(0, steps, 25):
    e.step(25); st = e.get_stats()
    gmax = np.maximum(gmax, st[:, 0]); gmaxr = np.maximum(gmaxr, st[:, 1])
    if s % 100 == 75:
        print(f"step {s+25:5d}: ncon mean {st[:,0].mean():5.1f} p99 {np.quantile(st[:,0],0.99):4.0f} max {st[:,0].max():3d} | nefc mean {st[:,1].mean():6.1f} max {st[:,1].max():4d} | iter mean {st[:,2].mean():5.1f} | flags {np.bincount(st[:,3], minlength=8)[:8]}")
print("per-env max over rollout (sampled every 25 steps): ncon hist", np.bincount(gmax)[20:], "starting at 20")
print("max nefc", gmaxr.max(), " envs with ncon>32:", (gmax > 32).sum(), " >30:", (gmax > 30).sum(), " >28:", (gmax>28).sum())
