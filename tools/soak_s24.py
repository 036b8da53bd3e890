"""S24 soak (debug tool): steps 4096 envs for a long time and reports contact counts and sticky capacity flags (contacts beyond
contact_capacity, rows beyond maxefc, patches beyond the patch pool).  python tools/soak_s24.py [nenv] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mujoco_sim_amd as ms
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
m = ms.scene("s24"); e = ms.Engine(m, nenv); e.load_s24()
done = 0
for mark in sorted({400, 1000, 2000, 5000, 10000, 20000, steps}):
    if mark > steps: break
    e.step(mark - done); done = mark
    st = e.get_stats()
    print("step %6d: ncon mean %.1f max %d  nefc mean %.1f max %d  sweeps mean %.1f  capacity-flagged envs %d  reset envs %d" %
          (mark, st[:, 0].mean(), st[:, 0].max(), st[:, 1].mean(), st[:, 1].max(), st[:, 2].mean(), ((st[:, 3] & 3) != 0).sum(), ((st[:, 3] & 4) != 0).sum()), flush=True)
