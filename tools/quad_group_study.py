"""Offline study: Gauss-Seidel steps per sweep on settled S24 piles for the current independent-PAIR schedule (two blocks per
step, the two 32-lane halves) against groups of up to FOUR mutually independent blocks (one per 16-lane row), both with the
two-tree-first sequence + first fit.  Prints the mean over envs and the distribution of the heavy envs (the launch lasts as long
as its slowest environment)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mujoco_sim_amd as ms
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = ms.scene("s24"); e = ms.Engine(m, nenv); e.load_s24(); e.step(400); e.synchronize()
gb = m.array("geom_bodyid")
def steps(blocks, G):
    seq = [b for b in blocks if len(b) == 2] + [b for b in blocks if len(b) < 2]
    used = [False] * len(seq); n = 0
    for i in range(len(seq)):
        if used[i]: continue
        used[i] = True; trees = set(seq[i]); cnt = 1
        for j in range(i + 1, len(seq)):
            if cnt >= G: break
            if used[j] or (set(seq[j]) & trees): continue
            used[j] = True; trees |= set(seq[j]); cnt += 1
        n += 1
    return n
rows = []
for env in range(nenv):
    c = e.get_contacts(env)
    blocks = [tuple(sorted({int(gb[g[0]]), int(gb[g[1]])} - {0})) for g in c["geom"]]
    deg = max([sum(1 for b in blocks if k in b) for k in range(1, 5)] + [0])
    rows.append((len(blocks), steps(blocks, 2), steps(blocks, 4), deg))
r = np.array(rows, dtype=float)
print("envs %d: blocks/env %.2f; steps per sweep: pairs %.2f, quads %.2f, busiest-body bound %.2f" % (nenv, r[:, 0].mean(), r[:, 1].mean(), r[:, 2].mean(), r[:, 3].mean()))
heavy = r[np.argsort(-r[:, 1])[: max(1, nenv // 50)]]
print("heaviest 2%% of envs: blocks %.1f, pairs %.1f, quads %.1f, bound %.1f;  max over envs: pairs %d quads %d" % (heavy[:, 0].mean(), heavy[:, 1].mean(), heavy[:, 2].mean(), heavy[:, 3].mean(), r[:, 1].max(), r[:, 2].max()))
