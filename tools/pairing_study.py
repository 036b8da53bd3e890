"""Offline study of the dual-block pairing on settled S24 envs: blocks per pair-step for several schedules."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mujoco_sim_amd as ms
nenv = 512
m = ms.scene("s24"); e = ms.Engine(m, nenv); e.load_s24(); e.step(400); e.synchronize()
gb = m.array("geom_bodyid")
def bodies(g1, g2):
    b = tuple(sorted({int(gb[g1]), int(gb[g2])} - {0}))
    return b
def greedy(blocks):          # current: block i, then first later unvisited independent block
    used = [False] * len(blocks); steps = 0
    for i, bi in enumerate(blocks):
        if used[i]: continue
        used[i] = True
        for j in range(i + 1, len(blocks)):
            if not used[j] and not (set(bi) & set(blocks[j])): used[j] = True; break
        steps += 1
    return steps
def two_body_first(blocks):
    order = sorted(range(len(blocks)), key=lambda i: (-len(blocks[i]), i))
    return greedy([blocks[i] for i in order])
def best_partner(blocks):    # i in order; partner = the independent unvisited block with the fewest remaining options
    n = len(blocks); used = [False] * n; steps = 0
    indep = [[j for j in range(n) if j != i and not (set(blocks[i]) & set(blocks[j]))] for i in range(n)]
    order = sorted(range(n), key=lambda i: (len(indep[i]), i))
    for i in order:
        if used[i]: continue
        used[i] = True
        cand = [j for j in indep[i] if not used[j]]
        if cand:
            j = min(cand, key=lambda j: (sum(1 for k in indep[j] if not used[k]), j)); used[j] = True
        steps += 1
    return steps
def max_matching(blocks):
    import itertools
    try:
        import scipy.sparse.csgraph as cg, scipy.sparse as sp
    except Exception:
        return None
    n = len(blocks)
    # general graph matching not in scipy; upper bound by n - ceil(n/2)
    return (n + 1) // 2
tot = dict(n=0, greedy=0, tbf=0, bp=0, ub=0)
for env in range(nenv):
    c = e.get_contacts(env)
    blocks = [bodies(g[0], g[1]) for g in c["geom"]]
    tot["n"] += len(blocks); tot["greedy"] += greedy(blocks); tot["tbf"] += two_body_first(blocks); tot["bp"] += best_partner(blocks); tot["ub"] += (len(blocks) + 1) // 2
print("blocks/env %.2f" % (tot["n"] / nenv))
for k in ("greedy", "tbf", "bp", "ub"):
    print("%-8s steps/env %.2f  blocks/step %.3f" % (k, tot[k] / nenv, tot["n"] / tot[k]))
