"""Offline study: pair-step COST (condim-4 steps are ~30 % more expensive than condim-3 ones) for the current two-tree-first
greedy order and for a variant that prefers a partner of the same kind."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mujoco_sim_amd as ms
nenv = 512
m = ms.scene("s24"); e = ms.Engine(m, nenv); e.load_s24(); e.step(400); e.synchronize()
gb = m.array("geom_bodyid"); cd = m.array("geom_condim")
COST = {3: 95.0, 4: 125.0}
def blocks_of(env):
    c = e.get_contacts(env)
    out = []
    for g in c["geom"]:
        b = tuple(sorted({int(gb[g[0]]), int(gb[g[1]])} - {0}))
        out.append((b, int(max(cd[g[0]], cd[g[1]]))))
    return out
def run(blocks, prefer_same):
    order = sorted(range(len(blocks)), key=lambda i: (-len(blocks[i][0]), i))
    seq = [blocks[i] for i in order]
    used = [False] * len(seq); steps = 0; cost = 0.0
    for i, (bi, ki) in enumerate(seq):
        if used[i]: continue
        used[i] = True
        cands = [j for j in range(i + 1, len(seq)) if not used[j] and not (set(bi) & set(seq[j][0]))]
        q = None
        if cands:
            same = [j for j in cands if seq[j][1] == ki]
            q = (same[0] if (prefer_same and same) else cands[0])
            used[q] = True
        k = max(ki, seq[q][1]) if q is not None else ki
        steps += 1; cost += COST[k]
    return steps, cost
tot = {False: [0, 0.0], True: [0, 0.0]}; nb = 0; n4 = 0
for env in range(nenv):
    bl = blocks_of(env); nb += len(bl); n4 += sum(1 for b in bl if b[1] == 4)
    for ps in (False, True):
        s, c = run(bl, ps); tot[ps][0] += s; tot[ps][1] += c
print("blocks/env %.2f (condim-4: %.2f)" % (nb / nenv, n4 / nenv))
for ps in (False, True):
    print("prefer_same=%s: steps/env %.2f cost/env %.0f" % (ps, tot[ps][0] / nenv, tot[ps][1] / nenv))
