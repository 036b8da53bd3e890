"""S24 contact-mix / block-scheduling statistics (design aid): python tools/contact_mix.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mujoco_sim_amd as ms
m = ms.scene("s24"); e = ms.Engine(m, 256); e.load_s24(); e.step(500)
gb = m.array("geom_bodyid")
tot = single = 0; groups2 = groups4 = blocks = 0
for i in range(0, 256, 2):
    c = e.get_contacts(i)
    bodies = [tuple(sorted(set(int(gb[g]) for g in gg) - {0})) for gg in c["geom"]]
    tot += len(bodies); single += sum(1 for b in bodies if len(b) == 1)
    # greedy k-way grouping of independent blocks in order
    for kway in (2, 4):
        used = [False] * len(bodies); ng = 0
        for a in range(len(bodies)):
            if used[a]: continue
            used[a] = True; busy = set(bodies[a]); cnt = 1
            for b2 in range(a + 1, len(bodies)):
                if cnt >= kway: break
                if not used[b2] and not (busy & set(bodies[b2])):
                    used[b2] = True; busy |= set(bodies[b2]); cnt += 1
            ng += 1
        if kway == 2: groups2 += ng
        else: groups4 += ng
    blocks += len(bodies)
print(f"contacts {tot}, single-body {single/tot:.2%}; blocks/groups: 2-way {blocks/groups2:.2f}, 4-way {blocks/groups4:.2f}")
