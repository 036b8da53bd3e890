"""CPU study (oracle only): how many wave-steps a Gauss-Seidel sweep needs under (a) the legacy reordering schedules (patch order /
first-fit groups) and (b) a precedence-preserving list schedule of mj_solPGS's own constraint order (blocks that share no body
commute; a block may start once every EARLIER block sharing a body with it is done).  Also prints the critical path of the conflict
DAG = the bound for (b) at unlimited width.
python tools/dag_schedule_study.py s24|s24d|c2 [nenv]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mujoco_sim_amd as ms
import orc
from mujoco_sim_amd.engine import EP

which = sys.argv[1] if len(sys.argv) > 1 else "s24"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8


def blocks_of(d, m):
    """contact blocks in constraint order: (bodyA, bodyB or -1, rows)"""
    gb = m.geom_bodyid() if callable(getattr(m, "geom_bodyid", None)) else None
    out = []
    for c in d.contacts():
        if c["dist"] >= c.get("margin", 0.0) and False:
            continue
        g1, g2 = c["geom"]
        out.append((g1, g2, 1 if c["dim"] == 1 else 2 * (c["dim"] - 1)))
    return out


def list_schedule(items, cap):
    """items: (a, b) bodies (b = -1: none) in order; returns step of each item, nsteps, critical path"""
    last = {}; cnt = []; steps = []; depth = {}; crit = 0
    for (a, b) in items:
        e = max(last.get(a, 0), last.get(b, 0) if b >= 0 else 0)
        dp = 1 + max(depth.get(a, 0), depth.get(b, 0) if b >= 0 else 0)
        depth[a] = dp
        if b >= 0: depth[b] = dp
        crit = max(crit, dp)
        s = e
        while s < len(cnt) and cnt[s] >= cap: s += 1
        if s == len(cnt): cnt.append(0)
        cnt[s] += 1; steps.append(s)
        last[a] = s + 1
        if b >= 0: last[b] = s + 1
    return steps, len(cnt), crit


def first_fit(items, cap):
    """legacy: two-body items first, then first fit with reordering"""
    seq = [i for i, (a, b) in enumerate(items) if b >= 0] + [i for i, (a, b) in enumerate(items) if b < 0]
    used = [False] * len(items); n = 0
    for ii, i in enumerate(seq):
        if used[i]: continue
        used[i] = True; g = {items[i][0], items[i][1]} - {-1}; c = 1
        for j in seq[ii + 1:]:
            if c >= cap: break
            if used[j]: continue
            s = {items[j][0], items[j][1]} - {-1}
            if g & s: continue
            used[j] = True; g |= s; c += 1
        n += 1
    return n


if which in ("s24", "s24d"):
    m = ms.scene("s24") if which == "s24" else ms.scene("s24d")
    tab = m.s24_randomize(0, N)
    settle = 400
else:
    m = ms.scene("boxpile", 64); m.c.maxcon = 600; m.c.maxefc = 2400
    tab = ms.boxes_randomize(m, 0, N, jitter=0.01)
    settle = 200
import ctypes as C
ngeom = m.c.ngeom
geom_body = np.ctypeslib.as_array(m.c.geom_bodyid, shape=(ngeom,)).copy()
rows = []
for i in range(N):
    d = orc.OrcData(m.ptr)
    for k, w in EP.items():
        d.set_env_param(w, tab[k][i])
    d.set_qpos(tab["qpos"][i]); d.call("reset"); d.step(settle)
    con = d.contacts()
    efc_id = d.ifield("efc_id"); nefc = d.i("nefc")
    active = sorted(set(int(x) for x in efc_id[:nefc]))
    blk = []
    for k in active:
        c = con[k]; b1, b2 = int(geom_body[c["geom"][0]]), int(geom_body[c["geom"][1]])
        a, b = (b1, b2) if b1 > 0 and b2 > 0 else (max(b1, b2), -1)
        if b >= 0 and a > b: a, b = b, a
        blk.append((a, b, 1 if c["dim"] == 1 else 2 * (c["dim"] - 1)))
    if which == "c2":
        it = [(a, b) for a, b, n in blk]
        _, ns, crit = list_schedule(it, 16)
        _, ns4, _ = list_schedule(it, 4)
        rows.append((len(blk), first_fit(it, 16), ns, crit, ns4))
    else:
        # patches in row order: maximal runs of one body pair with <= 16 rows
        pat = []; cur = None; r = 0
        for a, b, n in blk:
            if cur != (a, b) or r + n > 16: pat.append((a, b)); cur = (a, b); r = 0
            r += n
        # legacy patch order: sorted by (single-body last, pair, order)
        sb = sorted(range(len(blk)), key=lambda i: (0 if blk[i][1] >= 0 else 1, blk[i][0], blk[i][1] if blk[i][1] >= 0 else 63, i))
        pat2 = []; cur = None; r = 0
        for i in sb:
            a, b, n = blk[i]
            if cur != (a, b) or r + n > 16: pat2.append((a, b)); cur = (a, b); r = 0
            r += n
        used = [False] * len(pat2); legacy = 0
        for i in range(len(pat2)):
            if used[i]: continue
            used[i] = True; g = {pat2[i][0], pat2[i][1]} - {-1}; c = 1
            for j in range(i + 1, len(pat2)):
                if c >= 4: break
                if used[j]: continue
                s = {pat2[j][0], pat2[j][1]} - {-1}
                if g & s: continue
                used[j] = True; g |= s; c += 1
            legacy += 1
        _, ns, crit = list_schedule(pat, 4)
        rows.append((len(blk), len(pat2), legacy, len(pat), ns, crit))
rows = np.array(rows)
if which == "c2":
    print("blocks, legacy groups(16), row-order list schedule(16) steps, critical path, list schedule(4) steps")
else:
    print("blocks, legacy patches, legacy steps, row-order patches, row-order list-schedule steps, critical path")
print(rows)
print("mean", rows.mean(axis=0))
