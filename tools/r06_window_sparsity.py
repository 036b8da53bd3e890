"""S24 / S24D: how sparse are the windows of the window kernel over the BODIES?  A window = 16 consecutive constraint rows (4 pyramid rows per
contact); a row of J^ touches the 6 dofs of each of its (at most two) free bodies.  Prints, per env and per wavefront of four envs (launch order:
sorted by window count), the share of (window, body) slots that hold a non-zero — what a between-window product over 24 dofs multiplies by zero.
python tools/r06_window_sparsity.py [s24|s24d] [nenv] [steps]"""
import sys, os, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_sim_amd as ms
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "s24"
nenv = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
args = types.SimpleNamespace(envs_per_gpu=nenv, pack=0, maxcon=0, pen_half=0.0)
w = bench.WORKLOADS[name](ms, args, 0, 0, None)
e = w.eng; m = w.model
e.step(steps); e.synchronize()
lib = m.lib
gb = np.array([m.c.geom_bodyid[g] for g in range(m.c.ngeom)])
nb = m.c.nbody - 1
masks = []     # per env: [nwin] bit masks of bodies touched
nbod_hist = np.zeros(nb + 1, int)
for i in range(nenv):
    c = e.get_contacts(i)
    rows = []
    for g1, g2 in c["geom"]:
        b = 0
        for g in (g1, g2):
            if gb[g] > 0: b |= 1 << (gb[g] - 1)
        rows += [b] * 4
    nw = (len(rows) + 15) // 16
    mk = [0] * nw
    for r, b in enumerate(rows): mk[r // 16] |= b
    masks.append(mk)
    for x in mk: nbod_hist[bin(x).count("1")] += 1
tot = sum(len(mk) for mk in masks)
used = sum(bin(x).count("1") for mk in masks for x in mk)
print(name, "envs", nenv, "windows", tot, "mean per env %.2f" % (tot / nenv), "bodies per window histogram", nbod_hist.tolist(), "-> non-zero (window, body) slots per env: %.3f" % (used / (tot * nb)))
order = sorted(range(nenv), key=lambda i: -len(masks[i]))
tw = uw = 0
hist4 = np.zeros(nb + 1, int)
for k in range(0, nenv, 4):
    grp = [masks[i] for i in order[k:k + 4]]
    nw = max(len(g) for g in grp)
    for wi in range(nw):
        u = 0
        for g in grp:
            if wi < len(g): u |= g[wi]
        hist4[bin(u).count("1")] += 1; uw += bin(u).count("1"); tw += nb
print("  wavefronts of four (sorted by window count): bodies per window histogram", hist4.tolist(), "-> non-zero slots %.3f" % (uw / tw))
# pairs of windows (the kernel sweeps the register windows in pairs: one transpose-reduce per pair)
tp = up = 0
for k in range(0, nenv, 4):
    grp = [masks[i] for i in order[k:k + 4]]
    nw = max(len(g) for g in grp)
    for wi in range(0, nw, 2):
        u = 0
        for g in grp:
            for j in (wi, wi + 1):
                if j < len(g): u |= g[j]
        up += bin(u).count("1"); tp += nb
print("  ... per PAIR of windows: non-zero slots %.3f" % (up / tp))
