"""CPU study (oracle only): how much the Gauss-Seidel visiting order matters on settled S24 piles at the default 100-sweep cap.
Each env is settled for 400 steps, then the SAME state is stepped 150 more steps under each order: the device's contact-patch
order (patch_pgs.h), the independent-pair order of the block sweeps, and plain constraint-row order (mj_solPGS).
python tools/order_study.py [nenv]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mujoco_sim_amd as ms
import orc
from helpers import oracle_s24
from test_oracle_pinning import _clone
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
L = orc.lib(); m = ms.scene("s24"); tab = m.s24_randomize(0, N)
def setorder(o):   # "patch" | "pair" | "row"
    L.orc_set_pgs_row_order(1 if o == "row" else 0); L.orc_set_pgs_patch_order(1 if o == "patch" else 0)
res = {("patch", "row"): [], ("pair", "row"): [], ("patch", "pair"): []}
its = {"patch": [], "pair": [], "row": []}
try:
    for i in range(N):
        setorder("patch")
        s = oracle_s24(m, tab, i); s.step(400)
        c = {o: _clone(m, tab, i, s) for o in ("patch", "pair", "row")}
        acc1 = {}
        for k in range(150):
            for o in c:
                setorder(o); c[o].step(1)
                if k == 0: acc1[o] = c[o].f("qacc").copy(); its[o].append(c[o].i("solver_iter"))
        for (a, b) in res:
            res[(a, b)].append((float(np.abs(acc1[a] - acc1[b]).max()), float(np.abs(c[a].f("qpos") - c[b].f("qpos")).max())))
finally:
    L.orc_set_pgs_row_order(0); L.orc_set_pgs_patch_order(-1)
for k, v in res.items():
    v = np.array(v)
    print("%-5s vs %-4s: 1-step |d qacc| max %.3e median %.3e ; 150-step |d qpos| max %.3e median %.3e" % (k[0], k[1], v[:, 0].max(), np.median(v[:, 0]), v[:, 1].max(), np.median(v[:, 1])))
print("sweeps at the first step:", {o: (float(np.mean(x)), int(np.sum(np.array(x) >= 100))) for o, x in its.items()}, "(mean, envs at the cap) of", N)
