"""Debug probe: C2 with device and oracle both in mj_solPGS row order, teacher-forced; prints what happens at the worst env-step."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mujoco_sim_amd as ms, orc
from mujoco_sim_amd import capi
from mujoco_sim_amd.engine import EP
m = ms.scene("boxpile", 64); m.c.maxcon = 600; m.c.maxefc = 2400
nenv = 4
lib = capi.load(); L = orc.lib()
lib.mjh_set_pgs_row_order(1); e = ms.Engine(m, nenv); lib.mjh_set_pgs_row_order(0)
tab = e.load_tables(ms.boxes_randomize(m, 0, nenv, jitter=0.01))
ds = []
for i in range(nenv):
    d = orc.OrcData(m.ptr)
    for k, wh in EP.items(): d.set_env_param(wh, tab[k][i])
    d.set_qpos(tab["qpos"][i]); d.call("reset"); d.step(200); ds.append(d)
L.orc_set_pgs_row_order(1)
for k in range(40):
    e.set_state(qpos=np.array([d.f("qpos") for d in ds]), qvel=np.array([d.f("qvel") for d in ds]),
                time=np.array([d.f("time")[0] for d in ds]), warmstart=np.array([d.f("qacc_warmstart") for d in ds]))
    e.step(1, False)
    for d in ds: d.step(1, 0)
    _, q, v, w = e.get_state(); st = e.get_stats()
    qo = np.array([d.f("qpos") for d in ds]); vo = np.array([d.f("qvel") for d in ds]); ao = np.array([d.f("qacc") for d in ds])
    ev = np.abs(v - vo).max(axis=1); ea = np.abs(w - ao).max(axis=1)
    if k >= 28 and k <= 32:
        i = 0
        j = int(np.abs(w[i] - ao[i]).argmax())
        print("step", k, "env0: dev ncon/nefc/iter", st[i, :3], "orc", ds[i].i("ncon"), ds[i].i("nefc"), ds[i].i("solver_iter"), "qvel err %.2e qacc err %.2e at dof %d (body %d) qacc dev %.4f orc %.4f" % (ev[i], ea[i], j, j // 6, w[i][j], ao[i][j]))
        con = ds[i].contacts()
        b = j // 6 + 1
        print("   contacts of that body:", [(c["geom"], round(c["dist"], 7)) for c in con if b in c["geom"]][:12])
L.orc_set_pgs_row_order(0)
