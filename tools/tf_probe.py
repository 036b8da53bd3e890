"""Debug probe: S24 teacher-forced (default engine vs the oracle in row order); prints the worst env-steps and what differs there.
python tools/tf_probe.py [nenv] [nsteps] [schedule]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mujoco_sim_amd as ms, orc
from mujoco_sim_amd import capi
from helpers import oracle_s24
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
sched = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lib = capi.load(); lib.mjh_set_pgs_row_order(sched)
m = ms.scene("s24"); e = ms.Engine(m, nenv); tab = e.load_s24()
ds = [oracle_s24(m, tab, i) for i in range(nenv)]
for d in ds: d.step(400)
rel = lambda a, b: np.abs(a - b).max(axis=1) / np.maximum(1.0, np.abs(b).max(axis=1))
worst = []
for k in range(nsteps):
    q0 = np.array([d.f("qpos") for d in ds]); v0 = np.array([d.f("qvel") for d in ds]); w0 = np.array([d.f("qacc_warmstart") for d in ds])
    e.set_state(qpos=q0, qvel=v0, time=np.array([d.f("time")[0] for d in ds]), warmstart=w0)
    dcs = [e.get_contacts(i) for i in range(nenv)]       # the contact set the step is going to use (snapshot at the state just set)
    e.step(1, False)
    for d in ds: d.step(1, 0)
    _, q, v, w = e.get_state(); st = e.get_stats()
    vo = np.array([d.f("qvel") for d in ds]); ao = np.array([d.f("qacc") for d in ds])
    ev = rel(v, vo)
    for i in range(nenv):
        agree = st[i, 0] == ds[i].i("ncon") and st[i, 1] == ds[i].i("nefc")
        if ev[i] > 2e-5 and agree:
            j = int(np.abs(w[i] - ao[i]).argmax())
            con = ds[i].contacts()
            dc = dcs[i]
            np.savez(os.path.join(ROOT, "gpurun_out", f"tf_outlier_env{i}_step{k}.npz"), qpos=q0[i], qvel=v0[i], ws=w0[i], dev_qacc=w[i], orc_qacc=ao[i], env=i,
                     **{f"tab_{kk}": np.asarray(vv[i]) for kk, vv in tab.items()})
            print(f"step {k} env {i}: qvel rel {ev[i]:.2e}; dev ncon/nefc/iter/flags {st[i]}, orc {ds[i].i('ncon')} {ds[i].i('nefc')} {ds[i].i('solver_iter')}; worst qacc dof {j}: dev {w[i][j]:.5f} orc {ao[i][j]:.5f}")
            print("   orc contacts (geoms, dist):", [(c["geom"], round(c["dist"], 6)) for c in con])
            try:
                print("   dev contacts:", [(tuple(int(x) for x in g), round(float(dd), 6)) for dd, g in zip(dc["dist"], dc["geom"])])
            except Exception as ex:
                print("   (dev contacts unavailable)", ex)
