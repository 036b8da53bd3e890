"""round 6 diagnostic: S24D, every env one step against the oracle; the env-steps whose qvel differs by more than 1e-3 although the contact COUNTS agree —
are their contact RECORDS the same, and do the device's other sweep forms (16-row only; the fused kernel's sweep) agree with the default forms on them?"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mujoco_sim_amd as ms
import orc
from mujoco_sim_amd.engine import EP
from test_gpu_round5 import _s24d_seeds
from test_gpu_teacher_forced import _same_contacts
nenv = 4096
m, e, tab = _s24d_seeds(list(range(nenv)))
e.set_cohorts(3); e.step(400 + int(sys.argv[1]) if len(sys.argv) > 1 else 400)
os.environ["MJH_WINDOW64"] = "0"; os.environ["MJH_WINDOW32"] = "0"
_, e16, _ = _s24d_seeds(list(range(nenv)))
del os.environ["MJH_WINDOW64"]; del os.environ["MJH_WINDOW32"]
lib = ms.capi.load(); lib.mjh_set_window_solver(0)
_, ef, _ = _s24d_seeds(list(range(nenv)), window=False)
lib.mjh_set_window_solver(1)
L = orc.lib(); L.orc_set_threads(16)
t, q, v, w = e.get_state()
for x in (e16, ef):
    x.set_state(qpos=q, qvel=v, time=t, warmstart=w)
dcs = None
e.step(1); e16.step(1); ef.step(1)
_, q1, v1, _ = e.get_state(); st = e.get_stats()
_, q16, v16, _ = e16.get_state(); _, qf, vf, _ = ef.get_state(); stf = ef.get_stats()
B = 512
ds = [orc.OrcData(m.ptr) for _ in range(B)]
arr = (C.c_void_p * B)(*[d.d for d in ds])
rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
bad = []
for b0 in range(0, nenv, B):
    for k, d in enumerate(ds):
        i = b0 + k
        for key, wh in EP.items(): d.set_env_param(wh, tab[key][i])
        d.f("qpos")[:] = q[i]; d.f("qvel")[:] = v[i]; d.f("qacc_warmstart")[:] = w[i]; d.f("qacc")[:] = w[i]; d.f("time")[0] = t[i]
    L.orc_step_many(arr, B, 1, 0)
    for k, d in enumerate(ds):
        i = b0 + k
        ev = rel(v1[i], d.f("qvel"))
        if st[i, 0] == d.i("ncon") and st[i, 1] == d.i("nefc") and ev > 1e-3:
            bad.append((i, ev, int(st[i, 1]), int(st[i, 2]), d.i("solver_iter"), rel(v16[i], d.f("qvel")), rel(vf[i], d.f("qvel")), rel(v1[i], v16[i]), d.contacts()))
print(f"{len(bad)} env-steps of {nenv} with equal counts and qvel error > 1e-3")
e2 = None
for (i, ev, rows, it, oit, ev16, evf, d16, oc) in bad[:12]:
    # contact records of the device at the state before the step: a fresh engine on that one env
    mm, ee, tt = _s24d_seeds([i]); ee.set_state(qpos=q[i:i+1], qvel=v[i:i+1], time=t[i:i+1], warmstart=w[i:i+1]); dc = ee.get_contacts(0); ee.close()
    print(f"env {i}: rows {rows} sweeps dev {it} / oracle {oit}: qvel err default forms {ev:.2e}, 16-row form only {ev16:.2e}, fused kernel {evf:.2e}; default vs 16-row {d16:.2e}; same contact records {_same_contacts(dc, oc)}")
