import os, sys, numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mujoco_sim_amd as ms
import orc
from helpers import load_model_tables
from test_robot_fixtures import robot_command
from mujoco_sim_amd import capi
lib = capi.load()
for name in sys.argv[1:]:
  for layout in (1, 2):
    lib.mjh_set_layout_policy(layout)
    m, z = load_model_tables(os.path.join(ROOT, "tests", "golden", f"robot_{name}.npz"))
    KEEP = [int(k) for k in z["keep"]]
    nenv = 4
    e = ms.Engine(m, nenv); e.set_controlled_dofs(z["controlled"].astype(np.int32))
    d = orc.OrcData(m.ptr); d.ifield("controlled")[:] = z["controlled"]
    d.f("qvel")[:] = z["qvel0"]; e.set_state(qvel=np.tile(z["qvel0"], (nenv, 1)))
    out = []
    for k in range(1, KEEP[-1] + 1):
        cmd = robot_command(m, k)
        e.set_cmd(ddq=np.tile(cmd, (nenv, 1))); e.step(1, True)
        d.f("ddq")[:] = cmd; d.step(1, 1)
        if k in KEEP:
            t, q, v, w = e.get_state()
            out.append((k, float(np.abs(q[0] - z[f"qpos_{k}"]).max()), float(np.abs(v[0] - z[f"qvel_{k}"]).max())))
            e.set_state(qpos=np.tile(d.f("qpos"), (nenv, 1)), qvel=np.tile(d.f("qvel"), (nenv, 1)), warmstart=np.tile(d.f("qacc_warmstart"), (nenv, 1)))
    print(name, "layout", layout, " ".join(f"k{k}: {a:.1e}/{b:.1e}" for k, a, b in out))
lib.mjh_set_layout_policy(0)
