"""Debug: replays a dumped S24 env-step (tools/tf_probe.py: gpurun_out/tf_outlier_*.npz) on the device under every schedule and
sweep cap, next to the oracle.  python tools/tf_replay.py file.npz"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mujoco_sim_amd as ms, orc
from mujoco_sim_amd import capi
from mujoco_sim_amd.engine import EP
z = np.load(sys.argv[1]); lib = capi.load(); L = orc.lib()
m = ms.scene("s24")
tab = {k[4:]: np.repeat(z[k][None], 2, axis=0) for k in z.files if k.startswith("tab_")}
def oracle(mode):
    L.orc_set_pgs_row_order(mode)
    d = orc.OrcData(m.ptr)
    for k, w in EP.items(): d.set_env_param(w, z["tab_" + k])
    d.set_qpos(z["qpos"]); d.call("reset"); d.f("qvel")[:] = z["qvel"]; d.f("qacc_warmstart")[:] = z["ws"]; d.f("qacc")[:] = z["ws"]
    d.step(1); L.orc_set_pgs_row_order(1)
    return d.f("qacc").copy(), d.i("solver_iter")
it0 = m.c.opt.iterations
for it in (1, 2, 5, 20, 50, 100, 200, 1000):
    m.c.opt.iterations = it
    row = [f"cap {it:5d}:"]
    for mode in (1, 2, 0):
        lib.mjh_set_pgs_row_order(mode)
        e = ms.Engine(m, 2); e.load_tables(tab)
        e.set_state(qpos=np.repeat(z["qpos"][None], 2, 0), qvel=np.repeat(z["qvel"][None], 2, 0), warmstart=np.repeat(z["ws"][None], 2, 0))
        e.step(1, False); _, q, v, w = e.get_state(); st = e.get_stats()
        ao, io = oracle(mode)
        j = int(np.abs(w[0] - ao).argmax())
        row.append(f"mode {mode}: dev[23] {w[0][23]:+.5f} orc[23] {ao[23]:+.5f} max|d| {np.abs(w[0] - ao).max():.2e} @dof{j} iters dev {st[0][2]} orc {io} |")
        e.close()
    print(" ".join(row))
m.c.opt.iterations = it0; lib.mjh_set_pgs_row_order(1)
# contact records at the dumped state: device against oracle
lib.mjh_set_pgs_row_order(1)
e = ms.Engine(m, 2); e.load_tables(tab)
e.set_state(qpos=np.repeat(z["qpos"][None], 2, 0), qvel=np.repeat(z["qvel"][None], 2, 0), warmstart=np.repeat(z["ws"][None], 2, 0))
dc = e.get_contacts(0)
d = orc.OrcData(m.ptr)
for k, w in EP.items(): d.set_env_param(w, z["tab_" + k])
d.set_qpos(z["qpos"]); d.call("reset"); d.f("qvel")[:] = z["qvel"]; d.f("qacc_warmstart")[:] = z["ws"]; d.f("qacc")[:] = z["ws"]
d.step(1)
oc = d.contacts()
for k, c in enumerate(oc):
    print(k, c["geom"], "dist dev %.7f orc %.7f" % (dc["dist"][k], c["dist"]), "dpos %.2e" % np.abs(dc["pos"][k] - c["pos"]).max(), "dframe %.2e" % np.abs(dc["frame"][k] - c["frame"]).max(), "dim", c["dim"])
    if np.abs(dc["frame"][k] - c["frame"]).max() > 1e-4: print("    dev frame", dc["frame"][k].round(5), "\n    orc frame", c["frame"].round(5))
e.forward()
for name in ("bias", "smooth"):
    try:
        x = e.get_field(name)[0]; print(name, "max |dev - orc|", np.abs(x - d.f("qfrc_" + name if name != "smooth" else "qacc_smooth")).max())
    except Exception as ex: print(name, ex)
