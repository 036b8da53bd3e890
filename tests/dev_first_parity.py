"""Ad-hoc GPU-vs-oracle comparison used while bringing the kernel up (not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import mujoco_sim_amd as ms
import orc

np.set_printoptions(precision=5, suppress=True, linewidth=200)
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = ms.scene("s24")
e = ms.Engine(m, nenv)
print("lds bytes", e.lds_bytes, "nenv", nenv)
tab = e.load_s24()
ds = []
for i in range(nenv):
    d = orc.OrcData(m.ptr)
    for k, w in ms.engine.EP.items():
        d.set_env_param(w, tab[k][i])
    d.set_qpos(tab["qpos"][i]); d.call("reset")
    ds.append(d)

def cmp(tag, a, b):
    a = np.asarray(a); b = np.asarray(b)
    err = np.abs(a - b).max() if a.size else 0.0
    print(f"{tag:28s} maxabs {err:.3e}  ref {np.abs(b).max() if b.size else 0:.3e}")
    return err

# stage-by-stage at step 0
e.forward(); e.synchronize()
for d in ds: d.call("forward")
xp, xq = e.get_body_state()
cmp("xpos", xp.reshape(nenv, -1), [d.f("xpos") for d in ds])
cmp("xquat", xq.reshape(nenv, -1), [d.f("xquat") for d in ds])
cmp("qfrc_bias", e.get_field("qfrc_bias"), [d.f("qfrc_bias") for d in ds])
cmp("qacc_smooth", e.get_field("qacc_smooth"), [d.f("qacc_smooth") for d in ds])
cmp("qacc", e.get_field("qacc"), [d.f("qacc") for d in ds])
st = e.get_stats()
print("stats gpu", st[:4].tolist(), "orc", [(d.i("ncon"), d.i("nefc"), d.i("solver_iter")) for d in ds[:4]])
c = e.get_contacts(0); oc = ds[0].contacts()
print("contacts gpu", len(c["dist"]), "orc", len(oc))
if len(oc) == len(c["dist"]) and len(oc):
    cmp("contact dist", c["dist"], [x["dist"] for x in oc]); cmp("contact pos", c["pos"], [x["pos"] for x in oc])
    cmp("contact frame", c["frame"], [x["frame"] for x in oc])
for nsteps in [1, 9, 40, 150, 200]:
    t0 = time.time(); e.step(nsteps); e.synchronize(); tg = time.time() - t0
    for d in ds: d.step(nsteps)
    t, q, v, w = e.get_state()
    print(f"--- after +{nsteps} steps (gpu {tg*1e3:.1f} ms)")
    eq = cmp("qpos", q, [d.f("qpos") for d in ds]); cmp("qvel", v, [d.f("qvel") for d in ds])
    st = e.get_stats()
    print("stats gpu", st[:4].tolist(), "orc", [(d.i("ncon"), d.i("nefc"), d.i("solver_iter"), d.i("warn")) for d in ds[:4]])
    perenv = np.abs(q - np.array([d.f("qpos") for d in ds])).max(axis=1)
    print("per-env qpos err", perenv)
