"""Sub-wave packing of small models (mjh_model_replicate): env-steps/s against the number of environments per wavefront.
    python tools/pack_bench.py [total_envs] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_sim_amd as ms
from mujoco_sim_amd.tables import load_model_tables

total = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
c5, z5 = load_model_tables(os.path.join(ROOT, "tests", "golden", "robot_c5_pendulum_bowl_mesh.npz"))
cases = [("C3 arm7 (7 hinges, limits, computed torque + mj_inverse)", ms.scene("arm7", 1), (1, 2, 4, 8), None),
         ("C1 pendulum (3 ball joints)", ms.scene("pendulum"), (1, 2, 3, 6), np.tile([0.3, 0.0, 0.0], 3)),
         ("C5 pendulum + bowl (37 static mesh geoms)", c5, (1, 2, 3, 6), z5["qvel0"])]
rng = np.random.default_rng(0)
for name, m, Gs, v0 in cases:
    for G in Gs:
        r = m.replicate(G) if G > 1 else m
        try:
            e = ms.Engine(r, total // G)
        except Exception as ex:
            print("%s: x%d per wave: %s" % (name, G, str(ex)[:80])); continue
        if "arm7" in name:
            e.set_controlled_dofs(np.ones(r.nv, dtype=np.int32))
            lo, hi = r.array("jnt_range").reshape(-1, 2).T
            e.set_cmd(ddq=rng.uniform(-1, 1, size=(total // G, r.nv)))
        if v0 is not None:
            e.set_state(qvel=np.tile(v0, (total // G, G)) * rng.uniform(0.5, 1.5, size=(total // G, 1)))
        e.step(100, True); e.synchronize()
        t0 = time.perf_counter()
        for k in range(steps // 10):
            e.step(10, True)
        e.synchronize(); dt = time.perf_counter() - t0
        st = e.get_stats()
        n = (steps // 10) * 10
        print("%s: x%d per wave (nv %d, lds %d B): %.3f ms/step, %.1f M env-steps/s, flagged %d" %
              (name, G, r.nv, e.lds_bytes, dt / n * 1e3, (total // G) * G * n / dt / 1e6, int((st[:, 3] != 0).sum())))
        e.close()
