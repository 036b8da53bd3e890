"""Throughput of BASELINE config C2 (64 free boxes per env, nv 384) on the many-body path: python tools/c2_bench.py [nenv] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mujoco_sim_amd as ms
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m = ms.scene("boxpile", 64); m.c.maxcon = 600; m.c.maxefc = 2400
e = ms.Engine(m, nenv)
rng = np.random.default_rng(1)
q0 = np.tile(m.array("qpos0"), (nenv, 1))
q0[:, 0::7] += rng.uniform(-0.01, 0.01, (nenv, 64)); q0[:, 1::7] += rng.uniform(-0.01, 0.01, (nenv, 64))
e.set_initial_qpos(q0); e.reset()
print("lds", e.lds_bytes, "B/env; settling ..."); sys.stdout.flush()
t0 = time.perf_counter(); e.step(200); e.synchronize(); print("200 settle steps: %.2f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); e.step(steps); e.synchronize(); dt = time.perf_counter() - t0
st = e.get_stats()
print("C2: nenv %d, %.1f ms/step, %.0f env-steps/s; mean ncon %.0f max %d, mean nefc %.0f, mean sweeps %.0f, flagged envs %d" %
      (nenv, dt / steps * 1e3, nenv * steps / dt, st[:, 0].mean(), st[:, 0].max(), st[:, 1].mean(), st[:, 2].mean(), int((st[:, 3] != 0).sum())))
