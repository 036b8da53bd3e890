"""Soak test of the generic convex / mesh narrow phase: many envs of the mixed object pool (cubes, spheres, cylinders,
an ellipsoid, convex meshes) dropped in a heap and re-thrown every 500 steps.  python tools/soak_convex.py [nenv] [steps]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_sim_amd as ms
from mujoco_sim_amd import capi

nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
lib = capi.load()
D = lambda *a: (C.c_double * len(a))(*a)
rng = np.random.default_rng(0)
b = lib.mjh_builder_create()
o = capi.Option(); lib.mjh_builder_get_option(b, o); o.timestep = 0.005; lib.mjh_builder_set_option(b, o)
lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, D(2, 0.05, 0.01), 4, -1, -1, -1)
for k, (x, y) in enumerate([(-0.5, 0), (0.5, 0), (0, -0.5), (0, 0.5)]):      # a pen so that the round ones stay in the heap
    lib.mjh_builder_add_geom(b, b"wall%d" % k, 0, 6, D(0.5 if y else 0.03, 0.5 if x else 0.03, 0.15), D(x, y, 0.15), None, None, -1, -1, -1, -1)
pts = rng.normal(size=(50, 3)); pts /= np.linalg.norm(pts, axis=1)[:, None]; pts *= 0.1 * rng.uniform(0.7, 1, size=(50, 1))
pts = np.ascontiguousarray(pts)
mid = lib.mjh_builder_add_mesh(b, pts.ctypes.data_as(C.POINTER(C.c_double)), len(pts), None, 0, None)      # no faces: bounding-box inertia
pool = [(6, (0.10, 0.08, 0.06)), (2, (0.09, 0, 0)), (5, (0.08, 0.10, 0)), (4, (0.12, 0.08, 0.05)), (3, (0.05, 0.1, 0)), (5, (0.11, 0.04, 0)), (7, None), (7, None)]
for k, (gt, size) in enumerate(pool):
    bd = lib.mjh_builder_add_body(b, b"o%d" % k, 0, D(0, 0, 0.4 + 0.25 * k), None, 0.0)
    lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
    if gt == 7:
        lib.mjh_builder_add_mesh_geom(b, None, bd, mid, None, None, None, -1, -1, -1, -1)
    else:
        lib.mjh_builder_add_geom(b, None, bd, gt, D(*size), None, None, None, -1, -1, -1, -1)
lib.mjh_builder_set_capacity(b, 64, 64 * 6)
m = ms.Model(lib.mjh_builder_compile(b), lib); lib.mjh_builder_destroy(b)
e = ms.Engine(m, nenv)
nb = len(pool)
print("nv", m.nv, "npair", m.npair, "lds", e.lds_bytes)


def throw():
    q = np.zeros((nenv, m.nq)); v = np.zeros((nenv, m.nv))
    for k in range(nb):
        q[:, 7*k:7*k+3] = np.c_[rng.uniform(-0.25, 0.25, nenv), rng.uniform(-0.25, 0.25, nenv), 0.35 + 0.22 * k + rng.uniform(0, 0.05, nenv)]
        x = rng.normal(size=(nenv, 4)); q[:, 7*k+3:7*k+7] = x / np.linalg.norm(x, axis=1)[:, None]
        v[:, 6*k:6*k+6] = rng.normal(size=(nenv, 6)) * [0.5, 0.5, 0.5, 3, 3, 3]
    e.set_state(qpos=q, qvel=v, warmstart=np.zeros((nenv, m.nv)))


flag_total = np.zeros(3, dtype=int); maxcon = 0
t0 = time.perf_counter()
for s in range(0, steps, 100):
    if s % 500 == 0:
        throw()
    e.step(100)
    st = e.get_stats()
    flag_total += [(st[:, 3] & 1 != 0).sum(), (st[:, 3] & 2 != 0).sum(), (st[:, 3] & 4 != 0).sum()]
    maxcon = max(maxcon, int(st[:, 0].max()))
_, q, v, _ = e.get_state()
dt = time.perf_counter() - t0
z = q[:, 2::7]
print("steps %d x %d envs in %.1f s (%.0f env-steps/s); contact overflow %d, row overflow %d, NaN resets %d (env-checks); max ncon %d; z range [%.3f, %.3f]; finite %s; |v| max %.2f"
      % (steps, nenv, dt, nenv * steps / dt, *flag_total, maxcon, z.min(), z.max(), np.isfinite(q).all() and np.isfinite(v).all(), np.abs(v).max()))
e.close()
