"""Shared test helpers: scene construction through the C ABI builder, oracle set-up."""
import ctypes as C

import numpy as np

import mujoco_sim_amd as ms
import orc
from mujoco_sim_amd import capi
from mujoco_sim_amd.engine import EP


def D(*a):
    return (C.c_double * len(a))(*a)


def set_opt(lib, b, **kw):
    o = capi.Option()
    lib.mjh_builder_get_option(b, C.byref(o))
    for k, v in kw.items():
        if k == "gravity":
            o.gravity[:] = v
        else:
            setattr(o, k, v)
    lib.mjh_builder_set_option(b, C.byref(o))


def free_body_model(lib, geom_type=2, size=(0.1, 0.1, 0.1), pos=(0, 0, 10), floor=False, **opt):
    b = lib.mjh_builder_create()
    set_opt(lib, b, **{**dict(timestep=0.005), **opt})
    if floor:
        lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, -1, -1, -1, -1)
    bd = lib.mjh_builder_add_body(b, b"obj", 0, D(*pos), None, 0.0)
    lib.mjh_builder_add_joint(b, b"free", bd, 0, None, None, None, 0, 0, 0, 0, 0)
    lib.mjh_builder_add_geom(b, b"g", bd, geom_type, D(*size), None, None, None, -1, -1, -1, -1)
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    return m


def hinge_pendulum_model(lib, damping=0.5, mass=2.0, length=1.0, inertia=0.1, **opt):
    b = lib.mjh_builder_create()
    set_opt(lib, b, **{**dict(timestep=0.005), **opt})
    bd = lib.mjh_builder_add_body(b, b"p", 0, D(0, 0, 2), None, 0.0)
    lib.mjh_builder_add_joint(b, b"h", bd, 3, D(0, 0, 0), D(0, 1, 0), None, damping, 0, 0, 0, 0)
    lib.mjh_builder_set_inertial(b, bd, mass, D(0, 0, -length), None, D(inertia, inertia, inertia))
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    return m


def two_link_model(lib, l1=1.0, l2=0.7, m1=1.5, m2=0.8, **opt):
    """planar double pendulum in the x-z plane (hinges about y), point masses at the link tips"""
    b = lib.mjh_builder_create()
    set_opt(lib, b, **{**dict(timestep=0.002), **opt})
    b1 = lib.mjh_builder_add_body(b, b"l1", 0, D(0, 0, 3), None, 0.0)
    lib.mjh_builder_add_joint(b, b"j1", b1, 3, D(0, 0, 0), D(0, 1, 0), None, 0, 0, 0, 0, 0)
    lib.mjh_builder_set_inertial(b, b1, m1, D(0, 0, -l1), None, D(1e-9, 1e-9, 1e-9))
    b2 = lib.mjh_builder_add_body(b, b"l2", b1, D(0, 0, -l1), None, 0.0)
    lib.mjh_builder_add_joint(b, b"j2", b2, 3, D(0, 0, 0), D(0, 1, 0), None, 0, 0, 0, 0, 0)
    lib.mjh_builder_set_inertial(b, b2, m2, D(0, 0, -l2), None, D(1e-9, 1e-9, 1e-9))
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    return m


def oracle_s24(model, tab, i):
    d = orc.OrcData(model.ptr)
    for k, w in EP.items():
        d.set_env_param(w, tab[k][i])
    d.set_qpos(tab["qpos"][i])
    d.call("reset")
    return d


def quat_angle(q1, q2):
    """angle between two unit quaternions (sign-insensitive)"""
    d = np.abs(np.sum(np.asarray(q1) * np.asarray(q2), axis=-1))
    return 2 * np.arccos(np.clip(d, 0, 1))


from mujoco_sim_amd.tables import load_model_tables, save_model_tables  # noqa: E402,F401  (fixtures of compiled model tables)
