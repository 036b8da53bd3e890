"""ctypes harness for the fp64 test oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(_ROOT, "oracle", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")])
        L = C.CDLL(_PATH)
        vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.orc_make_data.restype = vp; L.orc_make_data.argtypes = [vp]
        L.orc_free_data.argtypes = [vp]
        L.orc_set_env_param.argtypes = [vp, C.c_int, dp]
        for f in ["orc_reset", "orc_kinematics", "orc_com_pos", "orc_crb", "orc_factor_m", "orc_collision",
                  "orc_make_constraint", "orc_project_constraint", "orc_com_vel", "orc_passive",
                  "orc_reference_constraint", "orc_energy", "orc_fwd_position", "orc_fwd_velocity",
                  "orc_fwd_acceleration", "orc_fwd_constraint", "orc_euler", "orc_controller",
                  "orc_set_odom_vels", "orc_step1", "orc_step2", "orc_forward", "orc_inverse", "orc_sensor_acc"]:
            getattr(L, f).argtypes = [vp]; getattr(L, f).restype = None
        L.orc_step.argtypes = [vp, C.c_int, C.c_int]
        L.orc_rne.argtypes = [vp, C.c_int, dp]
        L.orc_solve_m.argtypes = [vp, dp]
        L.orc_mul_m.argtypes = [vp, dp, dp]
        L.orc_field.restype = dp; L.orc_field.argtypes = [vp, C.c_char_p, ip]
        L.orc_int.restype = C.c_int; L.orc_int.argtypes = [vp, C.c_char_p]
        L.orc_int_field.restype = ip; L.orc_int_field.argtypes = [vp, C.c_char_p, ip]
        L.orc_get_contact.argtypes = [vp, C.c_int, dp, dp, dp, ip, ip]
        L.orc_box_box.restype = C.c_int; L.orc_box_box.argtypes = [dp] * 6 + [C.c_double] + [dp] * 3
        L.orc_convex_pair.restype = C.c_int
        L.orc_convex_pair.argtypes = [C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp, dp, dp, C.c_int, C.c_double, dp, dp, dp]
        L.orc_step_many.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int]
        L.orc_step_many_timed.restype = C.c_double; L.orc_step_many_timed.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_slot_mask.argtypes = [vp, C.c_uint]
        L.orc_set_pd.argtypes = [vp, dp, C.c_double, C.c_double]
        L.orc_set_pgs_row_order.argtypes = [C.c_int]
        L.orc_set_pgs_patch_order.argtypes = [C.c_int]
        _lib = L
    return _lib


class OrcData:
    """One environment of the oracle."""

    def __init__(self, model_ptr):
        self.L = lib()
        self.model = model_ptr
        self.d = self.L.orc_make_data(C.cast(model_ptr, C.c_void_p))

    def __del__(self):
        try:
            self.L.orc_free_data(self.d)
        except Exception:
            pass

    def f(self, name):
        """live numpy view of a double field."""
        n = C.c_int(0)
        p = self.L.orc_field(self.d, name.encode(), C.byref(n))
        if not p or n.value == 0:
            return np.zeros(0)
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def i(self, name):
        return self.L.orc_int(self.d, name.encode())

    def ifield(self, name):
        n = C.c_int(0)
        p = self.L.orc_int_field(self.d, name.encode(), C.byref(n))
        if not p or n.value == 0:
            return np.zeros(0, dtype=np.int32)
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def call(self, fn, *a):
        getattr(self.L, "orc_" + fn)(self.d, *a)

    def step(self, n=1, with_inverse=0):
        self.L.orc_step(self.d, n, with_inverse)

    def set_env_param(self, which, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self.L.orc_set_env_param(self.d, which, v.ctypes.data_as(C.POINTER(C.c_double)))

    def set_qpos(self, q, as_initial=True):
        self.f("qpos")[:] = q
        if as_initial:
            self.f("initial_qpos")[:] = q

    def set_pd(self, target, kp, kd):
        t = np.ascontiguousarray(target, dtype=np.float64)
        self.L.orc_set_pd(self.d, t.ctypes.data_as(C.POINTER(C.c_double)), float(kp), float(kd))

    def contacts(self):
        out = []
        for k in range(self.i("ncon")):
            dist = C.c_double(); pos = (C.c_double * 3)(); fr = (C.c_double * 9)(); g = (C.c_int * 2)(); dim = C.c_int()
            self.L.orc_get_contact(self.d, k, C.byref(dist), pos, fr, g, C.byref(dim))
            out.append(dict(dist=dist.value, pos=np.array(pos), frame=np.array(fr), geom=(g[0], g[1]), dim=dim.value))
        return out

    def rne(self, flg_acc):
        res = np.zeros(self.f("qvel").shape[0])
        self.L.orc_rne(self.d, flg_acc, res.ctypes.data_as(C.POINTER(C.c_double)))
        return res

    def solve_m(self, x):
        x = np.array(x, dtype=np.float64)
        self.L.orc_solve_m(self.d, x.ctypes.data_as(C.POINTER(C.c_double)))
        return x

    def mul_m(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64); r = np.zeros_like(v)
        self.L.orc_mul_m(self.d, r.ctypes.data_as(C.POINTER(C.c_double)), v.ctypes.data_as(C.POINTER(C.c_double)))
        return r
