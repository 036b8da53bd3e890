"""Compiled model tables <-> .npz: every field of struct mjh_model (include/mjhip.h) in one file, so that a model compiled once
(e.g. from a reference data file by the MJCF loader) can be rebuilt where the file is not available.  Plumbing for tests,
fixtures and the bench tools; no physics."""
import ctypes as C

import numpy as np

from . import capi
from .engine import Model

_OPT_FIELDS = ["timestep", "iterations", "tolerance", "impratio", "noslip_iterations", "disableflags", "noslip_tolerance"]


def save_model_tables(m, path, **extra):
    d = {"int__" + k: np.int64(getattr(m.c, k)) for k in capi._INT_SIZES + capi._INT_SIZES2}
    d["meaninertia"] = np.float64(m.c.meaninertia)
    for k in _OPT_FIELDS:
        d["opt__" + k] = np.float64(getattr(m.c.opt, k))
    d["opt__gravity"] = np.array(list(m.c.opt.gravity), dtype=np.float64)
    for n, t, _ in capi._ARRAYS + capi._ARRAYS2 + capi._ARRAYS3:
        d["arr__" + n] = m.array(n)
    for kind, objtype, n in (("body", 0, m.c.nbody), ("jnt", 1, m.c.njnt), ("geom", 2, m.c.ngeom), ("site", 3, m.c.nsite), ("sensor", 4, m.c.nsensor)):
        d["names__" + kind] = np.array([(m.lib.mjh_id2name(m.ptr, objtype, i) or b"").decode() for i in range(n)], dtype=str)
    d.update(extra)
    np.savez_compressed(path, **d)


def load_model_tables(path):
    """-> (ms.Model over a ctypes mjh_model whose arrays are numpy buffers kept alive by the object, npz)"""
    z = np.load(path)
    st = capi.Model()
    for k in capi._INT_SIZES:
        setattr(st, k, int(z["int__" + k]))
    for k in capi._INT_SIZES2:                      # fixtures written before sites / sensors / mocap bodies existed: none
        setattr(st, k, int(z["int__" + k]) if "int__" + k in z else 0)
    st.meaninertia = float(z["meaninertia"])
    for k in _OPT_FIELDS:
        if "opt__" + k not in z:       # fixtures written before the field existed keep the struct's default
            continue
        v = z["opt__" + k]
        setattr(st.opt, k, int(v) if k in ("iterations", "noslip_iterations", "disableflags") else float(v))
    for i in range(3):
        st.opt.gravity[i] = float(z["opt__gravity"][i])
    keep = []
    for n, t, _ in capi._ARRAYS + capi._ARRAYS2 + capi._ARRAYS3:
        if "arr__" + n not in z:
            if n != "body_mocapid":
                continue                             # (the struct's pointer stays NULL with a zero count)
            src = -np.ones(st.nbody, dtype=np.int32)
        else:
            src = z["arr__" + n]
        a = np.ascontiguousarray(src, dtype=np.int32 if t == "i" else np.float64)
        if a.size == 0:
            a = np.zeros(1, dtype=a.dtype)
        keep.append(a)
        setattr(st, n, a.ctypes.data_as(capi.c_int_p if t == "i" else capi.c_double_p))
    for kind in ("body", "jnt", "geom", "site", "sensor"):          # name tables (mjh_name2id / mjh_id2name), if the file has them
        if "names__" + kind in z:
            bufs = [C.create_string_buffer(str(x).encode()) for x in z["names__" + kind]]
            arr = (C.c_char_p * max(1, len(bufs)))(*[C.cast(b, C.c_char_p) for b in bufs])
            keep.append((bufs, arr))
            setattr(st, kind + "_names", C.cast(arr, C.POINTER(C.c_char_p)))
    m = Model(C.pointer(st))
    m._keep = (st, keep)
    m.note = "rebuilt from compiled tables"
    return m, z
