"""Thin Python mirror of the C ABI: a `Model` (compiled scene) and an `Engine`
(nenv environments on one HIP device).  Names follow the reference's step loop
(src/mj_main.cpp:76-113): step1 / inverse / step2 / forward, set_cmd == MjHWInterface::write,
get_joint_state == MjHWInterface::read.  All compute happens in libmjhip.so."""
import ctypes as C

import numpy as np

from . import capi


class MjhError(RuntimeError):
    pass


def _chk(lib, rc, what):
    if rc is not None and rc < 0:
        raise MjhError(f"{what} failed ({rc}): {lib.mjh_last_error().decode()}")
    return rc


class Model:
    """Owns an mjh_model*."""

    def __init__(self, ptr, lib=None):
        self.lib = lib or capi.load()
        if not ptr:
            raise MjhError("model build failed: " + self.lib.mjh_last_error().decode())
        self.ptr = ptr
        self.c = ptr.contents

    def __getattr__(self, name):
        c = self.__dict__.get("c")
        if c is not None and name in capi._INT_SIZES + capi._INT_SIZES2 + ["meaninertia", "opt"]:
            return getattr(c, name)
        raise AttributeError(name)

    def array(self, name):
        return self.c.array(name)

    def name2id(self, objtype, name):
        return self.lib.mjh_name2id(self.ptr, objtype, name.encode())

    def replicate(self, copies):
        """sub-wave packing: `copies` instances of the moving bodies in one model (mjh_model_replicate)"""
        return Model(self.lib.mjh_model_replicate(self.ptr, int(copies)), self.lib)

    def s24_randomize(self, env0, nenv, seed_base=0x5EED0000):
        """Per-env S24 tables: dict of float64 arrays."""
        c = self.c
        out = dict(
            qpos=np.zeros((nenv, c.nq)), geom_size=np.zeros((nenv, 3 * c.ngeom)), geom_rbound=np.zeros((nenv, c.ngeom)),
            body_mass=np.zeros((nenv, c.nbody)), body_inertia=np.zeros((nenv, 3 * c.nbody)),
            body_invweight0=np.zeros((nenv, 2 * c.nbody)), dof_invweight0=np.zeros((nenv, c.nv)))
        rc = self.lib.mjh_scene_s24_randomize(self.ptr, env0, nenv, seed_base, *[capi.dptr(out[k]) for k in
                                              ["qpos", "geom_size", "geom_rbound", "body_mass", "body_inertia",
                                               "body_invweight0", "dof_invweight0"]])
        _chk(self.lib, rc, "mjh_scene_s24_randomize")
        return out


def boxes_randomize(model, env0, nenv, seed_base=0x5EED0000, jitter=0.01):
    """Per-env tables of a free-box scene (C2): dict of float64 arrays like Model.s24_randomize."""
    c = model.c
    out = dict(
        qpos=np.zeros((nenv, c.nq)), geom_size=np.zeros((nenv, 3 * c.ngeom)), geom_rbound=np.zeros((nenv, c.ngeom)),
        body_mass=np.zeros((nenv, c.nbody)), body_inertia=np.zeros((nenv, 3 * c.nbody)),
        body_invweight0=np.zeros((nenv, 2 * c.nbody)), dof_invweight0=np.zeros((nenv, c.nv)))
    rc = model.lib.mjh_scene_boxes_randomize(model.ptr, env0, nenv, seed_base, float(jitter), *[capi.dptr(out[k]) for k in
                                             ["qpos", "geom_size", "geom_rbound", "body_mass", "body_inertia",
                                              "body_invweight0", "dof_invweight0"]])
    _chk(model.lib, rc, "mjh_scene_boxes_randomize")
    return out


def load_mjcf(xml=None, path=None, paths=None):
    """MJCF-subset loader (mj_loadXML boundary, mj_util.h:185-193); `paths`: world file + robot files composed into one model"""
    lib = capi.load()
    if paths:
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        ptr = lib.mjh_load_mjcf_files(arr, len(paths))
    else:
        ptr = lib.mjh_load_mjcf_file(path.encode()) if path else lib.mjh_load_mjcf_string(xml.encode())
    m = Model(ptr, lib)
    m.note = lib.mjh_load_note().decode()
    return m


def scene(name, *args):
    lib = capi.load()
    fn = {"s24": lib.mjh_scene_s24, "s24pen": lib.mjh_scene_s24_pen, "pendulum": lib.mjh_scene_pendulum, "arm7": lib.mjh_scene_arm7,
          "boxpile": lib.mjh_scene_boxpile}[name]
    return Model(fn(*args), lib)


EP = dict(geom_size=0, geom_rbound=1, body_mass=2, body_inertia=3, body_invweight0=4, dof_invweight0=5)


class Engine:
    def __init__(self, model, nenv, device=0, stream=None):
        self.lib = model.lib
        self.model = model
        self.nenv = nenv
        h = C.c_void_p()
        _chk(self.lib, self.lib.mjh_create(model.ptr, nenv, device, C.c_void_p(stream or 0), C.byref(h)), "mjh_create")
        self.h = h
        self.nq, self.nv, self.nbody, self.ngeom = model.nq, model.nv, model.nbody, model.ngeom

    @classmethod
    def from_handle(cls, model, handle, nenv):
        """view of an engine owned by someone else (a device of an mjh_group): close() does not destroy it"""
        self = cls.__new__(cls)
        self.lib = model.lib; self.model = model; self.nenv = nenv; self.h = C.c_void_p(handle); self._borrowed = True
        self.nq, self.nv, self.nbody, self.ngeom = model.nq, model.nv, model.nbody, model.ngeom
        return self

    def close(self):
        if getattr(self, "h", None):
            if not getattr(self, "_borrowed", False):
                self.lib.mjh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stepping (mj_main.cpp:83-110)
    def step1(self): _chk(self.lib, self.lib.mjh_step1(self.h), "mjh_step1")
    def step2(self): _chk(self.lib, self.lib.mjh_step2(self.h), "mjh_step2")
    def inverse(self): _chk(self.lib, self.lib.mjh_inverse(self.h), "mjh_inverse")
    def forward(self): _chk(self.lib, self.lib.mjh_forward(self.h), "mjh_forward")
    def step(self, n=1, with_inverse=False): _chk(self.lib, self.lib.mjh_step(self.h, n, int(with_inverse)), "mjh_step")
    def synchronize(self): _chk(self.lib, self.lib.mjh_synchronize(self.h), "mjh_synchronize")

    def set_timestep(self, dt): _chk(self.lib, self.lib.mjh_set_timestep(self.h, float(dt)), "mjh_set_timestep")
    @property
    def timestep(self): return self.lib.mjh_get_timestep(self.h)

    # ---- launch scheduling / timing (include/mjhip.h "launch scheduling")
    def set_cohorts(self, n): _chk(self.lib, self.lib.mjh_set_cohorts(self.h, int(n)), "mjh_set_cohorts")
    @property
    def cohorts(self): return self.lib.mjh_get_cohorts(self.h)
    def set_steps_per_launch(self, n): _chk(self.lib, self.lib.mjh_set_steps_per_launch(self.h, int(n)), "mjh_set_steps_per_launch")
    @property
    def steps_per_launch(self): return self.lib.mjh_get_steps_per_launch(self.h)
    @property
    def launches_per_step(self):
        """host-side launches mjh_step issues per cohort-step (1: fused kernel, or a launch chain queued as one captured graph)"""
        return self.lib.mjh_launches_per_step(self.h)

    def set_launch_timing(self, on=True): _chk(self.lib, self.lib.mjh_set_launch_timing(self.h, int(on)), "mjh_set_launch_timing")   # N > 1: every N-th launch
    def get_launch_timing(self):
        """-> (mean step-kernel duration [ms], launches) since the last call"""
        ms_, cnt = C.c_double(0), C.c_int(0)
        _chk(self.lib, self.lib.mjh_get_launch_timing(self.h, C.byref(ms_), C.byref(cnt)), "mjh_get_launch_timing")
        return ms_.value, cnt.value

    # ---- commands (mj_hw_interface.cpp:73-91)
    def set_cmd(self, ddq=None, dq=None, env0=0):
        a = None if ddq is None else np.ascontiguousarray(ddq, dtype=np.float64).reshape(-1, self.nv)
        b = None if dq is None else np.ascontiguousarray(dq, dtype=np.float64).reshape(-1, self.nv)
        n = (a if a is not None else b).shape[0]
        _chk(self.lib, self.lib.mjh_set_cmd(self.h, env0, n, capi.dptr(a), capi.dptr(b)), "mjh_set_cmd")

    def set_pd_controller(self, kp, kd):
        """in-engine joint-space PD effort controller (ros_control's effort controllers for all envs at once)"""
        _chk(self.lib, self.lib.mjh_set_pd_controller(self.h, float(kp), float(kd)), "mjh_set_pd_controller")

    def set_pd_target(self, target, env0=0):
        t = np.ascontiguousarray(target, dtype=np.float64).reshape(-1, self.nv)
        _chk(self.lib, self.lib.mjh_set_pd_target(self.h, env0, t.shape[0], capi.dptr(t)), "mjh_set_pd_target")

    def set_controlled_dofs(self, mask):
        m = np.ascontiguousarray(mask, dtype=np.int32)
        _chk(self.lib, self.lib.mjh_set_controlled_dofs(self.h, capi.iptr(m)), "mjh_set_controlled_dofs")

    def set_odom(self, lin, ang, angq):
        a, b, c = (np.ascontiguousarray(x, dtype=np.int32) for x in (lin, ang, angq))
        _chk(self.lib, self.lib.mjh_set_odom_dofs(self.h, capi.iptr(a), capi.iptr(b), capi.iptr(c)), "mjh_set_odom_dofs")

    def set_odom_vel(self, twist, env0=0):
        t = np.ascontiguousarray(twist, dtype=np.float64).reshape(-1, 6)
        _chk(self.lib, self.lib.mjh_set_odom_vel(self.h, env0, t.shape[0], capi.dptr(t)), "mjh_set_odom_vel")

    # ---- state
    def get_state(self, env0=0, n=None):
        n = self.nenv - env0 if n is None else n
        t = np.zeros(n); q = np.zeros((n, self.nq)); v = np.zeros((n, self.nv)); w = np.zeros((n, self.nv))
        _chk(self.lib, self.lib.mjh_get_state(self.h, env0, n, capi.dptr(t), capi.dptr(q), capi.dptr(v), capi.dptr(w)), "mjh_get_state")
        return t, q, v, w

    def set_state(self, qpos=None, qvel=None, time=None, warmstart=None, env0=0):
        arrs = [None if x is None else np.ascontiguousarray(x, dtype=np.float64) for x in (time, qpos, qvel, warmstart)]
        n = next(a for a in arrs if a is not None)
        n = n.shape[0] if n.ndim > 1 or arrs[0] is n else 1
        if qpos is not None:
            n = arrs[1].reshape(-1, self.nq).shape[0]
        elif qvel is not None:
            n = arrs[2].reshape(-1, self.nv).shape[0]
        _chk(self.lib, self.lib.mjh_set_state(self.h, env0, n, *[capi.dptr(a) for a in arrs]), "mjh_set_state")

    def get_joint_state(self, env0=0, n=None):
        n = self.nenv - env0 if n is None else n
        q = np.zeros((n, self.nq)); v = np.zeros((n, self.nv)); f = np.zeros((n, self.nv))
        _chk(self.lib, self.lib.mjh_get_joint_state(self.h, env0, n, capi.dptr(q), capi.dptr(v), capi.dptr(f)), "mjh_get_joint_state")
        return q, v, f

    def get_body_state(self, env0=0, n=None):
        n = self.nenv - env0 if n is None else n
        p = np.zeros((n, self.nbody, 3)); q = np.zeros((n, self.nbody, 4))
        _chk(self.lib, self.lib.mjh_get_body_state(self.h, env0, n, capi.dptr(p), capi.dptr(q)), "mjh_get_body_state")
        return p, q

    def get_geom_state(self, env0=0, n=None):
        n = self.nenv - env0 if n is None else n
        p = np.zeros((n, self.ngeom, 3)); m = np.zeros((n, self.ngeom, 9))
        _chk(self.lib, self.lib.mjh_get_geom_state(self.h, env0, n, capi.dptr(p), capi.dptr(m)), "mjh_get_geom_state")
        return p, m

    def get_field(self, name, env0=0, n=None):
        n = self.nenv - env0 if n is None else n
        w = 2 if name == "energy" else self.nv
        out = np.zeros((n, w))
        _chk(self.lib, self.lib.mjh_get_field(self.h, name.encode(), env0, n, capi.dptr(out)), "mjh_get_field")
        return out

    def get_stats(self, env0=0, n=None):
        n = self.nenv - env0 if n is None else n
        out = np.zeros((n, 4), dtype=np.int32)
        _chk(self.lib, self.lib.mjh_get_stats(self.h, env0, n, capi.iptr(out)), "mjh_get_stats")
        return out

    def get_contacts(self, env):
        mc = self.model.maxcon
        dist = np.zeros(mc); pos = np.zeros((mc, 3)); fr = np.zeros((mc, 9)); g = np.zeros((mc, 2), dtype=np.int32)
        n = _chk(self.lib, self.lib.mjh_get_contacts(self.h, env, capi.dptr(dist), capi.dptr(pos), capi.dptr(fr), capi.iptr(g)), "mjh_get_contacts")
        return dict(dist=dist[:n], pos=pos[:n], frame=fr[:n], geom=g[:n])

    def mulM(self, vec, env0=0):
        v = np.ascontiguousarray(vec, dtype=np.float64).reshape(-1, self.nv); r = np.zeros_like(v)
        _chk(self.lib, self.lib.mjh_mulM(self.h, env0, v.shape[0], capi.dptr(v), capi.dptr(r)), "mjh_mulM")
        return r

    def set_xfrc_applied(self, xfrc, env0=0):
        """d->xfrc_applied: [n, nbody, 6] force + torque per body at its centre of mass, world frame"""
        x = np.ascontiguousarray(xfrc, dtype=np.float64).reshape(-1, 6 * self.nbody)
        _chk(self.lib, self.lib.mjh_set_xfrc_applied(self.h, env0, x.shape[0], capi.dptr(x)), "mjh_set_xfrc_applied")

    def get_xfrc_applied(self, env0=0, n=None):
        n = self.nenv - env0 if n is None else n
        out = np.zeros((n, self.nbody, 6))
        _chk(self.lib, self.lib.mjh_get_xfrc_applied(self.h, env0, n, capi.dptr(out)), "mjh_get_xfrc_applied")
        return out

    def get_sensordata(self, env0=0, n=None):
        n = self.nenv - env0 if n is None else n
        out = np.zeros((n, self.model.nsensordata))
        _chk(self.lib, self.lib.mjh_get_sensordata(self.h, env0, n, capi.dptr(out)), "mjh_get_sensordata")
        return out

    def set_mocap_pose(self, mocapid, pos=None, quat=None, env0=0):
        p = None if pos is None else np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        q = None if quat is None else np.ascontiguousarray(quat, dtype=np.float64).reshape(-1, 4)
        n = (p if p is not None else q).shape[0]
        _chk(self.lib, self.lib.mjh_set_mocap_pose(self.h, env0, n, mocapid, capi.dptr(p), capi.dptr(q)), "mjh_set_mocap_pose")

    def set_env_param(self, name, values, env0=0):
        v = np.ascontiguousarray(values, dtype=np.float64)
        v = v.reshape(v.shape[0], -1)
        _chk(self.lib, self.lib.mjh_set_env_param(self.h, EP[name], env0, v.shape[0], capi.dptr(v)), "mjh_set_env_param")

    def set_initial_qpos(self, qpos, env0=0):
        q = np.ascontiguousarray(qpos, dtype=np.float64).reshape(-1, self.nq)
        _chk(self.lib, self.lib.mjh_set_initial_qpos(self.h, env0, q.shape[0], capi.dptr(q)), "mjh_set_initial_qpos")

    def reset(self, env_ids=None):
        if env_ids is None:
            _chk(self.lib, self.lib.mjh_reset(self.h, None, 0), "mjh_reset")
        else:
            ids = np.ascontiguousarray(env_ids, dtype=np.int32)
            _chk(self.lib, self.lib.mjh_reset(self.h, capi.iptr(ids), ids.shape[0]), "mjh_reset")

    def transplant_state_from(self, other, full_qpos=True):
        """add_old_state() (mj_sim.cpp:465-558): name-matched copy of the per-body state of `other` into this engine"""
        return _chk(self.lib, self.lib.mjh_transplant_state(other.h, self.h, int(bool(full_qpos))), "mjh_transplant_state")

    def set_slot_active(self, body, active, env0=0, n=None):
        n = self.nenv - env0 if n is None else n
        _chk(self.lib, self.lib.mjh_set_slot_active(self.h, env0, n, body, int(active)), "mjh_set_slot_active")

    def spawn_objects(self, envs, bodies, pos, quat=None, vel=None):
        """batched spawn service: n (env, body) pairs, pose [n,3] / [n,4] and twist [n,6] in one call"""
        ev = np.ascontiguousarray(envs, dtype=np.int32); bd = np.ascontiguousarray(bodies, dtype=np.int32)
        a = [None if x is None else np.ascontiguousarray(x, dtype=np.float64) for x in (pos, quat, vel)]
        _chk(self.lib, self.lib.mjh_spawn_objects(self.h, len(ev), capi.iptr(ev), capi.iptr(bd), *[capi.dptr(x) for x in a]), "mjh_spawn_objects")

    def destroy_objects(self, envs, bodies):
        ev = np.ascontiguousarray(envs, dtype=np.int32); bd = np.ascontiguousarray(bodies, dtype=np.int32)
        _chk(self.lib, self.lib.mjh_destroy_objects(self.h, len(ev), capi.iptr(ev), capi.iptr(bd)), "mjh_destroy_objects")

    def set_body_pose(self, env, body, pos, quat=None, vel=None):
        a = [None if x is None else np.ascontiguousarray(x, dtype=np.float64) for x in (pos, quat, vel)]
        _chk(self.lib, self.lib.mjh_set_body_pose(self.h, env, body, *[capi.dptr(x) for x in a]), "mjh_set_body_pose")

    def export_state_device(self, device_ptr):
        _chk(self.lib, self.lib.mjh_export_state_device(self.h, C.c_void_p(device_ptr)), "mjh_export_state_device")

    @property
    def state_stride(self):
        return self.lib.mjh_state_stride(self.h)

    @property
    def lds_bytes(self):
        return self.lib.mjh_lds_bytes(self.h)

    def solver_order(self):
        """2: mj_solPGS's row order (default); legacy: 1 contact patches, 0 independent pairs / groups of blocks (mjh_solver_order)"""
        return self.lib.mjh_solver_order(self.h)

    def patch_sweep(self):
        """1: contact-patch form of the sweeps (small free-body models), 0: a block form (mjh_patch_sweep)"""
        return self.lib.mjh_patch_sweep(self.h)

    def window_solver(self):
        """1: mjh_step = assemble launch + mjh_window_kernel (four envs per wavefront, rows in registers), 0: one fused launch"""
        return self.lib.mjh_window_solver(self.h)

    def pgs_schedule(self):
        """1: row order, list-scheduled (default), 2: row order, strictly sequential, 0: legacy reordering schedule (mjh_pgs_schedule)"""
        return self.lib.mjh_pgs_schedule(self.h)

    def dense_solver(self):
        """1: articulated many-body model solved by the dense row-space solver (mjh_dense_solver)"""
        return self.lib.mjh_dense_solver(self.h)

    def load_tables(self, t):
        """apply per-env parameter tables + initial poses (dict as returned by *_randomize) and reset"""
        for k in ["geom_size", "geom_rbound", "body_mass", "body_inertia", "body_invweight0", "dof_invweight0"]:
            self.set_env_param(k, t[k])
        self.set_initial_qpos(t["qpos"])
        self.reset()
        return t

    def load_s24(self, seed_base=0x5EED0000, env_offset=0):
        """Apply the per-env S24 randomisation (sizes, masses, initial poses) and reset."""
        t = self.model.s24_randomize(env_offset, self.nenv, seed_base)
        for k in ["geom_size", "geom_rbound", "body_mass", "body_inertia", "body_invweight0", "dof_invweight0"]:
            self.set_env_param(k, t[k])
        self.set_initial_qpos(t["qpos"])
        self.reset()
        return t


class Group:
    """mjh_group: the environments of one simulation sharded over the GPUs of a node (one engine + stream per device, env ranges
    contiguous); the only exchange is publish() = pack + RCCL all-gather of the state slice (include/mjhip.h "multi-GPU")."""

    def __init__(self, model, nenv_total, devices):
        self.lib = model.lib; self.model = model; self.nenv = nenv_total
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        _chk(self.lib, self.lib.mjh_group_create(model.ptr, nenv_total, capi.iptr(dv), len(dv), C.byref(h)), "mjh_group_create")
        self.h = h
        self.ndev = self.lib.mjh_group_ndev(h)
        self.ranges = []
        self.engines = []
        for k in range(self.ndev):
            a, b = C.c_int(0), C.c_int(0)
            _chk(self.lib, self.lib.mjh_group_env_range(h, k, C.byref(a), C.byref(b)), "mjh_group_env_range")
            self.ranges.append((a.value, b.value))
            self.engines.append(Engine.from_handle(model, self.lib.mjh_group_engine(h, k), b.value))
        self.stride = self.lib.mjh_group_state_stride(h)

    @property
    def uses_rccl(self): return bool(self.lib.mjh_group_uses_rccl(self.h))
    def step(self, n=1, with_inverse=False): _chk(self.lib, self.lib.mjh_group_step(self.h, n, int(with_inverse)), "mjh_group_step")
    def step1(self): _chk(self.lib, self.lib.mjh_group_step1(self.h), "mjh_group_step1")
    def inverse(self): _chk(self.lib, self.lib.mjh_group_inverse(self.h), "mjh_group_inverse")
    def step2(self): _chk(self.lib, self.lib.mjh_group_step2(self.h), "mjh_group_step2")
    def synchronize(self): _chk(self.lib, self.lib.mjh_group_synchronize(self.h), "mjh_group_synchronize")

    def publish(self):
        """-> [nenv, 1 + nq + nv] float32: time | qpos | qvel of every env in env order (device 0's copy of the all-gather)"""
        out = np.zeros((self.nenv, self.stride), dtype=np.float32)
        _chk(self.lib, self.lib.mjh_group_publish(self.h, out.ctypes.data_as(C.POINTER(C.c_float))), "mjh_group_publish")
        return out

    def publish_device(self):
        """pack + all-gather without the host copy (the gathered state stays on every device: mjh_group_state_device)"""
        _chk(self.lib, self.lib.mjh_group_publish(self.h, None), "mjh_group_publish")

    def wait_publish(self, rank, stream=None):
        _chk(self.lib, self.lib.mjh_group_wait_publish(self.h, int(rank), C.c_void_p(stream or 0)), "mjh_group_wait_publish")

    def release_publish(self, rank, stream=None):
        """the consumer on `stream` is done reading device `rank`'s gathered state: the next publish may overwrite it"""
        _chk(self.lib, self.lib.mjh_group_release_publish(self.h, int(rank), C.c_void_p(stream or 0)), "mjh_group_release_publish")

    def set_publish_timing(self, on=True): _chk(self.lib, self.lib.mjh_group_set_publish_timing(self.h, int(on)), "mjh_group_set_publish_timing")
    def get_publish_timing(self):
        """-> (mean duration of the exchange [ms], publishes) since the last call"""
        ms_, cnt = C.c_double(0), C.c_int(0)
        _chk(self.lib, self.lib.mjh_group_get_publish_timing(self.h, C.byref(ms_), C.byref(cnt)), "mjh_group_get_publish_timing")
        return ms_.value, cnt.value

    def locate(self, env):
        r, l = C.c_int(0), C.c_int(0)
        _chk(self.lib, self.lib.mjh_group_locate(self.h, env, C.byref(r), C.byref(l)), "mjh_group_locate")
        return r.value, l.value

    def close(self):
        if getattr(self, "h", None):
            for e in self.engines:
                e.close()
            self.lib.mjh_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
