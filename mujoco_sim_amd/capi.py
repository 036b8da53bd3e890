"""ctypes view of the C ABI declared in include/mjhip.h.

The Python layer is plumbing for tests and bench.py; the product is the C-ABI
shared library `libmjhip.so` (HIP kernels + host C++).  Nothing here computes
physics, and there is no CPU fallback: if the library is missing the import
fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MJHIP_LIB") or os.path.join(_HERE, "libmjhip.so")   # (MJHIP_LIB: A/B builds of the library)

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class Option(C.Structure):
    _fields_ = [
        ("timestep", C.c_double),
        ("gravity", C.c_double * 3),
        ("iterations", C.c_int),
        ("tolerance", C.c_double),
        ("impratio", C.c_double),
        ("noslip_iterations", C.c_int),
        ("disableflags", C.c_int),
        ("noslip_tolerance", C.c_double),
    ]


_INT_SIZES = ["nq", "nv", "nbody", "njnt", "ngeom", "neq", "npair", "nM", "ntree", "nexclude", "maxcon", "maxefc", "nmesh", "nmeshvert"]

# (name, ctype, length expression) in the exact order of struct mjh_model
_ARRAYS = [
    ("body_parentid", "i", "nbody"), ("body_rootid", "i", "nbody"), ("body_weldid", "i", "nbody"),
    ("body_jntadr", "i", "nbody"), ("body_jntnum", "i", "nbody"), ("body_dofadr", "i", "nbody"),
    ("body_dofnum", "i", "nbody"), ("body_treeid", "i", "nbody"), ("body_level", "i", "nbody"),
    ("body_geomadr", "i", "nbody"), ("body_geomnum", "i", "nbody"),
    ("body_pos", "d", "3*nbody"), ("body_quat", "d", "4*nbody"), ("body_ipos", "d", "3*nbody"),
    ("body_iquat", "d", "4*nbody"), ("body_mass", "d", "nbody"), ("body_inertia", "d", "3*nbody"),
    ("body_gravcomp", "d", "nbody"), ("body_invweight0", "d", "2*nbody"),
    ("jnt_type", "i", "njnt"), ("jnt_qposadr", "i", "njnt"), ("jnt_dofadr", "i", "njnt"),
    ("jnt_bodyid", "i", "njnt"), ("jnt_limited", "i", "njnt"),
    ("jnt_pos", "d", "3*njnt"), ("jnt_axis", "d", "3*njnt"), ("jnt_stiffness", "d", "njnt"),
    ("jnt_range", "d", "2*njnt"), ("jnt_margin", "d", "njnt"), ("jnt_solref", "d", "2*njnt"),
    ("jnt_solimp", "d", "5*njnt"), ("qpos0", "d", "nq"), ("qpos_spring", "d", "nq"),
    ("dof_bodyid", "i", "nv"), ("dof_jntid", "i", "nv"), ("dof_parentid", "i", "nv"),
    ("dof_Madr", "i", "nv"), ("dof_treeid", "i", "nv"),
    ("dof_armature", "d", "nv"), ("dof_damping", "d", "nv"), ("dof_frictionloss", "d", "nv"),
    ("dof_invweight0", "d", "nv"), ("dof_solref", "d", "2*nv"), ("dof_solimp", "d", "5*nv"),
    ("tree_dofadr", "i", "ntree"), ("tree_dofnum", "i", "ntree"), ("tree_bodyid", "i", "ntree"),
    ("geom_type", "i", "ngeom"), ("geom_bodyid", "i", "ngeom"), ("geom_condim", "i", "ngeom"),
    ("geom_contype", "i", "ngeom"), ("geom_conaffinity", "i", "ngeom"), ("geom_priority", "i", "ngeom"),
    ("geom_pos", "d", "3*ngeom"), ("geom_quat", "d", "4*ngeom"), ("geom_size", "d", "3*ngeom"),
    ("geom_rbound", "d", "ngeom"), ("geom_friction", "d", "3*ngeom"), ("geom_solmix", "d", "ngeom"),
    ("geom_solref", "d", "2*ngeom"), ("geom_solimp", "d", "5*ngeom"), ("geom_margin", "d", "ngeom"),
    ("geom_gap", "d", "ngeom"),
    ("pair_geom1", "i", "npair"), ("pair_geom2", "i", "npair"),
    ("eq_type", "i", "neq"), ("eq_obj1id", "i", "neq"), ("eq_obj2id", "i", "neq"), ("eq_active", "i", "neq"),
    ("eq_data", "d", "11*neq"), ("eq_solref", "d", "2*neq"), ("eq_solimp", "d", "5*neq"),
    ("geom_dataid", "i", "ngeom"), ("mesh_vertadr", "i", "nmesh"), ("mesh_vertnum", "i", "nmesh"),
    ("mesh_vert", "d", "3*nmeshvert"),
]


# appended to struct mjh_model in round 2 (behind the name tables): sites, sensors, mocap bodies
_INT_SIZES2 = ["nsite", "nsensor", "nsensordata", "nmocap"]
_ARRAYS2 = [
    ("site_bodyid", "i", "nsite"), ("site_pos", "d", "3*nsite"), ("site_quat", "d", "4*nsite"),
    ("sensor_type", "i", "nsensor"), ("sensor_objid", "i", "nsensor"), ("sensor_adr", "i", "nsensor"),
]
_ARRAYS3 = [("body_mocapid", "i", "nbody")]


class Model(C.Structure):
    _fields_ = (
        [(n, C.c_int) for n in _INT_SIZES]
        + [("opt", Option), ("meaninertia", C.c_double)]
        + [(n, c_int_p if t == "i" else c_double_p) for n, t, _ in _ARRAYS]
        + [("body_names", C.POINTER(C.c_char_p)), ("jnt_names", C.POINTER(C.c_char_p)),
           ("geom_names", C.POINTER(C.c_char_p))]
        + [(n, C.c_int) for n in _INT_SIZES2]
        + [(n, c_int_p if t == "i" else c_double_p) for n, t, _ in _ARRAYS2]
        + [("site_names", C.POINTER(C.c_char_p)), ("sensor_names", C.POINTER(C.c_char_p))]
        + [(n, c_int_p if t == "i" else c_double_p) for n, t, _ in _ARRAYS3]
    )

    def array(self, name):
        """numpy copy of a model array."""
        import numpy as np

        for n, t, expr in _ARRAYS + _ARRAYS2 + _ARRAYS3:
            if n == name:
                ln = eval(expr, {}, {k: getattr(self, k) for k in _INT_SIZES + _INT_SIZES2})
                ptr = getattr(self, n)
                if ln == 0 or not ptr:
                    return np.zeros(0, dtype=np.int32 if t == "i" else np.float64)
                return np.ctypeslib.as_array(ptr, shape=(ln,)).copy()
        raise KeyError(name)


Model_p = C.POINTER(Model)

class LoadOptions(C.Structure):
    _fields_ = [("boundmass", C.c_double), ("boundinertia", C.c_double), ("robot_gravcomp", C.c_int), ("load_meshes", C.c_int),
                ("odom_joints", C.c_uint), ("nrobot_pose", C.c_int), ("robot_pose_body", C.POINTER(C.c_char_p)), ("robot_pose", c_double_p), ("parent_child_exclude", C.c_int)]


# every symbol include/mjhip.h declares: (name, restype, argtypes)
_vp = C.c_void_p
SYMBOLS = [
    ("mjh_builder_create", _vp, []),
    ("mjh_builder_destroy", None, [_vp]),
    ("mjh_builder_set_option", None, [_vp, C.POINTER(Option)]),
    ("mjh_builder_get_option", None, [_vp, C.POINTER(Option)]),
    ("mjh_builder_set_capacity", None, [_vp, C.c_int, C.c_int]),
    ("mjh_builder_set_bounds", None, [_vp, C.c_double, C.c_double]),
    ("mjh_builder_set_balanceinertia", None, [_vp, C.c_int]),
    ("mjh_builder_add_body", C.c_int, [_vp, C.c_char_p, C.c_int, c_double_p, c_double_p, C.c_double]),
    ("mjh_builder_set_inertial", C.c_int, [_vp, C.c_int, C.c_double, c_double_p, c_double_p, c_double_p]),
    ("mjh_builder_add_joint", C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p,
                                       C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    ("mjh_builder_add_geom", C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p,
                                      c_double_p, C.c_int, C.c_int, C.c_int, C.c_double]),
    ("mjh_builder_add_mesh", C.c_int, [_vp, c_double_p, C.c_int, c_int_p, C.c_int, c_double_p]),
    ("mjh_builder_add_mesh_stl", C.c_int, [_vp, C.c_char_p, c_double_p]),
    ("mjh_builder_add_mesh_geom", C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p,
                                           C.c_int, C.c_int, C.c_int, C.c_double]),
    ("mjh_builder_add_exclude", C.c_int, [_vp, C.c_int, C.c_int]),
    ("mjh_builder_add_eq_joint", C.c_int, [_vp, C.c_int, C.c_int, c_double_p]),
    ("mjh_builder_add_eq_connect", C.c_int, [_vp, C.c_int, C.c_int, c_double_p]),
    ("mjh_builder_add_eq_weld", C.c_int, [_vp, C.c_int, C.c_int, c_double_p, C.c_double]),
    ("mjh_builder_set_mocap", C.c_int, [_vp, C.c_int]),
    ("mjh_builder_add_site", C.c_int, [_vp, C.c_char_p, C.c_int, c_double_p, c_double_p]),
    ("mjh_builder_add_sensor", C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int]),
    ("mjh_builder_compile", Model_p, [_vp]),
    ("mjh_model_destroy", None, [Model_p]),
    ("mjh_model_replicate", Model_p, [Model_p, C.c_int]),
    ("mjh_name2id", C.c_int, [Model_p, C.c_int, C.c_char_p]),
    ("mjh_id2name", C.c_char_p, [Model_p, C.c_int, C.c_int]),
    ("mjh_load_mjcf_string", Model_p, [C.c_char_p]),
    ("mjh_load_mjcf_file", Model_p, [C.c_char_p]),
    ("mjh_load_mjcf_files", Model_p, [C.POINTER(C.c_char_p), C.c_int]),
    ("mjh_load_note", C.c_char_p, []),
    ("mjh_load_default_options", None, [C.POINTER(LoadOptions)]),
    ("mjh_load_mjcf_files_opt", Model_p, [C.POINTER(C.c_char_p), C.c_int, C.POINTER(LoadOptions)]),
    ("mjh_load_set_bounds", None, [C.c_double, C.c_double]),
    ("mjh_load_set_mesh_mode", None, [C.c_int]),
    ("mjh_load_set_robot_gravcomp", None, [C.c_int]),
    ("mjh_load_set_odom_joints", None, [C.c_uint]),
    ("mjh_load_set_parent_child_exclude", None, [C.c_int]),
    ("mjh_load_set_robot_pose", None, [C.c_char_p, c_double_p]),
    ("mjh_scene_s24", Model_p, []),
    ("mjh_scene_s24_pen", Model_p, [C.c_double, C.c_int]),
    ("mjh_scene_s24_randomize", C.c_int, [Model_p, C.c_int, C.c_int, C.c_uint] + [c_double_p] * 7),
    ("mjh_scene_boxes_randomize", C.c_int, [Model_p, C.c_int, C.c_int, C.c_uint, C.c_double] + [c_double_p] * 7),
    ("mjh_scene_pendulum", Model_p, []),
    ("mjh_scene_arm7", Model_p, [C.c_int]),
    ("mjh_scene_boxpile", Model_p, [C.c_int]),
    ("mjh_create", C.c_int, [Model_p, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    ("mjh_destroy", None, [_vp]),
    ("mjh_step1", C.c_int, [_vp]),
    ("mjh_step2", C.c_int, [_vp]),
    ("mjh_step", C.c_int, [_vp, C.c_int, C.c_int]),
    ("mjh_inverse", C.c_int, [_vp]),
    ("mjh_forward", C.c_int, [_vp]),
    ("mjh_mulM", C.c_int, [_vp, C.c_int, C.c_int, c_double_p, c_double_p]),
    ("mjh_synchronize", C.c_int, [_vp]),
    ("mjh_set_cmd", C.c_int, [_vp, C.c_int, C.c_int, c_double_p, c_double_p]),
    ("mjh_set_pd_controller", C.c_int, [_vp, C.c_double, C.c_double]),
    ("mjh_set_pd_target", C.c_int, [_vp, C.c_int, C.c_int, c_double_p]),
    ("mjh_set_controlled_dofs", C.c_int, [_vp, c_int_p]),
    ("mjh_set_odom_dofs", C.c_int, [_vp, c_int_p, c_int_p, c_int_p]),
    ("mjh_set_odom_vel", C.c_int, [_vp, C.c_int, C.c_int, c_double_p]),
    ("mjh_get_joint_state", C.c_int, [_vp, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p]),
    ("mjh_get_body_state", C.c_int, [_vp, C.c_int, C.c_int, c_double_p, c_double_p]),
    ("mjh_get_geom_state", C.c_int, [_vp, C.c_int, C.c_int, c_double_p, c_double_p]),
    ("mjh_get_state", C.c_int, [_vp, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]),
    ("mjh_set_state", C.c_int, [_vp, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]),
    ("mjh_get_field", C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int, c_double_p]),
    ("mjh_get_stats", C.c_int, [_vp, C.c_int, C.c_int, c_int_p]),
    ("mjh_get_contacts", C.c_int, [_vp, C.c_int, c_double_p, c_double_p, c_double_p, c_int_p]),
    ("mjh_set_xfrc_applied", C.c_int, [_vp, C.c_int, C.c_int, c_double_p]),
    ("mjh_get_xfrc_applied", C.c_int, [_vp, C.c_int, C.c_int, c_double_p]),
    ("mjh_get_sensordata", C.c_int, [_vp, C.c_int, C.c_int, c_double_p]),
    ("mjh_set_mocap_pose", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p]),
    ("mjh_set_env_param", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, c_double_p]),
    ("mjh_transplant_state", C.c_int, [_vp, _vp, C.c_int]),
    ("mjh_set_initial_qpos", C.c_int, [_vp, C.c_int, C.c_int, c_double_p]),
    ("mjh_reset", C.c_int, [_vp, c_int_p, C.c_int]),
    ("mjh_set_slot_active", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("mjh_set_body_pose", C.c_int, [_vp, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p]),
    ("mjh_spawn_objects", C.c_int, [_vp, C.c_int, c_int_p, c_int_p, c_double_p, c_double_p, c_double_p]),
    ("mjh_destroy_objects", C.c_int, [_vp, C.c_int, c_int_p, c_int_p]),
    ("mjh_export_state_device", C.c_int, [_vp, _vp]),
    ("mjh_state_stride", C.c_int, [_vp]),
    ("mjh_mirror_create", C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    ("mjh_mirror_destroy", None, [_vp]),
    ("mjh_mirror_update", C.c_int, [_vp, C.c_int]),
    ("mjh_mirror_wait", C.c_int, [_vp]),
    ("mjh_mirror_field", C.POINTER(C.c_float), [_vp, C.c_int, C.POINTER(C.c_int)]),
    ("mjh_mirror_time", c_double_p, [_vp]),
    ("mjh_debug_stage_cycles", C.c_int, [_vp, C.c_int, c_double_p]),
    ("mjh_debug_stage_raw", C.c_int, [_vp, C.c_int, _vp]),
    ("mjh_debug_solve_probe", C.c_int, [_vp, C.c_int, C.c_int, c_double_p, c_double_p]),
    ("mjh_debug_stop_at", C.c_int, [_vp, C.c_int, C.c_int]),
    ("mjh_set_timestep", C.c_int, [_vp, C.c_double]),
    ("mjh_get_timestep", C.c_double, [_vp]),
    ("mjh_set_layout_policy", None, [C.c_int]),
    ("mjh_set_window_solver", None, [C.c_int]),
    ("mjh_window_solver", C.c_int, [_vp]),
    ("mjh_set_pgs_row_order", None, [C.c_int]),
    ("mjh_set_cohorts", C.c_int, [_vp, C.c_int]),
    ("mjh_get_cohorts", C.c_int, [_vp]),
    ("mjh_set_steps_per_launch", C.c_int, [_vp, C.c_int]),
    ("mjh_get_steps_per_launch", C.c_int, [_vp]),
    ("mjh_set_chain_graph", None, [C.c_int]),
    ("mjh_launches_per_step", C.c_int, [C.c_void_p]),
    ("mjh_set_launch_timing", C.c_int, [_vp, C.c_int]),
    ("mjh_get_launch_timing", C.c_int, [_vp, c_double_p, C.POINTER(C.c_int)]),
    ("mjh_group_create", C.c_int, [Model_p, C.c_int, c_int_p, C.c_int, C.POINTER(_vp)]),
    ("mjh_group_destroy", None, [_vp]),
    ("mjh_group_ndev", C.c_int, [_vp]),
    ("mjh_group_nenv", C.c_int, [_vp]),
    ("mjh_group_engine", _vp, [_vp, C.c_int]),
    ("mjh_group_env_range", C.c_int, [_vp, C.c_int, c_int_p, c_int_p]),
    ("mjh_group_locate", C.c_int, [_vp, C.c_int, c_int_p, c_int_p]),
    ("mjh_group_step", C.c_int, [_vp, C.c_int, C.c_int]),
    ("mjh_group_step1", C.c_int, [_vp]),
    ("mjh_group_inverse", C.c_int, [_vp]),
    ("mjh_group_step2", C.c_int, [_vp]),
    ("mjh_group_reset", C.c_int, [_vp]),
    ("mjh_group_synchronize", C.c_int, [_vp]),
    ("mjh_group_publish", C.c_int, [_vp, C.POINTER(C.c_float)]),
    ("mjh_group_state_device", C.POINTER(C.c_float), [_vp, C.c_int]),
    ("mjh_group_state_stride", C.c_int, [_vp]),
    ("mjh_group_uses_rccl", C.c_int, [_vp]),
    ("mjh_group_set_transport", None, [C.c_int]),
    ("mjh_debug_rccl_exchange", C.c_int, [C.c_int, C.c_int, C.c_ulong, C.c_int]),
    ("mjh_group_set_host_threads", None, [C.c_int]),
    ("mjh_group_host_threads", C.c_int, [C.c_void_p]),
    ("mjh_group_wait_publish", C.c_int, [_vp, C.c_int, _vp]),
    ("mjh_group_release_publish", C.c_int, [_vp, C.c_int, _vp]),
    ("mjh_group_set_publish_timing", C.c_int, [_vp, C.c_int]),
    ("mjh_group_get_publish_timing", C.c_int, [_vp, c_double_p, C.POINTER(C.c_int)]),
    ("mjh_nenv", C.c_int, [_vp]),
    ("mjh_engine_model", Model_p, [_vp]),
    ("mjh_lds_bytes", C.c_int, [_vp]),
    ("mjh_solver_order", C.c_int, [_vp]),
    ("mjh_pgs_schedule", C.c_int, [_vp]),
    ("mjh_patch_sweep", C.c_int, [_vp]),
    ("mjh_dense_solver", C.c_int, [_vp]),
    ("mjh_query_lds_bytes", C.c_int, [Model_p]),
    ("mjh_query_lds_bytes_assemble", C.c_int, [Model_p]),
    ("mjh_debug_lds_layout", C.c_int, [Model_p, C.c_char_p, C.c_int]),
    ("mjh_host_run_pd", C.c_int, [_vp, C.c_int, c_double_p, C.c_double, C.c_double, C.c_long, c_double_p, c_double_p, c_double_p]),
    ("mjh_host_run_pd_group", C.c_int, [_vp, C.c_int, c_double_p, C.c_double, C.c_double, C.c_long, C.c_int, c_double_p, c_double_p, C.POINTER(C.c_float)]),
    ("mjh_host_run_realtime", C.c_int, [_vp, C.c_int, c_double_p, C.c_double, C.c_double, C.c_long, C.c_double, c_double_p]),
    ("mjh_last_error", C.c_char_p, []),
    ("mjh_version", C.c_char_p, []),
]

_lib = None


def load(path=None, strict=True):
    """Load libmjhip.so and attach prototypes.  Raises if the library is absent
    (no fallback: the HIP extension IS the product)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    for name, res, args in SYMBOLS:
        if not strict and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def dptr(a):
    """double* of a contiguous float64 numpy array (or None)."""
    if a is None:
        return None
    return a.ctypes.data_as(c_double_p)


def iptr(a):
    if a is None:
        return None
    return a.ctypes.data_as(c_int_p)
