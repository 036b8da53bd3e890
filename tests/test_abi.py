"""CPU tests of the C-ABI shared library: it loads, exports every symbol include/mjhip.h declares,
the host-side model compiler produces the documented tables, and the engine refuses to run without
a HIP device (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import mujoco_sim_amd as ms
from conftest import ROOT, has_gpu
from helpers import D, free_body_model, hinge_pendulum_model, two_link_model
from mujoco_sim_amd import capi


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "mjhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mjh_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) > 40
    raw = C.CDLL(capi.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(raw, s)]
    assert not missing, f"declared in include/mjhip.h but not exported: {missing}"
    bound = {n for n, _, _ in capi.SYMBOLS}
    assert declared <= bound | {"mjh_geom_rbound"}, declared - bound


def test_s24_model_tables(lib):
    m = ms.scene("s24")
    assert (m.nq, m.nv, m.nbody, m.ngeom, m.ntree) == (28, 24, 5, 9, 4)
    assert m.opt.timestep == 0.005 and m.opt.gravity[2] == -9.81 and m.opt.iterations == 100
    mass = m.array("body_mass")
    np.testing.assert_allclose(mass[1:], 1000 * 8 * 0.0875**3, rtol=1e-12)
    I = m.array("body_inertia").reshape(-1, 3)
    np.testing.assert_allclose(I[1], mass[1] / 3 * 2 * 0.0875**2 * np.ones(3), rtol=1e-12)
    # invweight0 of a free body: 1/m and mean(1/I)
    bw = m.array("body_invweight0").reshape(-1, 2)
    np.testing.assert_allclose(bw[1], [1 / mass[1], 1 / I[1, 0]], rtol=1e-10)
    np.testing.assert_allclose(m.array("dof_invweight0")[:6], [1 / mass[1]] * 3 + [1 / I[1, 0]] * 3, rtol=1e-10)
    # candidate pairs: 4 boxes x (floor + 4 walls) + 6 box-box; static-static pairs filtered
    assert m.npair == 26
    t = m.array("geom_type")
    g1, g2 = m.array("pair_geom1"), m.array("pair_geom2")
    assert (t[g1] <= t[g2]).all()
    assert m.array("geom_condim")[0] == 4 and np.allclose(m.array("geom_friction")[:3], [2, 0.05, 0.01])
    assert m.name2id(0, "box2") == 3 and m.name2id(1, "box0_free") == 0 and m.name2id(2, "floor") == 0
    assert m.name2id(0, "nope") == -1


def test_s24_randomize_is_deterministic_and_in_range(lib):
    m = ms.scene("s24")
    a = m.s24_randomize(0, 8)
    b = m.s24_randomize(4, 4)
    for k in a:
        np.testing.assert_array_equal(a[k][4:], b[k])
    hs = a["geom_size"].reshape(8, 9, 3)[:, 5:, :]
    assert hs.min() >= 0.05 and hs.max() <= 0.125 and np.unique(hs).size == hs.size
    q = a["qpos"].reshape(8, 4, 7)
    np.testing.assert_allclose(np.linalg.norm(q[:, :, 3:], axis=-1), 1, atol=1e-12)
    np.testing.assert_allclose(q[:, :, 2], np.broadcast_to(0.15 + 0.3 * np.arange(4), (8, 4)))
    assert np.abs(q[:, :, :2]).max() <= 0.05
    vol = 8 * hs.prod(-1)
    np.testing.assert_allclose(a["body_mass"][:, 1:], 1000 * vol, rtol=1e-12)


def test_pendulum_and_arm_models(lib):
    p = ms.scene("pendulum")
    assert (p.nq, p.nv, p.nbody) == (12, 9, 4)
    np.testing.assert_allclose(p.array("dof_damping"), 0.5)
    assert p.opt.gravity[2] == -0.1
    # sphere r=.1: m = 1000*4/3*pi*r^3
    np.testing.assert_allclose(p.array("body_mass")[1], 1000 * 4 / 3 * np.pi * 1e-3, rtol=1e-12)
    a = ms.scene("arm7", 1)
    assert (a.nq, a.nv, a.nbody, a.npair) == (7, 7, 8, 0)
    assert a.array("jnt_limited").all() and np.allclose(a.array("jnt_range")[6:8], [-3.0718, -0.0698])
    assert (a.array("body_gravcomp")[1:] == 1).all()
    # serial chain: dof i's parent is i-1; nM = 28
    np.testing.assert_array_equal(a.array("dof_parentid"), np.arange(-1, 6))
    assert a.nM == 28


def test_invweight0_of_hinge_pendulum(lib):
    m = hinge_pendulum_model(lib, mass=2.0, length=1.0, inertia=0.1)
    # M = I + m l^2 ; dof_invweight0 = 1/M ; body translational invweight = l^2/M /3 * (2 of 3 axes move) ...
    Mq = 0.1 + 2.0
    np.testing.assert_allclose(m.array("dof_invweight0"), [1 / Mq], rtol=1e-10)
    np.testing.assert_allclose(m.meaninertia, Mq, rtol=1e-10)
    bw = m.array("body_invweight0")[2:4]
    np.testing.assert_allclose(bw, [1.0 / Mq / 3, 1.0 / Mq / 3], rtol=1e-10)


def test_body_reordering_is_depth_first(lib):
    b = lib.mjh_builder_create()
    a1 = lib.mjh_builder_add_body(b, b"a", 0, D(0, 0, 1), None, 0.0)
    b1 = lib.mjh_builder_add_body(b, b"b", 0, D(1, 0, 1), None, 0.0)
    a2 = lib.mjh_builder_add_body(b, b"a_child", a1, D(0, 0, 1), None, 0.0)
    for bd, nm in ((a1, b"ja"), (b1, b"jb"), (a2, b"jc")):
        lib.mjh_builder_add_joint(b, nm, bd, 3, None, D(0, 1, 0), None, 0, 0, 0, 0, 0)
        lib.mjh_builder_add_geom(b, None, bd, 2, D(0.1, 0, 0), None, None, None, -1, -1, -1, -1)
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    assert [m.name2id(0, n) for n in ("a", "a_child", "b")] == [1, 2, 3]
    np.testing.assert_array_equal(m.array("tree_dofadr"), [0, 2])
    np.testing.assert_array_equal(m.array("tree_dofnum"), [2, 1])
    np.testing.assert_array_equal(m.array("dof_parentid"), [-1, 0, -1])


def test_builder_rejects_bad_input(lib):
    b = lib.mjh_builder_create()
    assert lib.mjh_builder_add_body(b, b"x", 7, None, None, 0.0) < 0
    assert b"parent" in lib.mjh_last_error()
    bd = lib.mjh_builder_add_body(b, b"x", 0, None, None, 0.0)
    assert lib.mjh_builder_add_joint(b, b"j", bd, 9, None, None, None, 0, 0, 0, 0, 0) < 0
    # massless jointed body without armature -> singular M
    lib.mjh_builder_add_joint(b, b"j", bd, 3, None, None, None, 0, 0, 0, 0, 0)
    assert not lib.mjh_builder_compile(b)
    assert b"positive definite" in lib.mjh_last_error()
    lib.mjh_builder_destroy(b)


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_engine_fails_loudly_without_gpu(lib):
    m = ms.scene("s24")
    with pytest.raises(ms.engine.MjhError, match="no HIP device"):
        ms.Engine(m, 4)


def test_s24_working_set_fits_eight_envs_per_cu(lib):
    """The headline scene (40-contact capacity) must keep 8 environments resident per CU.  LDS is allocated in
    1280-byte granules on gfx950 (measured: 15 328 B -> 10 per CU, 15 664 B -> 9), so the budget is 16 granules =
    20 480 B per env.  (The unrolled patch sweep keeps its row records in registers, 235 VGPRs = two waves per SIMD = 8 per CU
    whatever the LDS allows, DESIGN.md section 4c, so the patch pool takes the room.)"""
    m = ms.scene("s24")
    assert m.maxcon == 40
    nbytes = lib.mjh_query_lds_bytes(m.ptr)
    assert 0 < nbytes <= 16 * 1280, nbytes


def test_lds_layout_of_the_launches_that_wait_for_lds(lib):
    """Capacity planning without a device (mjh_debug_lds_layout, tools/lds_layout.py).  Two budgets found by measurement in round 5: the assemble-only
    launch of the window chain for S24D (contact capacity 96) fits 13.2 granules of 1280 B — nine environments per CU beside the window wavefronts
    (25.3 KB, six per CU: S24D 5.28 instead of 5.75 M env-steps/s) —, and C3's four arms per wavefront fit 16 granules (eight workgroups per CU = all
    2048 resident; pair-less models keep no LDS for per-env geom sizes)."""
    import ctypes as C
    def layout(m):
        buf = C.create_string_buffer(8192)
        n = lib.mjh_debug_lds_layout(m.ptr, buf, 8192)
        assert n > 0
        return dict((k, int(v)) for k, v in (ln.split() for ln in buf.value.decode().strip().splitlines()))
    s24d = ms.scene("s24pen", 0.175, 96)
    L = layout(s24d)
    assert L["lds_bytes_pre"] == lib.mjh_query_lds_bytes_assemble(s24d.ptr) <= 14 * 1280 and L["lds_bytes"] == lib.mjh_query_lds_bytes(s24d.ptr)
    # what that launch does not use lies behind its extent: the pair schedule, the condim-4 extension, the per-base scratch vectors
    assert min(L["sched"], L["ext"], L["bv"], L["phi"]) * 4 >= L["lds_bytes_pre"] and max(L["con"], L["blki"], L["blkf"]) * 4 < L["lds_bytes_pre"]
    s24 = ms.scene("s24")
    assert lib.mjh_query_lds_bytes_assemble(s24.ptr) == 18192          # (the headline's assemble-only launch: unchanged, its waves wait for SIMDs, not LDS)
    c3 = ms.scene("arm7", 1).replicate(4)
    L3 = layout(c3)
    assert L3["lds_bytes"] <= 16 * 1280 and L3["p_gsize"] == L3["gpos"] and L3["p_rbound"] == L3["gmat"], L3["lds_bytes"]


def test_model_replicate_keeps_instances_apart(lib):
    """mjh_model_replicate (sub-wave packing): moving trees are copied, static geometry is shared, no pair joins two instances"""
    import mujoco_sim_amd as ms
    m = ms.scene("s24")
    r = m.replicate(3)
    nstatic = int((m.array("body_weldid")[m.array("geom_bodyid")] == 0).sum())
    assert r.nv == 3 * m.nv and r.nq == 3 * m.nq and r.c.nbody == 1 + 3 * (m.c.nbody - 1) and r.c.ngeom == nstatic + 3 * (m.c.ngeom - nstatic)
    assert r.npair == 3 * m.npair and r.c.maxcon == 3 * m.c.maxcon
    gb = r.array("geom_bodyid"); root = r.array("body_rootid")
    inst = lambda b: 0 if b < m.c.nbody else 1 + (b - m.c.nbody) // (m.c.nbody - 1)
    for g1, g2 in zip(r.array("pair_geom1"), r.array("pair_geom2")):
        b1, b2 = gb[g1], gb[g2]
        assert b1 == 0 or b2 == 0 or inst(b1) == inst(b2)
    np.testing.assert_array_equal(r.array("qpos0").reshape(3, -1), np.tile(m.array("qpos0"), (3, 1)))
    assert lib.mjh_id2name(r.ptr, 0, m.c.nbody).endswith(b"#1")
    assert not lib.mjh_model_replicate(m.ptr, 0)
