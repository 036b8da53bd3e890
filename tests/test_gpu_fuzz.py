"""Randomised free-body scenes through the contact-patch sweep against the oracle: 3-5 free bodies of random primitive types, sizes,
condims (1 / 3 / 4) and poses dropped into a corner made of a floor and static walls.  Every scene exercises a different mix of
1-, 4- and 6-row contacts, one- and two-body patches, partially filled patches and step schedules."""
import numpy as np
import pytest

import mujoco_sim_amd as ms
import orc
from helpers import D, set_opt


def _scene(lib, rng):
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.004)
    lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, int(rng.choice([3, 4])), -1, -1, -1)
    lib.mjh_builder_add_geom(b, b"wall_x", 0, 6, D(0.02, 0.8, 0.3), D(-0.3, 0, 0.3), None, None, 3, -1, -1, -1)
    lib.mjh_builder_add_geom(b, b"wall_y", 0, 6, D(0.8, 0.02, 0.3), D(0, -0.3, 0.3), None, None, int(rng.choice([1, 3])), -1, -1, -1)
    n = int(rng.integers(3, 6))
    for i in range(n):
        gt = int(rng.choice([2, 3, 6, 6]))                       # sphere, capsule, box (twice as likely)
        size = {2: (rng.uniform(0.04, 0.08), 0, 0), 3: (rng.uniform(0.03, 0.05), rng.uniform(0.06, 0.12), 0),
                6: tuple(rng.uniform(0.04, 0.10, 3))}[gt]
        pos = (rng.uniform(-0.12, 0.12), rng.uniform(-0.12, 0.12), 0.16 + 0.34 * i)     # apart: contacts form as the bodies land
        quat = rng.normal(size=4) * 0.3 + np.array([1, 0, 0, 0]); quat /= np.linalg.norm(quat)
        bd = lib.mjh_builder_add_body(b, b"body%d" % i, 0, D(*pos), D(*quat), 0.0)
        lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
        lib.mjh_builder_add_geom(b, None, bd, gt, D(*size), None, None, None, int(rng.choice([1, 3, 3, 4])), -1, -1, -1)
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    m.c.maxcon = 56; m.c.maxefc = 56 * 6
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16, 17, 18])
def test_random_free_body_scenes_match_the_oracle(lib, seed):
    rng = np.random.default_rng(seed)
    m = _scene(lib, rng)
    e = ms.Engine(m, 2)
    assert e.solver_order() == 2 and e.patch_sweep() == 1
    d = orc.OrcData(m.ptr); d.call("reset")
    v0 = rng.normal(size=m.nv) * 0.3
    v0[0::6] -= 0.3; v0[1::6] -= 0.3                              # push everything towards the corner
    e.set_state(qvel=np.tile(v0, (2, 1))); d.f("qvel")[:] = v0
    done = 0; seen = 0
    pos = np.arange(m.nq).reshape(-1, 7)[:, :3].ravel()          # the translational coordinates of the free bodies
    # (at the last mark positions only: a sphere's orientation, or a capsule's spin about its axis, feels no contact and integrates
    #  rounding differences; later than that rolling bodies fork between fp32 and fp64 altogether)
    for n, tol, sel in ((1, 2e-5, slice(None)), (60, 2e-3, slice(None)), (140, 3e-2, pos)):
        e.step(n - done); d.step(n - done); done = n
        st = e.get_stats(); q = e.get_state()[1]
        assert st[0, 3] == 0 and d.i("warn") == 0
        np.testing.assert_array_equal(q[0], q[1])
        if st[0, 0] == d.i("ncon") and st[0, 1] == d.i("nefc"):    # same contact set: positions must agree
            np.testing.assert_allclose(q[0][sel], d.f("qpos")[sel], atol=tol, err_msg=f"seed {seed} step {n}")
            seen += 1
    assert seen >= 2, "the contact sets should agree at least up to step 60"
    assert np.isfinite(e.get_state()[1]).all()
    e.close()


def _table_scene(lib, condim):
    """A slab on the floor carrying four small boxes: the slab sits in 8 - 10 patches (two per small box with 6-row contacts, two
    with the floor), so the sweep schedule has more steps than the unrolled loop of patch_pgs.h covers (PP_NSU = 6)."""
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.004)
    lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, condim, -1, -1, -1)
    slab = lib.mjh_builder_add_body(b, b"slab", 0, D(0, 0, 0.031), None, 0.0)
    lib.mjh_builder_add_joint(b, None, slab, 0, None, None, None, 0, 0, 0, 0, 0)
    lib.mjh_builder_add_geom(b, None, slab, 6, D(0.2, 0.2, 0.03), None, None, None, condim, -1, -1, -1)
    for i, (x, y) in enumerate(((-0.1, -0.1), (0.1, -0.1), (-0.1, 0.1), (0.1, 0.1))):
        bd = lib.mjh_builder_add_body(b, b"box%d" % i, 0, D(x, y, 0.062 + 0.03 + 0.002 * i), None, 0.0)
        lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
        lib.mjh_builder_add_geom(b, None, bd, 6, D(0.04 + 0.005 * i, 0.05, 0.03), None, None, None, condim, -1, -1, -1)
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    m.c.maxcon = 48; m.c.maxefc = 48 * 6
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("condim", [3, 4])
def test_slab_with_four_boxes_long_schedule(lib, condim):
    m = _table_scene(lib, condim)
    e = ms.Engine(m, 2)
    assert e.solver_order() == 2 and e.patch_sweep() == 1
    d = orc.OrcData(m.ptr); d.call("reset")
    done = 0
    for n, tol in ((1, 2e-5), (40, 1e-3), (120, 5e-3)):
        e.step(n - done); d.step(n - done); done = n
        st = e.get_stats(); q = e.get_state()[1]
        assert st[0, 3] == 0 and d.i("warn") == 0
        np.testing.assert_array_equal(q[0], q[1])
        assert st[0, 0] == d.i("ncon") and st[0, 1] == d.i("nefc")
        if n > 1: assert st[0, 0] >= 16                           # the four boxes and the slab rest on four corners each
        np.testing.assert_allclose(q[0], d.f("qpos"), atol=tol, err_msg=f"condim {condim} step {n}")
    e.close()
