"""Articulated robots of the reference's test assets (pr2 nv 49, tiago nv 35, hsrb4s nv 32, ridgeback_panda nv 20 =
the C3 arm on its mobile base) compiled once by this
repo's MJCF loader into table fixtures (tests/golden/make_robot_fixtures.py): CPU checks of the fixtures and of the
oracle; the GPU parity test lives in test_gpu_parity.py."""
import os

import numpy as np
import pytest

import orc
from conftest import ROOT
from helpers import load_model_tables

ROBOTS = ["pr2", "tiago", "hsrb4s", "ridgeback_panda", "pr2_world", "hsrb4s_world", "pr2_mesh", "pr2_world_mesh", "c5_pendulum_bowl_mesh",
          "c4_pr2_world_objects_mesh", "tiago_mesh", "hsrb4s_mesh", "armar6_mesh", "ridgeback_panda_mesh"]
FILES = {"pr2": "pr2/pr2.xml", "tiago": "tiago/tiago.xml", "hsrb4s": "hsrb4s/hsrb4s.xml",
         "ridgeback_panda": "ridgeback_panda/ridgeback_panda.xml",
         "pr2_world": "../world/empty.xml+pr2/pr2.xml", "hsrb4s_world": "../world/empty.xml+hsrb4s/hsrb4s.xml",
         # PR2 with its 18 STL meshes (37 mesh geoms colliding as convex hulls); the others are compiled with the meshes off
         "pr2_mesh": "pr2/pr2.xml", "pr2_world_mesh": "../world/empty.xml+pr2/pr2.xml",
         # C5 as launched (launch/multi_mujoco_sim.launch:3-4): world = pendulum.xml, "robot" = bowl.xml (37 static mesh geoms)
         "c5_pendulum_bowl_mesh": "pendulum.xml+bowl.xml",
         # C4: PR2 on the world floor + 8 spawnable objects (cubes / spheres / cylinders; the pool text lives in the generator)
         "c4_pr2_world_objects_mesh": None,
         "tiago_mesh": "tiago/tiago.xml", "hsrb4s_mesh": "hsrb4s/hsrb4s.xml", "armar6_mesh": "armar/armar6.xml",
         # (loaded with mjh_load_set_parent_child_exclude(2): the wrapper's disable_parent_child_collision_level, mujoco_compile.cpp:250-290)
         "ridgeback_panda_mesh": "ridgeback_panda/ridgeback_panda.xml"}
PC_EXCLUDE = {"ridgeback_panda_mesh": 2}
REF = "/root/reference/model/test"


def robot_command(m, k):
    jt = m.array("jnt_type"); da = m.array("jnt_dofadr")
    ddq = np.zeros(m.nv)
    for j in range(m.njnt):
        if jt[j] in (2, 3):
            ddq[da[j]] = 0.8 * np.sin(0.05 * k + 0.37 * j)
    return ddq


@pytest.mark.parametrize("name", ROBOTS)
def test_oracle_reproduces_robot_golden(lib, name):
    m, z = load_model_tables(os.path.join(ROOT, "tests", "golden", f"robot_{name}.npz"))
    if name.startswith("c4"):
        assert m.ntree == 9 and m.nv == 49 + 8 * 6 and m.c.nmesh == 18
    elif name.startswith("c5"):
        assert m.ntree == 3 and m.nv == 9 and m.c.nmesh == 2 and (m.array("geom_type") == 7).sum() == 37
    else:
        assert m.ntree == 1 and m.nv >= 20
    if "_world" in name:
        assert m.array("geom_type")[0] == 0 and m.array("geom_condim")[0] == 4      # the world file's floor plane
    d = orc.OrcData(m.ptr)
    d.ifield("controlled")[:] = z["controlled"]
    keep = [int(k) for k in z["keep"]]
    d.f("qvel")[:] = z["qvel0"]
    if name.startswith("pr2") and name.endswith("_mesh"):
        assert m.c.nmesh == 18 and (m.array("geom_type") == 7).sum() == 37 and m.c.npair > 1000
    for k in range(1, min(100, keep[-1]) + 1):
        d.f("ddq")[:] = robot_command(m, k)
        d.step(1, 1)
        if k in keep:
            np.testing.assert_allclose(d.f("qpos"), z[f"qpos_{k}"], rtol=0, atol=1e-9)
            np.testing.assert_allclose(d.f("qfrc_inverse"), z[f"qfrc_inverse_{k}"], rtol=1e-7, atol=1e-7)
            assert d.i("nefc") == int(z[f"nefc_{k}"])
    # the computed-torque wrapper makes the controlled joints follow the commanded acceleration:
    # qacc of a controlled, unconstrained dof equals ddq (mj_sim.cpp:1055-1063)
    assert np.isfinite(d.f("qpos")).all()


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("name", ROBOTS)
def test_loader_still_produces_the_fixture_tables(lib, name):
    if FILES[name] is None:
        pytest.skip("composed with generator-side text")
    import mujoco_sim_amd as ms
    from mujoco_sim_amd import capi
    lib.mjh_load_set_bounds(1e-6, 1e-6)      # as the reference does before mj_loadXML (mj_sim.cpp:584-590)
    lib.mjh_load_set_mesh_mode(1 if name.endswith("_mesh") else 0)
    lib.mjh_load_set_parent_child_exclude(PC_EXCLUDE.get(name, 0))
    try:
        m = ms.load_mjcf(paths=[os.path.join(REF, r) for r in FILES[name].split("+")])
    finally:
        lib.mjh_load_set_bounds(0.0, 0.0); lib.mjh_load_set_mesh_mode(1); lib.mjh_load_set_parent_child_exclude(0)
    f, z = load_model_tables(os.path.join(ROOT, "tests", "golden", f"robot_{name}.npz"))
    for k in ("nq", "nv", "nbody", "njnt", "ngeom", "neq", "npair", "nM", "ntree", "nmesh", "nmeshvert"):
        assert getattr(m, k) == getattr(f, k), k
    for n, t, _ in capi._ARRAYS:
        np.testing.assert_allclose(m.array(n), f.array(n), rtol=1e-13, atol=1e-13, err_msg=n)
    # working set with the fixture's contact capacity fits one CU's LDS
    lds = lib.mjh_query_lds_bytes(f.ptr)
    assert 0 < lds <= 160 * 1024, lds
