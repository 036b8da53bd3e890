"""MJCF-subset loader — CPU tests.  The XML below is this repo's own test data; the last
test additionally parses the reference's pendulum.xml when the reference tree is present (CPU container only)."""
import os

import numpy as np
import pytest

import mujoco_sim_amd as ms
import orc
from helpers import D, set_opt

ARM = """<?xml version="1.0"?>
<mujoco model="two_link">
  <compiler angle="degree" autolimits="true"/>
  <option timestep="0.002" gravity="0 0 -9.81" iterations="50" tolerance="1e-9"><flag warmstart="enable"/></option>
  <default><joint damping="0.2"/><geom friction="0.8 0.01 0.001" condim="4"/></default>
  <!-- a comment <with> markup -->
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <body name="upper" pos="0 0 1.5" euler="0 0 90">
      <joint name="shoulder" axis="0 1 0" range="-90 45"/>
      <geom name="upper_g" type="capsule" size=".04 .25" pos="0 0 -.25"/>
      <body name="lower" pos="0 0 -.5">
        <joint name="elbow" type="hinge" axis="0 1 0" damping="0.05" limited="false" range="-10 10"/>
        <inertial pos="0 0 -.2" mass="1.2" diaginertia=".02 .02 .002"/>
        <geom type="box" size=".03 .03 .2" pos="0 0 -.2" friction="1.5"/>
      </body>
    </body>
    <body name="ball" pos="0.5 0 0.3" gravcomp="1">
      <freejoint name="ball_free"/>
      <geom type="sphere" size=".1" density="500"/>
      <geom type="mesh" mesh="nope"/>
    </body>
  </worldbody>
  <contact><exclude body1="upper" body2="lower"/></contact>
  <equality><joint joint1="elbow" joint2="shoulder" polycoef="0.1 2"/></equality>
  <actuator/>
</mujoco>
"""


def test_loader_matches_builder_api(lib):
    m = ms.load_mjcf(ARM)
    assert "mesh" in m.note and "actuator" in m.note
    assert (m.nq, m.nv, m.nbody, m.njnt, m.neq) == (9, 8, 4, 3, 1)
    assert m.opt.timestep == 0.002 and m.opt.iterations == 50 and m.opt.tolerance == 1e-9
    assert [m.name2id(1, n) for n in ("shoulder", "elbow", "ball_free")] == [0, 1, 2]
    # degrees -> radians for hinge ranges, autolimits, explicit limited="false"
    np.testing.assert_allclose(m.array("jnt_range")[:2], np.deg2rad([-90, 45]))
    np.testing.assert_array_equal(m.array("jnt_limited"), [1, 0, 0])
    np.testing.assert_allclose(m.array("dof_damping")[:2], [0.2, 0.05])
    # euler 0 0 90 (degrees) -> quaternion about z
    np.testing.assert_allclose(m.array("body_quat")[4:8], [np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)], atol=1e-12)
    # defaults + overrides on geoms
    fr = m.array("geom_friction").reshape(-1, 3)
    np.testing.assert_allclose(fr[m.name2id(2, "upper_g")], [0.8, 0.01, 0.001]); np.testing.assert_allclose(fr[2], [1.5, 0.01, 0.001])
    assert (m.array("geom_condim") == 4).all()
    # explicit inertial vs geom-derived (sphere, density 500)
    mass = m.array("body_mass")
    np.testing.assert_allclose(mass[2], 1.2); np.testing.assert_allclose(mass[3], 500 * 4 / 3 * np.pi * 1e-3)
    assert m.array("body_gravcomp")[3] == 1
    np.testing.assert_allclose(m.array("eq_data")[:2], [0.1, 2.0])
    # the upper/lower pair is excluded (and parent-filtered anyway); ball vs arm geoms remain
    pairs = set(zip(m.array("pair_geom1").tolist(), m.array("pair_geom2").tolist()))
    assert (1, 2) not in pairs and len(pairs) >= 3
    # same model through the builder API gives the same derived constants
    b = lib.mjh_builder_create(); set_opt(lib, b, timestep=0.002, iterations=50, tolerance=1e-9)
    fr = D(0.8, 0.01, 0.001)
    lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, .05), None, None, fr, 4, -1, -1, -1)
    u = lib.mjh_builder_add_body(b, b"upper", 0, D(0, 0, 1.5), D(np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)), 0.0)
    lib.mjh_builder_add_joint(b, b"shoulder", u, 3, None, D(0, 1, 0), D(*np.deg2rad([-90, 45])), 0.2, 0, 0, 0, 0)
    lib.mjh_builder_add_geom(b, b"upper_g", u, 3, D(.04, .25, 0), D(0, 0, -.25), None, fr, 4, -1, -1, -1)
    lo = lib.mjh_builder_add_body(b, b"lower", u, D(0, 0, -.5), None, 0.0)
    lib.mjh_builder_add_joint(b, b"elbow", lo, 3, None, D(0, 1, 0), None, 0.05, 0, 0, 0, 0)
    lib.mjh_builder_set_inertial(b, lo, 1.2, D(0, 0, -.2), None, D(.02, .02, .002))
    lib.mjh_builder_add_geom(b, None, lo, 6, D(.03, .03, .2), D(0, 0, -.2), None, D(1.5, 0.01, 0.001), 4, -1, -1, -1)
    ba = lib.mjh_builder_add_body(b, b"ball", 0, D(.5, 0, .3), None, 1.0)
    lib.mjh_builder_add_joint(b, b"ball_free", ba, 0, None, None, None, 0.2, 0, 0, 0, 0)
    lib.mjh_builder_add_geom(b, None, ba, 2, D(.1, 0, 0), None, None, fr, 4, -1, -1, 500.0)
    lib.mjh_builder_add_exclude(b, u, lo); lib.mjh_builder_add_eq_joint(b, 1, 0, D(0.1, 2, 0, 0, 0))
    ref = ms.Model(lib.mjh_builder_compile(b), lib); lib.mjh_builder_destroy(b)
    for name in ("body_mass", "body_inertia", "body_invweight0", "dof_invweight0", "qpos0", "geom_rbound", "dof_damping"):
        np.testing.assert_allclose(m.array(name), ref.array(name), rtol=1e-12, atol=1e-14, err_msg=name)
    assert abs(m.meaninertia - ref.meaninertia) < 1e-12
    # and it steps in the oracle
    d = orc.OrcData(m.ptr); d.step(50)
    assert np.isfinite(d.f("qpos")).all() and d.i("nefc") >= 1          # the joint equality row is always there


def test_loader_rejects_malformed_input(lib):
    for bad, msg in (("<mujoco><worldbody><body></worldbody></mujoco>", "mismatched"), ("<nope/>", "root element"),
                     ("<mujoco><worldbody><joint/></worldbody></mujoco>", "joint in worldbody"),
                     ("<mujoco><worldbody><body><joint type='screw'/></body></worldbody></mujoco>", "unknown joint type")):
        assert not lib.mjh_load_mjcf_string(bad.encode())
        assert msg in lib.mjh_last_error().decode(), (bad, lib.mjh_last_error())
    assert not lib.mjh_load_mjcf_file(b"/no/such/file.xml")


REF_PENDULUM = "/root/reference/model/test/pendulum.xml"


@pytest.mark.skipif(not os.path.exists(REF_PENDULUM), reason="reference tree only exists in the CPU container")
def test_reference_pendulum_xml_equals_the_restated_scene(lib):
    """C1: the programmatic scene (csrc/scenes.cpp) restates model/test/pendulum.xml exactly"""
    a = ms.load_mjcf(path=REF_PENDULUM)
    b = ms.scene("pendulum")
    assert (a.nq, a.nv, a.nbody, a.ngeom) == (b.nq, b.nv, b.nbody, b.ngeom)
    for name in ("body_pos", "body_mass", "body_inertia", "jnt_pos", "jnt_type", "dof_damping", "geom_type", "geom_size", "qpos0",
                 "body_invweight0", "dof_invweight0"):
        np.testing.assert_allclose(a.array(name), b.array(name), rtol=1e-12, atol=1e-14, err_msg=name)
    assert a.opt.timestep == b.opt.timestep and a.opt.gravity[2] == b.opt.gravity[2] == -0.1


WORLD = """<mujoco>
  <option timestep="0.004" gravity="0 0 -9.81"/>
  <worldbody><geom name="floor" type="plane" size="0 0 .05" condim="4" friction="2 0.05 0.01"/></worldbody>
</mujoco>"""
CART = """<mujoco model="cart">
  <compiler angle="radian"/>
  <option timestep="0.001"/>
  <default><geom friction="0.7 0.005 0.0001"/></default>
  <worldbody>
    <body name="cart" pos="0 0 0.06">
      <freejoint name="cart_free"/>
      <geom name="chassis" type="box" size=".2 .1 .02"/>
      <body name="frame" pos="0 0 0.05"><joint name="tilt" axis="0 1 0" limited="true" range="-1 1"/></body>
      <body name="wheel_l" pos="0 .12 0" quat="0.7071068 0.7071068 0 0"><joint name="wl" axis="0 0 1"/><geom type="cylinder" size=".06 .01"/></body>
      <body name="wheel_r" pos="0 -.12 0" quat="0.7071068 0.7071068 0 0"><joint name="wr" axis="0 0 1"/><geom type="cylinder" size=".06 .01"/></body>
    </body>
  </worldbody>
</mujoco>"""


def test_world_plus_robot_composition_and_bounds(lib, tmp_path):
    """mjh_load_mjcf_files: <option> from the first (world) file, <compiler>/<default> per file, bodies of every file in one
    model; mjh_load_set_bounds: the reference's boundmass / boundinertia floor makes the massless frame body admissible."""
    w = tmp_path / "world.xml"; w.write_text(WORLD)
    c = tmp_path / "cart.xml"; c.write_text(CART)
    with pytest.raises(Exception, match="positive definite"):
        ms.load_mjcf(paths=[str(w), str(c)])                       # massless moving body
    lib.mjh_load_set_bounds(1e-6, 1e-6)
    try:
        m = ms.load_mjcf(paths=[str(w), str(c)])
    finally:
        lib.mjh_load_set_bounds(0.0, 0.0)
    assert m.opt.timestep == 0.004                                  # the world file's option wins
    assert (m.nbody, m.njnt, m.nv, m.ngeom) == (5, 4, 9, 4)
    assert m.array("geom_type").tolist() == [0, 6, 5, 5] and m.array("geom_condim")[0] == 4
    np.testing.assert_allclose(m.array("geom_friction").reshape(-1, 3)[1], [0.7, 0.005, 0.0001])   # the cart file's default
    np.testing.assert_allclose(m.array("geom_friction").reshape(-1, 3)[0], [2, 0.05, 0.01])
    assert m.array("jnt_limited").tolist() == [0, 1, 0, 0]
    np.testing.assert_allclose(m.array("jnt_range")[2:4], [-1, 1])      # radians (the cart file's <compiler>)
    bm = m.array("body_mass"); assert bm[m.name2id(0, "frame")] == 1e-6
    # plane-cylinder + plane-box pairs are in the pair list; the cart settles on its wheels in the oracle
    pairs = set(zip(m.array("pair_geom1").tolist(), m.array("pair_geom2").tolist()))
    assert {(0, 1), (0, 2), (0, 3)} <= pairs
    d = orc.OrcData(m.ptr)
    d.step(600)
    assert d.i("ncon") >= 4 and np.abs(d.f("qvel")).max() < 0.05 and abs(d.f("qpos")[2] - 0.06) < 5e-3


def test_geom_fromto():
    """<geom fromto>: position = midpoint, z axis along the segment, half-length from the distance"""
    import mujoco_sim_amd as ms
    m = ms.load_mjcf("""<mujoco><worldbody><body pos="0 0 1"><freejoint/>
        <geom type="capsule" size="0.05" fromto="0 0 0 0.3 0 0.4"/>
        <geom type="cylinder" size="0.02" fromto="0 0 0 0 0 -0.2"/></body></worldbody></mujoco>""")
    note = ms.capi.load().mjh_load_note()
    assert m.c.ngeom == 2 and b"skipped" not in note and b"ignored" not in note and b"Newton" in note     # (only the always-on solver note)
    np.testing.assert_allclose(m.array("geom_size").reshape(-1, 3), [[0.05, 0.25, 0], [0.02, 0.1, 0]], atol=1e-12)
    np.testing.assert_allclose(m.array("geom_pos").reshape(-1, 3), [[0.15, 0, 0.2], [0, 0, -0.1]], atol=1e-12)
    q = m.array("geom_quat").reshape(-1, 4)
    w, x, y, z = q[0]
    zax = np.array([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)])
    np.testing.assert_allclose(zax, [0.6, 0, 0.8], atol=1e-12)
    np.testing.assert_allclose(np.abs(q[1]), [0, 1, 0, 0], atol=1e-12)          # pointing down: half a turn about x


def test_default_classes_childclass_and_include(tmp_path):
    """<default class> (nested, inherited), class= / childclass= on elements, and <include file> spliced in place"""
    import mujoco_sim_amd as ms
    (tmp_path / "parts").mkdir()
    (tmp_path / "parts" / "arm.xml").write_text("""<mujoco><body name="link2" pos="0 0 0.3">
        <joint name="j2" class="stiff"/><geom class="thin" fromto="0 0 0 0 0 0.3"/></body></mujoco>""")
    (tmp_path / "defs.xml").write_text("""<mujoco><default>
        <geom type="capsule" size="0.04" friction="0.7 0.005 0.0001"/><joint damping="0.2" axis="0 1 0"/>
        <default class="thin"><geom size="0.01"/></default>
        <default class="stiff"><joint stiffness="5" damping="1.5"/>
          <default class="stiffer"><joint stiffness="50"/></default></default>
      </default></mujoco>""")
    (tmp_path / "m.xml").write_text("""<mujoco><compiler angle="radian"/><include file="defs.xml"/>
      <worldbody><body name="link1" pos="0 0 1" childclass="stiffer">
          <joint name="j1"/><geom fromto="0 0 0 0 0 0.3"/><geom class="main" type="sphere" size="0.06" pos="0 0 0.3"/>
          <include file="parts/arm.xml"/>
      </body></worldbody></mujoco>""")
    m = ms.load_mjcf(path=str(tmp_path / "m.xml"))
    note = ms.capi.load().mjh_load_note()
    assert b"skipped" not in note and b"ignored" not in note and m.c.nbody == 3 and m.njnt == 2 and m.c.ngeom == 3
    # j1: childclass "stiffer" = stiff (damping 1.5) + stiffness 50, axis from the top-level default; j2: explicit class "stiff"
    np.testing.assert_allclose(m.array("jnt_stiffness"), [50, 5]); np.testing.assert_allclose(m.array("dof_damping"), [1.5, 1.5])
    np.testing.assert_allclose(m.array("jnt_axis").reshape(-1, 3), [[0, 1, 0], [0, 1, 0]])
    # geoms: capsule from "main" via the class chain (size 0.04), the sphere with its own size, the included one "thin"
    np.testing.assert_array_equal(m.array("geom_type"), [3, 2, 3])
    np.testing.assert_allclose(m.array("geom_size").reshape(-1, 3)[:, 0], [0.04, 0.06, 0.01])
    np.testing.assert_allclose(m.array("geom_friction").reshape(-1, 3)[:, 0], [0.7, 0.7, 0.7])


def test_compiler_balanceinertia():
    """<compiler balanceinertia="true"> (mujoco_compile.cpp:157-160): an inertia violating A + B >= C becomes its mean"""
    import mujoco_sim_amd as ms
    xml = """<mujoco><compiler balanceinertia="%s"/><worldbody><body pos="0 0 1"><freejoint/>
        <inertial pos="0 0 0" mass="1" diaginertia="0.1 0.2 0.5"/><geom type="sphere" size="0.1"/></body></worldbody></mujoco>"""
    np.testing.assert_allclose(ms.load_mjcf(xml % "false").array("body_inertia")[3:6], [0.1, 0.2, 0.5])
    np.testing.assert_allclose(ms.load_mjcf(xml % "true").array("body_inertia")[3:6], [0.8 / 3] * 3)


def test_robot_gravcomp_override(tmp_path, lib):
    """~disable_gravity: every body of a robot file gets gravcomp 1 (or 0), whatever the file says (mj_sim.cpp:301-310)"""
    import mujoco_sim_amd as ms
    (tmp_path / "world.xml").write_text('<mujoco><worldbody><geom type="plane" size="0 0 0.05"/><body name="crate" pos="1 0 0.2" gravcomp="0.5"><freejoint/><geom type="box" size="0.1 0.1 0.1"/></body></worldbody></mujoco>')
    (tmp_path / "robot.xml").write_text('<mujoco><worldbody><body name="base" pos="0 0 0.5" gravcomp="0"><freejoint/><geom type="box" size="0.2 0.2 0.1"/>'
                                        '<body name="arm" pos="0 0 0.2"><joint axis="0 1 0"/><geom type="capsule" size="0.03 0.2"/></body></body></worldbody></mujoco>')
    paths = [str(tmp_path / "world.xml"), str(tmp_path / "robot.xml")]
    try:
        np.testing.assert_allclose(ms.load_mjcf(paths=paths).array("body_gravcomp"), [0, 0.5, 0, 0])
        lib.mjh_load_set_robot_gravcomp(1)
        m = ms.load_mjcf(paths=paths)
        np.testing.assert_allclose(m.array("body_gravcomp"), [0, 0.5, 1, 1])          # world-file bodies keep theirs
        d = orc.OrcData(m.ptr); d.step(200)
        assert abs(d.f("qpos")[7 + 2] - 0.5) < 1e-6 and d.f("qpos")[2] < 0.2          # the robot floats, the crate fell
        lib.mjh_load_set_robot_gravcomp(0)
        np.testing.assert_allclose(ms.load_mjcf(paths=paths).array("body_gravcomp"), [0, 0.5, 0, 0])
    finally:
        lib.mjh_load_set_robot_gravcomp(-1)


def test_odom_joints_added_to_robot_roots(tmp_path, lib):
    """~add_odom_joints (mj_sim.cpp:337-415): slide / hinge joints named after the robot's root body, appended to it"""
    import mujoco_sim_amd as ms
    (tmp_path / "world.xml").write_text('<mujoco><option gravity="0 0 0"/><worldbody><geom type="plane" size="0 0 0.05"/></worldbody></mujoco>')
    (tmp_path / "robot.xml").write_text('<mujoco><worldbody><body name="ridgeback" pos="0 0 0.3"><geom type="box" size="0.3 0.2 0.1"/>'
                                        '<body name="arm" pos="0 0 0.2"><joint name="shoulder" axis="0 1 0"/><geom type="capsule" size="0.03 0.2"/></body></body></worldbody></mujoco>')
    paths = [str(tmp_path / "world.xml"), str(tmp_path / "robot.xml")]
    try:
        lib.mjh_load_set_odom_joints(1 | 32)                      # lin x + ang z  ->  lin y comes along
        m = ms.load_mjcf(paths=paths)
    finally:
        lib.mjh_load_set_odom_joints(0)
    names = [lib.mjh_id2name(m.ptr, 1, j).decode() for j in range(m.njnt)]
    assert names == ["ridgeback_lin_odom_x_joint", "ridgeback_lin_odom_y_joint", "ridgeback_ang_odom_z_joint", "shoulder"]
    np.testing.assert_array_equal(m.array("jnt_type"), [2, 2, 3, 3])
    np.testing.assert_allclose(m.array("jnt_axis").reshape(-1, 3), [[1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 1, 0]])
    assert m.nv == 4 and ms.load_mjcf(paths=paths).nv == 1      # and nothing is added without the option
    d = orc.OrcData(m.ptr); d.f("qvel")[:] = [0.5, 0, 1.0, 0]; d.step(100)
    np.testing.assert_allclose(d.f("qpos")[[0, 2]], [0.1, 0.2], atol=1e-2)       # 100 steps of 2 ms: the base drives and turns on its odom joints


def test_robot_pose_init(tmp_path, lib):
    """~pose_init (mj_sim.cpp:312-335): pos + euler of a robot's root body, by name"""
    import mujoco_sim_amd as ms
    (tmp_path / "world.xml").write_text('<mujoco><worldbody><geom type="plane" size="0 0 0.05"/></worldbody></mujoco>')
    (tmp_path / "robot.xml").write_text('<mujoco><worldbody><body name="tiago" pos="9 9 9"><freejoint/><geom type="box" size="0.2 0.2 0.1"/></body></worldbody></mujoco>')
    paths = [str(tmp_path / "world.xml"), str(tmp_path / "robot.xml")]
    try:
        lib.mjh_load_set_robot_pose(b"tiago", D(1.0, 2.0, 0.5, 0, 0, np.pi / 2))
        m = ms.load_mjcf(paths=paths)
    finally:
        lib.mjh_load_set_robot_pose(None, None)
    np.testing.assert_allclose(m.array("qpos0"), [1, 2, 0.5, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)], atol=1e-12)
    # roll, pitch, yaw as tf2's setRPY composes them: R = Rz(yaw) Ry(pitch) Rx(roll)
    from scipy.spatial.transform import Rotation
    rpy = (0.3, -0.4, 1.1)
    lib.mjh_load_set_robot_pose(b"tiago", D(0, 0, 1, *rpy))
    try:
        w, x, y, z = ms.load_mjcf(paths=paths).array("qpos0")[3:7]
    finally:
        lib.mjh_load_set_robot_pose(None, None)
    np.testing.assert_allclose(Rotation.from_quat([x, y, z, w]).as_matrix(), Rotation.from_euler("xyz", rpy).as_matrix(), atol=1e-12)   # extrinsic xyz = fixed-axis RPY
    np.testing.assert_allclose(ms.load_mjcf(paths=paths).array("qpos0")[:3], [9, 9, 9])


def test_per_call_load_options_do_not_leak_into_later_loads(tmp_path):
    """mjh_load_mjcf_files_opt: the rosparams of MjSim::init_tmp as an argument of ONE load (boundmass / boundinertia, gravcomp on
    robot bodies, odom joints): a plain load afterwards sees the defaults again"""
    import ctypes as C
    import mujoco_sim_amd as ms
    from mujoco_sim_amd import capi
    lib = capi.load()
    world = tmp_path / "w.xml"; robot = tmp_path / "r.xml"
    world.write_text('<mujoco><worldbody><geom type="plane" size="0 0 0.05"/></worldbody></mujoco>')
    robot.write_text('<mujoco><worldbody><body name="base" pos="0 0 0.5"><joint name="j" type="hinge" axis="0 1 0"/>'
                     '<inertial pos="0 0 0" mass="1e-3" diaginertia="1e-4 1e-4 1e-4"/></body></worldbody></mujoco>')
    paths = (C.c_char_p * 2)(str(world).encode(), str(robot).encode())
    o = capi.LoadOptions(); lib.mjh_load_default_options(C.byref(o))
    o.boundmass = 1e-2; o.boundinertia = 1e-3; o.robot_gravcomp = 1; o.odom_joints = 0b100011          # lin x, lin y, yaw
    m = ms.Model(lib.mjh_load_mjcf_files_opt(paths, 2, C.byref(o)), lib)
    assert m.array("body_mass")[1] == 1e-2 and m.array("body_inertia")[3] == 1e-3 and m.array("body_gravcomp")[1] == 1 and m.njnt == 4
    assert m.name2id(1, "base_ang_odom_z_joint") >= 0
    m2 = ms.load_mjcf(paths=[str(world), str(robot)])                         # the thread's settings were left alone
    assert m2.array("body_mass")[1] == 1e-3 and m2.array("body_gravcomp")[1] == 0 and m2.njnt == 1


def test_parent_child_exclude_level(lib):
    """launch argument disable_parent_child_collision_level (mujoco_sim.launch:7; mujoco_compile.cpp:250-290): every body is excluded from
    colliding with its first `level` ancestors.  A chain of four overlapping spheres: level 0 keeps MuJoCo's own parent filter only
    (grandparent and great-grandparent pairs collide), level 2 leaves the pair three links apart, level 3 none."""
    import mujoco_sim_amd as ms
    xml = ('<mujoco><worldbody><body name="a" pos="0 0 1"><joint type="hinge" axis="0 1 0"/><geom type="sphere" size="0.2"/>'
           '<body name="b" pos="0.1 0 0"><joint type="hinge" axis="0 1 0"/><geom type="sphere" size="0.2"/>'
           '<body name="c" pos="0.1 0 0"><joint type="hinge" axis="0 1 0"/><geom type="sphere" size="0.2"/>'
           '<body name="d" pos="0.1 0 0"><joint type="hinge" axis="0 1 0"/><geom type="sphere" size="0.2"/>'
           '</body></body></body></body></worldbody></mujoco>')
    npair = {}
    try:
        for level in (0, 1, 2, 3):
            lib.mjh_load_set_parent_child_exclude(level)
            npair[level] = ms.load_mjcf(xml=xml).c.npair
    finally:
        lib.mjh_load_set_parent_child_exclude(0)
    assert npair == {0: 3, 1: 3, 2: 1, 3: 0}, npair           # 6 pairs, 3 adjacent ones always filtered; level 2 also a-c, b-d; level 3 also a-d
    assert ms.load_mjcf(xml=xml).c.npair == 3                   # (the setting was put back)
