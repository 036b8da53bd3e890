"""CPU known-answer tests of the features added behind the B1 / F1 / F4 rows of SURVEY.md §8: d->xfrc_applied (mj_sim.cpp:499),
force / torque sensors at sites (mj_sim.cpp:973-1014, mj_ros.cpp:1933-1966), mocap bodies and <weld> / <connect> equalities (the
`~receive` mode, mj_sim.cpp:847-960) — oracle against closed forms, and the MJCF loader's handling of the elements."""
import numpy as np

import mujoco_sim_amd as ms
import orc
from helpers import D, set_opt


def _free_box(lib, gravity=(0, 0, -9.81), floor=False, ipos=(0, 0, 0)):
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.002, gravity=list(gravity))
    if floor:
        lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, -1, -1, -1, -1)
    bd = lib.mjh_builder_add_body(b, b"box", 0, D(0, 0, 1.0), None, 0.0)
    lib.mjh_builder_add_joint(b, b"free", bd, 0, None, None, None, 0, 0, 0, 0, 0)
    lib.mjh_builder_add_geom(b, b"g", bd, 6, D(0.1, 0.15, 0.2), D(*ipos), None, None, -1, -1, -1, -1)
    return b, bd


def _compile(lib, b):
    m = ms.Model(lib.mjh_builder_compile(b), lib); lib.mjh_builder_destroy(b)
    return m


def test_xfrc_applied_force_at_the_com_and_torque(lib):
    """a force m g upward at the centre of mass cancels gravity exactly (also for a body whose COM is off its frame origin);
    a world-frame torque gives alpha = I^-1 tau in the body frame"""
    b, bd = _free_box(lib, ipos=(0.05, -0.02, 0.03))
    m = _compile(lib, b)
    d = orc.OrcData(m.ptr); d.call("reset")
    mass = m.array("body_mass")[1]; I = m.array("body_inertia")[3:6]
    d.f("xfrc_applied")[6:9] = [0, 0, 9.81 * mass]
    d.call("forward")
    np.testing.assert_allclose(d.f("qacc"), 0, atol=1e-9)
    tau = np.array([0.3, -0.2, 0.5])
    d.f("xfrc_applied")[9:12] = tau
    d.call("forward")
    np.testing.assert_allclose(d.f("qacc")[3:6], tau / I, rtol=1e-9)           # identity orientation: body frame = world frame
    # the linear acceleration of the FRAME ORIGIN follows from alpha about the centre of mass: a_o = alpha x (o - com)
    np.testing.assert_allclose(d.f("qacc")[:3], np.cross(tau / I, -np.array([0.05, -0.02, 0.03])), atol=1e-9)


def _pendulum_with_sensors(lib, gravity=(0, 0, -9.81), site_quat=None):
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.002, gravity=list(gravity))
    bd = lib.mjh_builder_add_body(b, b"arm", 0, D(0, 0, 2), None, 0.0)
    lib.mjh_builder_add_joint(b, b"hinge", bd, 3, D(0, 0, 0), D(0, 1, 0), None, 0, 0, 0, 0, 0)
    lib.mjh_builder_set_inertial(b, bd, 2.0, D(0, 0, -0.7), None, D(0.05, 0.08, 0.03))
    s = lib.mjh_builder_add_site(b, b"root", bd, D(0, 0, 0), None if site_quat is None else D(*site_quat))
    lib.mjh_builder_add_sensor(b, b"f", 4, s); lib.mjh_builder_add_sensor(b, b"t", 5, s)
    return _compile(lib, b)


def test_force_torque_sensor_on_a_hinge_pendulum_closed_form(lib):
    """sensor at the hinge: the force the world exerts on the arm is m (a_com - g); the torque about the hinge point has no
    component along the (frictionless) hinge axis; values in the SITE frame (= body frame here)"""
    m = _pendulum_with_sensors(lib)
    assert (m.nsite, m.nsensor, m.nsensordata) == (1, 2, 6) and m.name2id(3, "root") == 0 and m.name2id(4, "t") == 1
    d = orc.OrcData(m.ptr); d.call("reset")
    th, w = 0.6, 1.3
    d.f("qpos")[0] = th; d.f("qvel")[0] = w
    d.call("forward")
    mass, l, Iyy = 2.0, 0.7, 0.08
    alpha = d.f("qacc")[0]
    np.testing.assert_allclose(alpha, -mass * 9.81 * l * np.sin(th) / (Iyy + mass * l * l), rtol=1e-9)   # rotation about +y by th
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    r = R @ np.array([0, 0, -l])                                             # hinge -> COM, world
    a_com = np.cross([0, alpha, 0], r) + np.cross([0, w, 0], np.cross([0, w, 0], r))
    f_world = mass * (a_com - np.array([0, 0, -9.81]))
    sd = d.f("sensordata")
    np.testing.assert_allclose(sd[0:3], R.T @ f_world, rtol=1e-9, atol=1e-9)
    assert abs(sd[4]) < 1e-9                                                   # no torque along the hinge axis (site y = world y)
    # angular momentum balance about the hinge point for the two other components
    Iw = R @ np.diag([0.05, 0.08, 0.03]) @ R.T
    om, al = np.array([0, w, 0]), np.array([0, alpha, 0])
    tq_world = Iw @ al + np.cross(om, Iw @ om) + np.cross(r, mass * a_com) - np.cross(r, mass * np.array([0, 0, -9.81]))
    np.testing.assert_allclose(sd[3:6], R.T @ tq_world, rtol=1e-8, atol=1e-9)


def test_sensor_values_rotate_with_the_site_frame(lib):
    q = np.array([np.cos(0.4), np.sin(0.4) * 0.6, 0, np.sin(0.4) * 0.8])
    a = _pendulum_with_sensors(lib); bq = _pendulum_with_sensors(lib, site_quat=q)
    out = []
    for m in (a, bq):
        d = orc.OrcData(m.ptr); d.call("reset"); d.f("qpos")[0] = 0.3; d.f("qvel")[0] = -0.5; d.call("forward")
        out.append(d.f("sensordata").copy())
    w, x, y, z = q
    R = np.array([[1 - 2*(y*y+z*z), 2*(x*y - w*z), 2*(x*z + w*y)], [2*(x*y + w*z), 1 - 2*(x*x+z*z), 2*(y*z - w*x)], [2*(x*z - w*y), 2*(y*z + w*x), 1 - 2*(x*x+y*y)]])
    np.testing.assert_allclose(out[1][:3], R.T @ out[0][:3], atol=1e-9); np.testing.assert_allclose(out[1][3:], R.T @ out[0][3:], atol=1e-9)


def test_sensor_sees_contact_forces_and_xfrc_as_external(lib):
    """two stacked links: a base box resting on the floor carries an upper body on a slide joint held by its limit; the sensor
    on the UPPER body reads that body's weight, whatever the floor contact does to the lower one; pulling the upper body with
    xfrc_applied reduces the reading by exactly the pull (external forces are not interaction forces)"""
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.002)
    lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, -1, -1, -1, -1)
    base = lib.mjh_builder_add_body(b, b"base", 0, D(0, 0, 0.0995), None, 0.0)
    lib.mjh_builder_add_joint(b, b"free", base, 0, None, None, None, 0, 0, 0, 0, 0)
    lib.mjh_builder_add_geom(b, b"bg", base, 6, D(0.2, 0.2, 0.1), None, None, None, -1, -1, -1, -1)
    up = lib.mjh_builder_add_body(b, b"upper", base, D(0, 0, 0.3), None, 0.0)
    lib.mjh_builder_add_joint(b, b"slide", up, 2, None, D(0, 0, 1), D(0.0, 0.5), 0, 0, 0, 0, 0)
    lib.mjh_builder_set_inertial(b, up, 1.5, D(0, 0, 0), None, D(0.01, 0.01, 0.01))
    s = lib.mjh_builder_add_site(b, b"s", up, D(0, 0, 0), None)
    lib.mjh_builder_add_sensor(b, b"f", 4, s)
    m = _compile(lib, b)
    d = orc.OrcData(m.ptr); d.call("reset")
    d.step(1500)
    assert d.i("ncon") == 4 and np.abs(d.f("qvel")).max() < 1e-3
    np.testing.assert_allclose(d.f("sensordata"), [0, 0, 1.5 * 9.81], atol=2e-2)
    d.f("xfrc_applied")[6*2 + 2] = 5.0                                          # 5 N upward on the upper body
    d.step(1500)
    np.testing.assert_allclose(d.f("sensordata"), [0, 0, 1.5 * 9.81 - 5.0], atol=2e-2)


def _welded_box(lib, connect=False, gravity=(0, 0, 0)):
    b, bd = _free_box(lib, gravity=gravity)
    ref = lib.mjh_builder_add_body(b, b"box_ref", 0, D(0, 0, 1.0), None, 0.0)
    lib.mjh_builder_add_geom(b, b"rg", ref, 6, D(0.1, 0.15, 0.2), None, None, None, -1, 0, 0, -1)
    assert lib.mjh_builder_set_mocap(b, ref) == 0
    if connect:
        assert lib.mjh_builder_add_eq_connect(b, bd, ref, D(0.1, 0.0, 0.2)) >= 0
    else:
        assert lib.mjh_builder_add_eq_weld(b, bd, ref, None, 0.9) >= 0           # as MjSim::init_references writes it (mj_sim.cpp:933-938)
    return _compile(lib, b)


def test_weld_to_a_mocap_body_pulls_the_body_to_the_mocap_pose(lib):
    """the `~receive` mechanism (mj_sim.cpp:847-960): body welded (torquescale 0.9) to its mocap clone follows d->mocap_pos /
    mocap_quat; 6 equality rows while welded; without gravity it converges onto the target, with gravity it hangs a hair below"""
    m = _welded_box(lib)
    assert m.nmocap == 1 and m.neq == 1 and m.array("body_mocapid").tolist() == [-1, -1, 0]
    d = orc.OrcData(m.ptr); d.call("reset")
    d.call("forward")
    assert d.i("nefc") == 6 and np.abs(d.f("efc_pos")).max() < 1e-12           # at qpos0 the weld is satisfied
    tgt_p = np.array([0.3, -0.2, 1.4]); ang = 0.7; ax = np.array([1, 2, 2]) / 3.0
    tgt_q = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
    d.f("mocap_pos")[:] = tgt_p; d.f("mocap_quat")[:] = tgt_q
    d.step(1500)
    np.testing.assert_allclose(d.f("qpos")[:3], tgt_p, atol=1e-4)
    assert 2 * np.arccos(min(1.0, abs(d.f("qpos")[3:7] @ tgt_q))) < 1e-3
    assert np.abs(d.f("qvel")).max() < 1e-3
    # with gravity: a stiff spring, not a rigid joint (solref 0.02 1): sag below the target of order g tc^2
    m2 = _welded_box(lib, gravity=(0, 0, -9.81))
    d2 = orc.OrcData(m2.ptr); d2.call("reset"); d2.step(1500)
    sag = 1.0 - d2.f("qpos")[2]
    assert 0 < sag < 5e-3


def test_connect_holds_the_anchor_like_a_ball_joint(lib):
    m = _welded_box(lib, connect=True, gravity=(0, 0, -9.81))
    d = orc.OrcData(m.ptr); d.call("reset"); d.f("qvel")[3:6] = [0.5, -0.3, 0.2]
    worst = 0.0
    for _ in range(600):
        d.step(1)
        R = d.f("xmat")[9:18].reshape(3, 3); p = d.f("xpos")[3:6] + R @ np.array([0.1, 0.0, 0.2])
        worst = max(worst, np.abs(p - (np.array([0, 0, 1.0]) + np.array([0.1, 0.0, 0.2]))).max())
    assert d.i("nefc") == 3 and worst < 5e-3                                     # the anchor stays put (soft constraint: millimetres)
    assert np.abs(d.f("qvel")[3:6]).max() > 1e-2                                  # while the body swings about it


MJCF = """<mujoco><option timestep="0.002" gravity="0 0 -9.81"/>
<worldbody>
  <geom name="floor" type="plane" size="0 0 0.05"/>
  <body name="arm" pos="0 0 2"><joint name="h" type="hinge" axis="0 1 0"/><inertial pos="0 0 -0.5" mass="1" diaginertia="0.1 0.1 0.1"/>
    <site name="wrist" pos="0 0 -0.5" quat="1 0 0 0"/></body>
  <body name="cube" pos="1 0 1"><freejoint/><geom type="box" size="0.1 0.1 0.1"/></body>
  <body name="cube_ref" pos="1 0 1" mocap="true"><geom type="box" size="0.1 0.1 0.1" contype="0" conaffinity="0"/></body>
</worldbody>
<equality><weld body1="cube" body2="cube_ref" torquescale="0.9"/><connect body1="arm" body2="cube" anchor="0 0 -1"/></equality>
<sensor><force name="wrist_force" site="wrist"/><torque site="wrist"/><accelerometer site="wrist"/></sensor>
</mujoco>"""


def test_mjcf_loader_reads_sites_sensors_mocap_weld_and_connect():
    m = ms.load_mjcf(xml=MJCF)
    assert (m.nsite, m.nsensor, m.nsensordata, m.nmocap, m.neq) == (1, 2, 6, 1, 2)
    assert m.array("eq_type").tolist() == [1, 0] and "accelerometer" in m.note
    assert m.name2id(4, "wrist_force") == 0 and m.array("sensor_type").tolist() == [4, 5]
    cube, ref, arm = m.name2id(0, "cube"), m.name2id(0, "cube_ref"), m.name2id(0, "arm")
    assert m.array("body_mocapid")[ref] == 0 and m.array("eq_obj1id").tolist() == [cube, arm] and m.array("eq_obj2id").tolist() == [ref, cube]
    dat = m.array("eq_data").reshape(-1, 11)
    np.testing.assert_allclose(dat[0], [0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0.9], atol=1e-12)       # the clone sits on the body: identity relpose
    np.testing.assert_allclose(dat[1][:6], [0, 0, -1, -1, 0, 0], atol=1e-12)                  # arm-frame anchor (0,0,1 world) seen from the cube at (1,0,1)
    assert m.maxefc >= 6 + 3
    d = orc.OrcData(m.ptr); d.call("reset"); d.step(50)
    assert d.i("nefc") >= 9 and np.isfinite(d.f("sensordata")).all()
