"""Known-answer tests that PIN the fp64 oracle (oracle/mjh_oracle.c).

The reference holds no numeric expectations for this path (its tests are assertion-free ROS
clients) and its arithmetic library (libmujoco 2.3.7) is absent, so the oracle is
"parity unpinned" with respect to the reference; these analytic cases are what
anchors it instead.  Everything here runs on the CPU."""
import numpy as np
import pytest

import mujoco_sim_amd as ms
import orc
from helpers import D, free_body_model, hinge_pendulum_model, set_opt, two_link_model

H = 0.005


def test_free_fall_exact(lib):
    """semi-implicit Euler: v_N = v0 + N h g ; z_N = z0 + N h v0 + h^2 g N(N+1)/2"""
    m = free_body_model(lib)
    d = orc.OrcData(m.ptr)
    d.f("qvel")[:3] = [1.0, 0.0, 2.0]
    N, g = 200, -9.81
    d.step(N)
    assert abs(d.f("qpos")[2] - (10 + N * H * 2 + H * H * g * N * (N + 1) / 2)) < 1e-11
    assert abs(d.f("qvel")[2] - (2 + N * H * g)) < 1e-12
    assert abs(d.f("qpos")[0] - N * H) < 1e-12
    assert abs(d.f("time")[0] - N * H) < 1e-12


def test_torque_free_rotation_spherical_inertia(lib):
    """constant body-frame omega; q_N = q_0 * exp(N h omega / 2)"""
    m = free_body_model(lib, gravity=[0, 0, 0])
    d = orc.OrcData(m.ptr)
    w = np.array([0.3, -0.5, 0.7])
    d.f("qvel")[3:6] = w
    N = 300
    d.step(N)
    ang = np.linalg.norm(w) * H * N
    qe = np.r_[np.cos(ang / 2), w / np.linalg.norm(w) * np.sin(ang / 2)]
    assert np.abs(d.f("qpos")[3:7] - qe).max() < 1e-12
    assert np.abs(d.f("qvel")[3:6] - w).max() < 1e-14


def test_asymmetric_body_conserves_angular_momentum(lib):
    m = free_body_model(lib, geom_type=6, size=(0.1, 0.2, 0.3), pos=(0, 0, 1), gravity=[0, 0, 0], timestep=0.0005)
    d = orc.OrcData(m.ptr)
    d.f("qvel")[3:6] = [1.0, 5.0, 0.5]
    I = m.array("body_inertia")[3:6]

    def Lw():
        return d.f("xmat")[9:18].reshape(3, 3) @ (I * d.f("qvel")[3:6])

    d.call("step1"); L0 = Lw().copy(); E0 = d.f("energy").sum(); d.call("step2")
    d.step(2000); d.call("step1")
    assert np.abs(Lw() - L0).max() / np.linalg.norm(L0) < 2e-3     # first-order integrator drift
    assert abs(d.f("energy").sum() - E0) / E0 < 5e-3


def test_damped_hinge_pendulum_recurrence(lib):
    """implicit damping: qd+ = qd + h (tau_g(q) - d qd)/(I + h d) ; q+ = q + h qd+"""
    mass, length, inertia, damp = 2.0, 1.0, 0.1, 0.5
    m = hinge_pendulum_model(lib, damping=damp, mass=mass, length=length, inertia=inertia)
    d = orc.OrcData(m.ptr)
    q, v = 0.7, 0.0
    d.f("qpos")[0] = q
    Iy = inertia + mass * length**2
    for _ in range(600):
        tau = -mass * 9.81 * length * np.sin(q)
        v = v + H * (tau - damp * v) / (Iy + H * damp)
        q = q + H * v
    d.step(600)
    assert abs(d.f("qpos")[0] - q) < 1e-12 and abs(d.f("qvel")[0] - v) < 1e-12


def test_two_link_mass_matrix_and_bias_closed_form(lib):
    l1, l2, m1, m2 = 1.0, 0.7, 1.5, 0.8
    m = two_link_model(lib, l1, l2, m1, m2)
    d = orc.OrcData(m.ptr)
    th = np.array([0.4, -0.9]); thd = np.array([1.3, -0.6])
    d.f("qpos")[:] = th; d.f("qvel")[:] = thd
    d.call("fwd_position"); d.call("fwd_velocity")
    c2 = np.cos(th[1])
    Mref = np.array([[m1 * l1**2 + m2 * (l1**2 + l2**2 + 2 * l1 * l2 * c2), m2 * (l2**2 + l1 * l2 * c2)],
                     [m2 * (l2**2 + l1 * l2 * c2), m2 * l2**2]])
    M = np.array([d.mul_m(np.eye(2)[i]) for i in range(2)])
    np.testing.assert_allclose(M, Mref, atol=1e-8)
    hh = m2 * l1 * l2 * np.sin(th[1])
    cor = np.array([-hh * (2 * thd[0] * thd[1] + thd[1] ** 2), hh * thd[0] ** 2])
    g = 9.81
    grav = np.array([g * (m1 * l1 * np.sin(th[0]) + m2 * (l1 * np.sin(th[0]) + l2 * np.sin(th.sum()))), g * m2 * l2 * np.sin(th.sum())])
    np.testing.assert_allclose(d.f("qfrc_bias"), cor + grav, atol=1e-8)


def test_dynamics_identities_on_arm(lib):
    m = ms.scene("arm7", 0)
    d = orc.OrcData(m.ptr)
    rng = np.random.default_rng(0)
    d.f("qpos")[:] = rng.uniform(-1, 1, 7); d.f("qpos")[3] = -1.5
    d.f("qvel")[:] = rng.uniform(-1, 1, 7)
    d.call("fwd_position"); d.call("fwd_velocity")
    x = rng.normal(size=7)
    np.testing.assert_allclose(d.mul_m(d.solve_m(x)), x, atol=1e-12)       # mulM o solveM = id
    a = rng.normal(size=7); d.f("qacc")[:] = a
    np.testing.assert_allclose(d.rne(1) - d.mul_m(a), d.f("qfrc_bias"), atol=1e-12)  # RNE(q,qd,qdd) - M qdd = bias
    M = np.array([d.mul_m(np.eye(7)[i]) for i in range(7)])
    assert np.abs(M - M.T).max() < 1e-14 and np.linalg.eigvalsh(M).min() > 0
    # gravity part of the bias = dV/dq (finite differences of the potential energy)
    q0 = d.f("qpos").copy(); d.f("qvel")[:] = 0
    d.call("fwd_position"); d.call("fwd_velocity"); bias = d.f("qfrc_bias").copy()
    gnum = np.zeros(7)
    for i in range(7):
        for s in (1e-6, -1e-6):
            d.f("qpos")[:] = q0; d.f("qpos")[i] += s
            d.call("kinematics"); d.call("com_pos"); d.call("energy")
            gnum[i] += np.sign(s) * d.f("energy")[0] / 2e-6
    np.testing.assert_allclose(bias, gnum, atol=1e-6)


def test_inverse_of_forward_round_trip(lib):
    """mj_inverse contract used by MjHWInterface::read (mj_hw_interface.cpp:61-69):
    qfrc_inverse(q, qd, qdd from forward(tau)) == tau, also with active limit rows"""
    m = ms.scene("arm7", 0)
    d = orc.OrcData(m.ptr)
    rng = np.random.default_rng(1)
    d.f("qpos")[:] = [0.1, 0.5, 0.1, -1.5, 0.2, 1.0, 0.1]
    d.f("qvel")[:] = rng.uniform(-0.5, 0.5, 7)
    tau = rng.normal(size=7) * 3
    d.call("fwd_position"); d.call("fwd_velocity")
    d.f("qfrc_applied")[:] = tau
    d.call("fwd_acceleration"); d.call("fwd_constraint")
    assert d.i("nefc") == 0
    d.call("inverse")
    np.testing.assert_allclose(d.f("qfrc_inverse"), tau, atol=1e-10)


def test_energy_conserved_by_undamped_double_pendulum(lib):
    m = two_link_model(lib)
    d = orc.OrcData(m.ptr)
    d.f("qpos")[:] = [1.0, 0.5]
    E = []
    for _ in range(2000):
        d.call("step1"); E.append(d.f("energy").sum()); d.call("step2")
    E = np.array(E)
    assert np.abs(E - E[0]).max() < 2e-2 * abs(E[0] - E.min() + 30)  # bounded first-order drift, no blow-up
    assert np.abs(E - E[0]).max() < 0.5


def _settle(d, n):
    d.step(n)
    return d


def test_sphere_rests_on_plane_with_weight_and_impedance_depth(lib):
    """at rest: sum of pyramid-row forces = m g, penetration follows the solref/solimp model"""
    r = 0.1
    m = free_body_model(lib, geom_type=2, size=(r, 0, 0), pos=(0, 0, r), floor=True)
    d = _settle(orc.OrcData(m.ptr), 600)
    mass = m.array("body_mass")[1]
    assert np.abs(d.f("qvel")).max() < 1e-5 and np.abs(d.f("qacc")[:3]).max() < 1e-4  # PGS tolerance 1e-8 leaves O(1e-6) creep
    assert d.i("ncon") == 1 and d.i("nefc") == 4
    f = d.f("efc_force")
    np.testing.assert_allclose(f.sum(), mass * 9.81, rtol=1e-5)
    np.testing.assert_allclose(d.f("qfrc_constraint")[2], mass * 9.81, rtol=1e-5)
    # closed-form depth: m g = 4 D K imp |dist| with D = 1/(2 mu^2 (1-imp)/imp * (1+mu^2)/m)
    dist = -d.contacts()[0]["dist"]
    mu, d0, dw, width, K = 1.0, 0.9, 0.95, 0.001, 1 / (0.95**2 * 0.02**2)
    x = min(dist / width, 1.0)
    y = 2 * x * x if x <= 0.5 else 1 - 2 * (1 - x) ** 2
    imp = d0 + y * (dw - d0)
    R = 2 * mu**2 * (1 - imp) / imp * (1 + mu**2) / mass
    np.testing.assert_allclose(4 / R * K * imp * dist, mass * 9.81, rtol=1e-4)
    assert 0 < dist < 2e-3   # penetration <= 2 mm (BASELINE.md invariant)


def test_box_rests_flat_with_four_corner_contacts(lib):
    hs = (0.1, 0.08, 0.05)
    m = free_body_model(lib, geom_type=6, size=hs, pos=(0, 0, hs[2]), floor=True)
    d = _settle(orc.OrcData(m.ptr), 600)
    mass = m.array("body_mass")[1]
    assert d.i("ncon") == 4 and d.i("nefc") == 16
    np.testing.assert_allclose(d.f("efc_force").sum(), mass * 9.81, rtol=1e-5)
    assert np.abs(d.f("qvel")).max() < 1e-5
    # orientation stays identity, equal load per corner
    np.testing.assert_allclose(d.f("qpos")[3:7], [1, 0, 0, 0], atol=1e-4)
    per_corner = d.f("efc_force").reshape(4, 4).sum(1)
    np.testing.assert_allclose(per_corner, mass * 9.81 / 4, rtol=1e-3)


def test_pgs_matches_an_independent_qp_solver(lib):
    """dual QP: min 0.5 f'AR f + b'f, f >= 0 — solved by scipy L-BFGS-B on the recorded (AR, b)"""
    from scipy.optimize import minimize

    m = ms.scene("s24")
    m.c.opt.iterations = 5000
    m.c.opt.tolerance = 0.0
    tab = m.s24_randomize(3, 1)
    from helpers import oracle_s24

    d = oracle_s24(m, tab, 0)
    d.step(250)
    d.call("step1"); d.call("fwd_acceleration"); d.call("fwd_constraint")
    n = d.i("nefc")
    assert n >= 20
    AR = d.f("efc_AR").reshape(n, n).copy(); b = d.f("efc_b").copy(); f = d.f("efc_force").copy()
    np.testing.assert_allclose(AR, AR.T, atol=1e-12)
    res = minimize(lambda x: 0.5 * x @ AR @ x + b @ x, np.zeros(n), jac=lambda x: AR @ x + b, bounds=[(0, None)] * n,
                   method="L-BFGS-B", options=dict(maxiter=20000, ftol=1e-16, gtol=1e-12))
    cost = lambda x: 0.5 * x @ AR @ x + b @ x
    assert cost(f) <= cost(res.x) + 1e-9 * abs(cost(res.x))
    # KKT: gradient >= 0 where f = 0, ~0 where f > 0
    g = AR @ f + b
    assert g[f == 0].min(initial=0) > -1e-7 and np.abs(g[f > 0]).max(initial=0) < 1e-6
    # primal map: qacc = qacc_smooth + M^-1 J^T f
    nv = m.nv
    J = d.f("efc_J").reshape(n, nv)
    np.testing.assert_allclose(d.f("qacc"), d.f("qacc_smooth") + d.solve_m(J.T @ f), atol=1e-9)


def test_joint_limit_holds_and_is_unilateral(lib):
    m = ms.scene("arm7", 1)   # gravcomp on: only the commanded torque moves joints
    d = orc.OrcData(m.ptr)
    q0 = np.array([0.0, 0.0, 0.0, -1.5, 0.0, 1.0, 0.0]); d.set_qpos(q0); d.call("reset")
    for _ in range(400):
        d.f("qfrc_applied")[:] = 0; d.f("qfrc_applied")[0] = 8.0   # push joint 1 into its upper limit 2.8973
        d.call("fwd_position"); d.call("fwd_velocity"); d.call("fwd_acceleration"); d.call("fwd_constraint"); d.call("euler")
    q = d.f("qpos")[0]
    assert 2.8973 < q < 2.8973 + 0.02, q          # soft limit: small violation, no run-away
    assert d.i("nefc") >= 1 and d.f("efc_force").min() >= 0


def test_controller_semantics(lib):
    """MjSim::controller (mj_sim.cpp:1055-1077): tau = M ddq + bias[controlled]; velocity override; commands consumed"""
    m = ms.scene("arm7", 0)
    d = orc.OrcData(m.ptr)
    d.set_qpos([0.1, 0.5, 0.1, -1.5, 0.2, 1.0, 0.1]); d.call("reset")
    d.ifield("controlled")[:] = [1, 1, 0, 1, 0, 0, 1]
    ddq = np.array([1.0, -2.0, 0.5, 0, 0, 3.0, 0]); dq = np.array([0, 0, 0, 0.25, 0, 0, 0])
    d.f("ddq")[:] = ddq; d.f("dq")[:] = dq
    d.call("step1")
    tau = d.mul_m(ddq) + d.f("qfrc_bias") * d.ifield("controlled")
    np.testing.assert_allclose(d.f("qfrc_applied"), tau, atol=1e-12)
    assert d.f("qvel")[3] == 0.25 and np.all(d.f("ddq") == 0) and np.all(d.f("dq") == 0)
    # with full computed torque on every dof and gravity only, qacc == ddq (no constraints active)
    d2 = orc.OrcData(m.ptr); d2.set_qpos([0.1, 0.5, 0.1, -1.5, 0.2, 1.0, 0.1]); d2.call("reset")
    d2.ifield("controlled")[:] = 1; d2.f("ddq")[:] = ddq
    d2.call("step1"); d2.call("fwd_acceleration"); d2.call("fwd_constraint")
    np.testing.assert_allclose(d2.f("qacc"), ddq, atol=1e-9)


def test_odom_velocity_rotation(lib):
    """MjSim::set_odom_vels (mj_sim.cpp:1079-1153): commanded twist rotated by the odom yaw"""
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.005, gravity=[0, 0, 0])
    bd = lib.mjh_builder_add_body(b, b"base", 0, D(0, 0, 0.5), None, 0.0)
    for nm, tp, ax in ((b"lx", 2, (1, 0, 0)), (b"ly", 2, (0, 1, 0)), (b"az", 3, (0, 0, 1))):
        lib.mjh_builder_add_joint(b, nm, bd, tp, None, D(*ax), None, 0, 0, 0, 0, 0)
    lib.mjh_builder_add_geom(b, b"g", bd, 6, D(0.2, 0.2, 0.1), None, None, None, -1, 0, 0, -1)
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    d = orc.OrcData(m.ptr)
    d.ifield("odom_lin")[:] = [0, 1, -1]; d.ifield("odom_ang")[:] = [-1, -1, 2]; d.ifield("odom_angq")[:] = [-1, -1, 2]
    d.f("odom_vel")[:] = [1.0, 0, 0, 0, 0, 0.5]
    d.f("qpos")[2] = np.pi / 2 - H * 0.0
    d.step(1)
    yaw = d.f("qpos")[2]
    np.testing.assert_allclose(d.f("qvel"), [np.cos(yaw), np.sin(yaw), 0.5], atol=1e-12)


def test_noslip_pass_removes_the_creep_of_soft_friction(lib):
    """option noslip_iterations (model/ontology/scene.xml:2-3): a box held by friction on a 0.3 rad incline creeps at
    ~1.5 mm/s under the regularised (soft) friction rows; the noslip sweeps (friction dimensions only, no regulariser)
    stop it, and they leave the normal forces (sum = m g cos) alone"""
    import mujoco_sim_amd as ms
    from helpers import D, set_opt
    out = {}
    for noslip in (0, 5):
        b = lib.mjh_builder_create(); set_opt(lib, b, timestep=0.005)
        o = ms.capi.Option(); lib.mjh_builder_get_option(b, o); o.noslip_iterations = noslip; lib.mjh_builder_set_option(b, o)
        ang = 0.3; tilt = D(np.cos(ang / 2), 0, np.sin(ang / 2), 0)
        lib.mjh_builder_add_geom(b, b"ramp", 0, 0, D(0, 0, 0.05), None, tilt, None, -1, -1, -1, -1)
        n = np.array([np.sin(ang), 0, np.cos(ang)])
        bd = lib.mjh_builder_add_body(b, b"box", 0, D(*(n * 0.0995)), tilt, 0.0)
        lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
        lib.mjh_builder_add_geom(b, None, bd, 6, D(0.1, 0.1, 0.1), None, None, None, -1, -1, -1, -1)
        m = ms.Model(lib.mjh_builder_compile(b), lib); lib.mjh_builder_destroy(b)
        assert m.opt.noslip_tolerance == 1e-6
        d = orc.OrcData(m.ptr); d.call("reset")
        d.step(400); p1 = d.f("qpos")[:3].copy(); d.step(400)
        f = d.f("efc_force")
        out[noslip] = (np.linalg.norm(d.f("qpos")[:3] - p1) / 2.0, f.sum(), d.i("ncon"))
    mg = 1000 * 0.008 * 9.81
    assert out[0][2] == out[5][2] == 4
    assert out[0][0] > 1e-3 and out[5][0] < 1e-5, out
    # pyramid edges: the sum of all edge forces is the normal force share, m g cos(angle) either way
    np.testing.assert_allclose(out[0][1], mg * np.cos(0.3), rtol=2e-3); np.testing.assert_allclose(out[5][1], mg * np.cos(0.3), rtol=2e-3)
