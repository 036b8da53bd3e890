"""GPU tests added in round 2: fp64 simulation time, read-only contact snapshots, the in-engine PD effort controller (all through
the C ABI, against the fp64 oracle where a physics result is involved)."""
import os

import numpy as np
import pytest

import mujoco_sim_amd as ms
import orc
from helpers import free_body_model, oracle_s24
from mujoco_sim_amd.engine import EP

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_simulation_time_is_fp64_and_equals_n_dt(lib):
    """d->time is mjtNum in the reference (ROS stamps, the 10 kHz gate, the RTF logic): per-env time is an fp64 accumulator,
    so after n steps it is n * dt to fp64 round-off — also late in a long run, where an fp32 clock would advance by
    0.0039 or 0.0078 instead of 0.005 (ulp(1e5 s) = 0.0078 in fp32)"""
    m = free_body_model(lib, geom_type=2, size=(0.1, 0.1, 0.1), pos=(0, 0, 10), floor=False, timestep=0.005, gravity=[0, 0, 0])
    e = ms.Engine(m, 3)
    e.step(2000)
    t = e.get_state()[0]
    assert np.all(np.abs(t - 2000 * 0.005) < 1e-9)
    # nine hours into a run
    e.set_state(time=np.array([32400.0, 1.0e5, 1.0e6]))
    e.step(1000)
    t = e.get_state()[0]
    np.testing.assert_allclose(t - np.array([32400.0, 1.0e5, 1.0e6]), 5.0, atol=2e-7)
    dts = []
    for _ in range(5):
        t0 = e.get_state()[0]; e.step(1); dts.append(e.get_state()[0] - t0)
    assert np.all(np.abs(np.array(dts) - 0.005) < 1e-9)          # every single step advances the clock by dt
    e.close()


def test_million_steps_keep_the_clock_exact(lib):
    """more than 1e6 steps of one tiny environment: time == n * dt (fp64 accumulation error ~1e-10 relative)"""
    m = free_body_model(lib, geom_type=2, size=(0.1, 0.1, 0.1), pos=(0, 0, 10), floor=False, timestep=0.005, gravity=[0, 0, 0])
    e = ms.Engine(m, 1)
    n = 1_000_100
    for _ in range(n // 20002 + 1):
        e.step(min(20002, n)); n -= min(20002, n)
        if n <= 0:
            break
    t = e.get_state()[0][0]
    assert abs(t - 1_000_100 * 0.005) < 1e-6, t
    e.close()


def test_get_contacts_is_read_only_between_step1_and_step2():
    """mjh_get_contacts is a snapshot: statistics, state, warm start, time and the step1 hand-over survive it, and the split
    step that it interrupts equals the uninterrupted one bit for bit"""
    m = ms.scene("s24")
    a = ms.Engine(m, 8); a.load_s24(); b = ms.Engine(m, 8); b.load_s24()
    a.step(150); b.step(150)
    st0 = a.get_stats().copy(); s0 = [x.copy() for x in a.get_state()]
    c = a.get_contacts(3)
    assert len(c["dist"]) == st0[3, 0] > 0
    assert np.array_equal(a.get_stats(), st0)
    for x, y in zip(a.get_state(), s0):
        assert np.array_equal(x, y)
    # in the middle of a split step
    a.step1(); b.step1()
    for env in range(8):
        a.get_contacts(env)
    a.step2(); b.step2()
    for x, y in zip(a.get_state(), b.get_state()):
        assert np.array_equal(x, y)
    assert np.array_equal(a.get_stats(), b.get_stats())
    a.close(); b.close()


def test_step_zero_then_split_api_keeps_env_order_valid():
    """mjh_step(0) on a fresh large engine must not mark the (zero-filled) launch order valid: a following split-API call
    would otherwise run every workgroup on env 0"""
    m = ms.scene("s24")
    a = ms.Engine(m, 1100); a.load_s24(); b = ms.Engine(m, 1100); b.load_s24()
    a.step(0)
    a.step1(); a.inverse(); a.step2()
    b.step1(); b.inverse(); b.step2()
    for x, y in zip(a.get_state(), b.get_state()):
        assert np.array_equal(x, y)
    q = a.get_state()[1]
    assert np.abs(q[5] - q[0]).max() > 0 and q[1099, 7 * 3 + 2] < 1.05 - 1e-4   # every env stepped (top box of the last env fell), not only env 0
    a.close(); b.close()


def test_in_engine_pd_controller_equals_host_pd_and_the_oracle():
    """C3's control law (ddq = Kp (q* - q) - Kd qd, Kp 200 / Kd 50 as model/ontology/box/box.yaml:8) run on
    the device equals the same law evaluated on the host through read() / write() every step, and the oracle driven the same way"""
    g = np.load(os.path.join(G, "arm7_golden.npz"))
    m = ms.scene("arm7", 1)
    nenv = 4
    rng = np.random.default_rng(5)
    lo, hi = m.array("jnt_range").reshape(-1, 2).T
    target = rng.uniform(lo, hi, size=(nenv, m.nv))
    dev = ms.Engine(m, nenv); host = ms.Engine(m, nenv)
    ds = []
    for e in (dev, host):
        e.set_initial_qpos(np.tile(g["q0"], (nenv, 1))); e.reset(); e.set_controlled_dofs(np.ones(7, dtype=np.int32))
    for i in range(nenv):
        d = orc.OrcData(m.ptr); d.set_qpos(g["q0"]); d.call("reset"); d.ifield("controlled")[:] = 1; ds.append(d)
    dev.set_pd_controller(200.0, 50.0); dev.set_pd_target(target)
    for s in range(1, 201):
        q, v, _ = host.get_joint_state()
        host.set_cmd(ddq=200.0 * (target - q) - 50.0 * v)
        host.step(1, True)
        for i, d in enumerate(ds):
            d.f("ddq")[:] = 200.0 * (target[i] - d.f("qpos")) - 50.0 * d.f("qvel")
            d.step(1, 1)
        if s % 50 == 0:
            dev.step(50, True)
            qd, vd, fd = dev.get_joint_state(); qh, vh, fh = host.get_joint_state()
            # host law: fp64 arithmetic on fp32 state, device law: fp32 — agreement to fp32 rounding of a 200 x gain
            np.testing.assert_allclose(qd, qh, atol=2e-4); np.testing.assert_allclose(vd, vh, atol=5e-3)
            for i, d in enumerate(ds):
                np.testing.assert_allclose(qd[i], d.f("qpos"), atol=1e-3)
                np.testing.assert_allclose(fd[i], d.f("qfrc_inverse"), rtol=5e-3, atol=0.1)
    assert np.abs(qd - target).mean() < 0.1           # the arms are converging on their (in-range) targets
    dev.set_pd_controller(0.0, 0.0)
    dev.close(); host.close()
