"""Round 5 GPU tests: what the window chain (csrc/window_pgs.h) owes the fused kernel besides the sweeps (odom velocity overwrite,
split-API statistics), the contact capacity of the small-free-body class beyond 64 contacts, and the solver check that bypasses the
oracle's own PGS."""
import os

import numpy as np
import pytest

import mujoco_sim_amd as ms
import orc
from helpers import oracle_s24

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _engine(m, nenv, window):
    lib = ms.capi.load()
    lib.mjh_set_window_solver(1 if window else 0)
    try:
        e = ms.Engine(m, nenv)
    finally:
        lib.mjh_set_window_solver(1)
    assert e.window_solver() == (1 if window else 0)
    return e


def test_odom_velocity_overwrite_in_the_window_chain(monkeypatch):
    """MjSim::set_odom_vels (mj_sim.cpp:1079-1153) runs behind mj_step2 whatever kernel integrates: a free box whose planar dofs are
    driven as odom dofs moves the same way through the window chain (assemble launch + mjh_window_kernel, the 16-row and the 32-row
    form) as through the fused kernel and the oracle (the window kernel used to skip the overwrite).  40 steps from
    reset: box 0 lands on the floor alone (the others are still falling), so the three trajectories are comparable to rounding."""
    m = ms.scene("s24")
    nenv = 8
    lin, ang, angq = [0, 1, -1], [-1, -1, 5], [-1, -1, -1]          # box 0: world x / y velocity and its body-z spin; no odom angle (free joints have none)
    twist = np.tile(np.array([[0.3, -0.2, 0, 0, 0, 0.7]]), (nenv, 1)) * np.linspace(0.5, 1.5, nenv)[:, None]
    outs = {}
    for name, window, w32 in (("window16", True, "0"), ("window32", True, "8"), ("fused", False, "0")):
        monkeypatch.setenv("MJH_WINDOW32", w32)
        e = _engine(m, nenv, window)
        tab = e.load_s24()
        e.set_odom(lin, ang, angq); e.set_odom_vel(twist)
        e.step(40)
        _, q, v, _ = e.get_state(); st = e.get_stats()
        e.step(110)
        _, q2, v2, _ = e.get_state()
        outs[name] = (q.copy(), v.copy(), st.copy(), q2.copy(), v2.copy())
        e.close()
    ref_q, ref_v = [], []
    for i in range(nenv):
        d = oracle_s24(m, tab, i)
        d.ifield("odom_lin")[:] = lin; d.ifield("odom_ang")[:] = ang; d.ifield("odom_angq")[:] = angq
        d.f("odom_vel")[:] = twist[i]
        d.step(40)
        ref_q.append(d.f("qpos").copy()); ref_v.append(d.f("qvel").copy())
    rq, rv = np.array(ref_q), np.array(ref_v)
    for name, (q, v, st, q2, v2) in outs.items():
        assert st[:, 0].min() >= 1, f"{name}: box 0 is on the floor, the sweeps ran"
        for vv in (v, v2):          # the overwritten dofs carry the command exactly, whatever the pile does later
            np.testing.assert_allclose(vv[:, [0, 1, 5]], twist[:, [0, 1, 5]], rtol=0, atol=1e-7, err_msg=name)
        np.testing.assert_allclose(q[:, :7], rq[:, :7], atol=1e-4, err_msg=name)            # the driven box against the oracle
        np.testing.assert_allclose(v[:, :6], rv[:, :6], atol=2e-3, err_msg=name)
        assert np.isfinite(q2).all()
    np.testing.assert_allclose(outs["window16"][0], outs["fused"][0], atol=1e-4)
    np.testing.assert_allclose(outs["window32"][0], outs["fused"][0], atol=1e-4)


def test_split_api_statistics_between_the_two_halves_are_this_steps():
    """mjh_step1 through the window chain hands every env over to mjh_step2; the contact / row counts mjh_get_stats returns between
    the two are those of THIS step's position stage, as after a plain mj_step1."""
    m = ms.scene("s24")
    nenv = 32
    e = _engine(m, nenv, True)
    e.load_s24()
    e.step(100)
    e.synchronize()
    _, q, v, w = e.get_state()
    f = _engine(m, nenv, False)
    f.load_s24()
    f.set_state(qpos=q, qvel=v, warmstart=w)
    # lift box 3 of every env out of contact: the counts of the next position stage differ from the last step's
    q2 = q.copy(); q2[:, 21 + 2] += 2.0
    e.set_state(qpos=q2); f.set_state(qpos=q2)
    before = e.get_stats()[:, :2].copy()
    e.step1(); f.step1()
    se, sf = e.get_stats(), f.get_stats()
    assert np.array_equal(se[:, :2], sf[:, :2]), "window hand-over and the fused step1 report the same counts"
    assert (se[:, 0] < before[:, 0]).any(), "and they are this step's (a box was lifted away)"
    e.step2(); f.step2()
    assert np.array_equal(e.get_stats()[:, :2], f.get_stats()[:, :2])
    e.close(); f.close()
