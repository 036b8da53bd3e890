"""CRBA / RNE / FK / COM / energy of the oracle pinned by something that shares no algorithm with it or with the product:
tests/indep_dyn.py — numpy, fp64, from the models' input tables — builds M(q) term by term from world-frame Jacobians and the bias force as
projected Newton-Euler with the accelerations taken by finite differences along the motion.  At RANDOM configurations and velocities (the
existing check in test_oracle_pinning.py sits at qpos0 only), for the robots of the reference's own files (free base + hinges + slides:
pr2, tiago, hsrb4s, ridgeback_panda, armar6), the 7-hinge arm (C3), the ball-joint pendulum (C1 / C5) and S24's free boxes.  CPU only."""
import os

import numpy as np
import pytest

import mujoco_sim_amd as ms
import orc
from indep_dyn import Tree
from mujoco_sim_amd.tables import load_model_tables

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODELS = ["s24", "pendulum", "arm7", "pr2", "tiago", "hsrb4s", "ridgeback_panda", "armar6_mesh"]


def _model(name):
    if name in ("s24", "pendulum"):
        return ms.scene(name)
    if name == "arm7":
        return ms.scene("arm7", 1)
    return load_model_tables(os.path.join(G, f"robot_{name}.npz"))[0]


def _dense(m, qM):
    madr, dpar = m.array("dof_Madr"), m.array("dof_parentid")
    M = np.zeros((m.nv, m.nv))
    for i in range(m.nv):
        a = madr[i]; j = i
        while j >= 0:
            M[i, j] = M[j, i] = qM[a]; a += 1; j = dpar[j]
    return M


@pytest.mark.parametrize("name", MODELS)
def test_jacobians_of_the_independent_tree_are_the_derivatives_of_its_own_forward_kinematics(name):
    """self-check of the checker: the analytic world-frame Jacobians (joint axes and anchors) against central differences of the forward
    kinematics alone — for free / ball joints in the body-frame angular-velocity convention"""
    m = _model(name); T = Tree(m)
    rng = np.random.default_rng(11)
    q = T.random_configuration(rng)
    Jp, Jr, _, _ = T.jacobians(q)
    Jpf, Jrf = T.jacobians_fd(q)
    assert np.abs(Jp - Jpf).max() < 2e-8 and np.abs(Jr - Jrf).max() < 2e-8, (np.abs(Jp - Jpf).max(), np.abs(Jr - Jrf).max())


@pytest.mark.parametrize("name", MODELS)
def test_oracle_fk_crba_rne_energy_against_the_independent_tree_at_random_states(name):
    m = _model(name); T = Tree(m)
    rng = np.random.default_rng(7)
    g = np.array(m.opt.gravity[:], float)
    d = orc.OrcData(m.ptr); d.call("reset")
    gc = m.array("body_gravcomp")
    worst = {"M": 0.0, "bias": 0.0, "com": 0.0, "mulM": 0.0, "solveM": 0.0, "E": 0.0}
    for trial in range(3):
        q = T.random_configuration(rng)
        v = rng.normal(size=m.nv) * np.where(np.arange(m.nv) < 6, 0.5, 1.5)
        d.set_qpos(q, as_initial=False); d.f("qvel")[:] = v
        d.call("fwd_position"); d.call("fwd_velocity")
        # frames and centres of mass (mj_kinematics, mj_comPos)
        c, Rw, _ = T.com_frames(q)
        xipos = d.f("xipos").reshape(-1, 3); ximat = d.f("ximat").reshape(-1, 3, 3)
        worst["com"] = max(worst["com"], np.abs(xipos[1:] - c[1:]).max(), np.abs(ximat[1:] - Rw[1:]).max())
        # mass matrix (mj_crb) — every entry
        M = T.mass_matrix(q)
        Mo = _dense(m, d.f("qM"))
        sc = np.sqrt(np.outer(np.diag(M), np.diag(M)))
        worst["M"] = max(worst["M"], (np.abs(Mo - M) / sc).max())
        # mj_mulM (mj_sim.cpp:1057) and the L^T D L solve through the independent dense matrix
        x = rng.normal(size=m.nv)
        worst["mulM"] = max(worst["mulM"], np.abs(d.mul_m(x) - M @ x).max() / np.abs(M @ x).max())
        d.call("factor_m")
        worst["solveM"] = max(worst["solveM"], np.abs(d.solve_m(x) - np.linalg.solve(M, x)).max() / np.abs(np.linalg.solve(M, x)).max())
        # bias force (mj_rne with flg_acc = 0): Coriolis + centrifugal + gravity, projected Newton-Euler with finite-difference accelerations
        b = T.bias(q, v, g)
        bo = d.f("qfrc_bias").copy()
        worst["bias"] = max(worst["bias"], np.abs(bo - b).max() / max(1.0, np.abs(b).max()))
        # RNE with accelerations (what mj_inverse uses, mj_hw_interface.cpp:61): RNE(q, v, a) = M a + bias
        a = rng.normal(size=m.nv)
        d.f("qacc")[:] = a
        ra = d.rne(1)
        worst["bias"] = max(worst["bias"], np.abs(ra - (M @ a + b)).max() / max(1.0, np.abs(M @ a + b).max()))
        # energy (the reference displays it: mj_visual.cpp:176; flag on in world/empty.xml): potential and kinetic
        d.call("energy")
        ke, pe = T.energy(q, v, g)
        eo = d.f("energy")
        worst["E"] = max(worst["E"], abs(eo[1] - ke) / max(1.0, abs(ke)), abs(eo[0] - pe) / max(1.0, abs(pe)))
    print(f"INDEP-DYN {name}: nv {m.nv}: " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    assert worst["com"] < 1e-12 and worst["M"] < 1e-10 and worst["mulM"] < 1e-11 and worst["solveM"] < 1e-8
    assert worst["bias"] < 1e-8, worst         # central differences of step 1e-6 on accelerations of order |v|^2 (measured: 1e-12 .. 6e-10)
    assert worst["E"] < 1e-10
    # gravity compensation lives in qfrc_passive, never in the bias: with gravcomp bodies the check above already held
    assert gc.shape[0] == m.nbody


def test_passive_gravity_compensation_is_the_gravity_part_of_the_bias():
    """arm7 (gravcomp = 1 on every link, the wrapper's disable_gravity: robot.yaml:19): qfrc_passive's gravcomp term equals the gravity part
    of the independent bias force (bias at v = 0), so that passive - bias carries no gravity"""
    m = _model("arm7"); T = Tree(m)
    rng = np.random.default_rng(3)
    g = np.array(m.opt.gravity[:], float)
    d = orc.OrcData(m.ptr); d.call("reset")
    for _ in range(3):
        q = T.random_configuration(rng)
        d.set_qpos(q, as_initial=False); d.f("qvel")[:] = 0
        d.call("fwd_position"); d.call("fwd_velocity")
        grav = T.bias(q, np.zeros(m.nv), g)
        np.testing.assert_allclose(d.f("qfrc_passive"), grav, atol=1e-9 * max(1.0, np.abs(grav).max()))
        np.testing.assert_allclose(d.f("qfrc_bias"), grav, atol=1e-9 * max(1.0, np.abs(grav).max()))
