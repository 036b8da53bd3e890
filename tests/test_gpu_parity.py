"""GPU parity tests: the HIP stepper, called through the C ABI (include/mjhip.h), against the fp64
oracle on identical seeded inputs, against the committed golden fixtures, and — at the full
BASELINE size (4096 envs) — through size-independent invariants.

Tolerances (fp32 device vs fp64 oracle, stated per BASELINE.md §3):
  1 step            <= 1e-5 relative on qpos/qvel
  smooth scenes     <= 1e-3 abs after 400-1000 steps (C1 pendulum, C3 arm)
  contact scenes    <= 1e-2 abs after 100+ steps for the large majority of envs (a contact appearing
                    one step earlier/later in fp32 forks the trajectory — chaotic pile), plus invariants
"""
import os

import numpy as np
import pytest

import mujoco_sim_amd as ms
import orc
from conftest import ROOT
from helpers import D, oracle_s24, quat_angle, set_opt, two_link_model
from mujoco_sim_amd.engine import EP, MjhError

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def s24():
    m = ms.scene("s24")
    nenv = 16
    e = ms.Engine(m, nenv)
    tab = e.load_s24()
    ds = [oracle_s24(m, tab, i) for i in range(nenv)]
    yield m, e, tab, ds
    e.close()


def test_s24_stage_parity_after_forward(s24):
    m, e, tab, ds = s24
    e.reset(); [d.call("reset") for d in ds]
    [d.step(60) for d in ds]                    # get some contacts first
    # stage parity is measured from IDENTICAL states: load the oracle's step-60 state into the engine
    e.set_state(qpos=np.array([d.f("qpos") for d in ds]), qvel=np.array([d.f("qvel") for d in ds]),
                time=np.array([d.f("time")[0] for d in ds]), warmstart=np.array([d.f("qacc_warmstart") for d in ds]))
    e.forward(); e.synchronize()
    for d in ds:
        d.call("forward")
    xp, xq = e.get_body_state()
    np.testing.assert_allclose(xp.reshape(16, -1), [d.f("xpos") for d in ds], atol=2e-5)
    gp, gm = e.get_geom_state()
    np.testing.assert_allclose(gm.reshape(16, -1), [d.f("geom_xmat") for d in ds], atol=2e-5)
    st = e.get_stats()
    same = [i for i, d in enumerate(ds) if st[i, 0] == d.i("ncon") and st[i, 1] == d.i("nefc")]
    assert len(same) >= 14, "contact sets should agree for almost every env after 60 steps"
    bias = e.get_field("qfrc_bias"); asm = e.get_field("qacc_smooth"); qacc = e.get_field("qacc")
    for i in same:
        d = ds[i]
        np.testing.assert_allclose(bias[i], d.f("qfrc_bias"), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(asm[i], d.f("qacc_smooth"), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(qacc[i], d.f("qacc"), rtol=2e-3, atol=2e-2)     # solver output: PGS at fp32
        c = e.get_contacts(i); oc = d.contacts()
        assert len(oc) == len(c["dist"])
        if oc:
            np.testing.assert_allclose(c["dist"], [x["dist"] for x in oc], atol=1e-5)
            np.testing.assert_allclose(c["pos"], [x["pos"] for x in oc], atol=1e-4)
            np.testing.assert_allclose(c["frame"], [x["frame"] for x in oc], atol=1e-4)
            assert [tuple(g) for g in c["geom"]] == [x["geom"] for x in oc]


# floors of the free-running tests = what the MI355X measures (r04: printed by the tests below), so that a regression that forks one more
# environment fails (the budgets used to be 10 of 16 and 3 of 6)
# measured r04 (row order, window sweep): 16 / 16 / 16 of 16 envs keep the oracle's contact-set history over 1 / 60 / 150 steps with the
# windows swept one by one (errors 8.5e-8 / 7.2e-6 / 9.3e-6), 16 / 15 / 14 with the windows swept in pairs (8.5e-8 / 7.5e-6 / 1.1e-5: the
# same accuracy per step — teacher-forced 7.1e-6 either way — another rounding, and a contact that appears a step earlier forks a pile);
# the golden fixture's 6 of 6 at every mark.  One env of slack below what the shipped kernels measure.
FORK_FLOOR_60, FORK_FLOOR_150, FORK_FLOOR_GOLDEN = 14, 13, 5


def _free_run(e, ds, nsteps):
    """free-running device and oracle, one step at a time -> per step: qpos error per env, and whether the env's contact-set
    HISTORY still agrees (same ncon and nefc in every step so far).  A contact that appears one step earlier or later in fp32
    forks a chaotic pile; an env is excused from the long-horizon tolerance from the step its history differs, not before."""
    n = len(ds)
    same = np.ones(n, dtype=bool); err = np.zeros((nsteps, n)); hist = np.zeros((nsteps, n), dtype=bool)
    for k in range(nsteps):
        e.step(1); [d.step(1) for d in ds]
        _, q, v, _ = e.get_state(); st = e.get_stats()
        same &= (st[:, 0] == [d.i("ncon") for d in ds]) & (st[:, 1] == [d.i("nefc") for d in ds])
        err[k] = np.abs(q - np.array([d.f("qpos") for d in ds])).max(axis=1); hist[k] = same
    return err, hist


def test_s24_trajectory_parity(s24):
    """free-running from the reset state: 1 step <= 1e-5; 60 and 150 steps for EVERY env whose contact-set history agrees with the
    oracle's (no "n of 16" allowance); the envs that forked are counted and bounded.  The settled regime the bench times is
    covered step by step in tests/test_gpu_teacher_forced.py."""
    m, e, tab, ds = s24
    e.reset(); [d.call("reset") for d in ds]
    err, hist = _free_run(e, ds, 150)
    _, q, v, _ = e.get_state()
    print("S24 free run: agreeing envs after 1/60/150 steps", hist[0].sum(), hist[59].sum(), hist[149].sum(),
          "max err of agreeing envs", err[0][hist[0]].max(), err[59][hist[59]].max(), err[149][hist[149]].max() if hist[149].any() else None)
    assert hist[0].all() and err[0].max() <= 1e-5
    assert err[59][hist[59]].max() < 1e-4 and hist[59].sum() >= FORK_FLOOR_60
    assert hist[149].sum() >= FORK_FLOOR_150 and err[149][hist[149]].max() < 5e-4
    # the envs that forked are not let go: they stay inside the pen and finite (a fork moves a box by centimetres, a defect by metres) ...
    assert np.isfinite(err).all() and err.max() < 0.5, err.max(axis=0)
    # ... and a fork starts as ONE manifold changing by a point or two (a contact crossing its margin a step early, a clipped vertex
    # appearing), not as a different contact set
    for i in np.nonzero(~hist[149])[0]:
        k = int(np.argmin(hist[:, i]))
        assert err[max(k - 1, 0), i] < 5e-4, (i, k, err[max(k - 1, 0), i])
    st = e.get_stats()
    assert (st[:, 3] == 0).all() and all(d.i("warn") == 0 for d in ds)


def test_s24_against_golden_fixture():
    """the committed oracle trajectories (tests/golden/s24_golden.npz: qpos at the marks + the per-step contact / row counts):
    every env whose contact-set history matches the golden one is within tolerance at every mark"""
    g = np.load(os.path.join(G, "s24_golden.npz"))
    m = ms.scene("s24")
    e = ms.Engine(m, 6)
    for k in EP:
        e.set_env_param(k, g[f"tab_{k}"])
    e.set_initial_qpos(g["tab_qpos"]); e.reset()
    hist_ncon, hist_nefc = g["hist_ncon"], g["hist_nefc"]            # [150, 6]
    same = np.ones(6, dtype=bool)
    marks = {int(mk): tol for mk, tol in zip(g["marks"], (2e-6, 2e-5, 2e-4, 2e-3))}
    for k in range(1, int(g["marks"][-1]) + 1):
        e.step(1)
        st = e.get_stats()
        same &= (st[:, 0] == hist_ncon[k - 1]) & (st[:, 1] == hist_nefc[k - 1])
        if k in marks:
            t, q, v, _ = e.get_state()
            ref = np.array([g[f"env{i}_step{k}_qpos"] for i in range(6)])
            err = np.abs(q - ref).max(axis=1)
            print("S24 golden: mark", k, "agreeing envs", int(same.sum()), "max err of agreeing", err[same].max() if same.any() else None, "max err of all", err.max())
            assert same.sum() >= (6 if k <= 10 else FORK_FLOOR_GOLDEN), (k, same)
            assert (err[same] < marks[k]).all(), (k, err, same)
            assert np.isfinite(err).all() and err.max() < 0.5, (k, err)          # (forked envs: still a pile in the pen)
            np.testing.assert_allclose(t, k * 0.005, rtol=1e-5)
    e.close()


def test_split_api_equals_fused(s24):
    """mjh_step1 -> [mjh_inverse] -> mjh_step2 as three launches == one fused mjh_step launch"""
    m, e, tab, ds = s24
    for inv in (0, 1):
        e.reset(); e.step(80); e.step(1, inv)
        _, qa, va, wa = e.get_state(); fa = e.get_joint_state()[2]
        e.reset(); e.step(80)          # deterministic replay reproduces the same pre-step state (incl. qacc)
        e.step1()
        if inv:
            e.inverse()
        e.step2()
        _, qb, vb, wb = e.get_state()
        # not bitwise: the split path re-normalises the (already unit) quaternions once more per launch
        np.testing.assert_allclose(qa, qb, atol=2e-6); np.testing.assert_allclose(va, vb, atol=2e-4, rtol=1e-4)
        np.testing.assert_allclose(wa, wb, atol=5e-2, rtol=1e-3)
        if inv:
            np.testing.assert_allclose(fa, e.get_joint_state()[2], atol=5e-2, rtol=1e-3)


def test_deterministic_replay(s24):
    m, e, tab, ds = s24
    outs = []
    for _ in range(2):
        e.reset(); e.step(120); outs.append(e.get_state())
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_pendulum_c1_vs_oracle_and_golden():
    g = np.load(os.path.join(G, "pendulum_golden.npz"))
    m = ms.scene("pendulum")
    e = ms.Engine(m, 2)
    v0 = np.tile([0.3, 0, 0, 0, 0.3, 0, 0, 0, 0.3], (2, 1))
    e.set_state(qvel=v0)
    done = 0
    for mk, tol in ((1, 1e-6), (100, 1e-5), (400, 1e-4)):
        e.step(mk - done); done = mk
        _, q, v, _ = e.get_state()
        np.testing.assert_allclose(q[0], g[f"step{mk}_qpos"], atol=tol)
        np.testing.assert_allclose(v[0], g[f"step{mk}_qvel"], atol=tol)
        np.testing.assert_array_equal(q[0], q[1])
    e.step(600)     # 1000 steps total (smooth scenes, 1000 steps <= 1e-3 abs)
    d = orc.OrcData(m.ptr); d.f("qvel")[:] = v0[0]; d.step(1000)
    _, q, v, _ = e.get_state()
    if d.i("ncon") == 0:
        np.testing.assert_allclose(v[0], d.f("qvel"), atol=1e-3)
        np.testing.assert_allclose(q[0], d.f("qpos"), atol=1e-3)
    e.close()


def test_arm7_c3_controller_inverse_limits():
    """C3: MjHWInterface::write -> controller -> mj_inverse -> read, every step, vs oracle + golden"""
    g = np.load(os.path.join(G, "arm7_golden.npz"))
    m = ms.scene("arm7", 1)
    nenv = 4
    e = ms.Engine(m, nenv)
    e.set_initial_qpos(np.tile(g["q0"], (nenv, 1))); e.reset()
    e.set_controlled_dofs(np.ones(7, dtype=np.int32))
    d = orc.OrcData(m.ptr); d.set_qpos(g["q0"]); d.call("reset"); d.ifield("controlled")[:] = 1
    for s in range(1, 301):
        q, v, f = e.get_joint_state()                       # read()
        e.set_cmd(ddq=200.0 * (g["target"] - q) - 50.0 * v)  # write(): effort interface
        e.step(1, with_inverse=True)
        d.f("ddq")[:] = 200.0 * (g["target"] - d.f("qpos")) - 50.0 * d.f("qvel")
        d.step(1, 1)
        if s in (1, 50, 300):
            q, v, f = e.get_joint_state()
            tol = 1e-5 if s == 1 else 1e-3
            np.testing.assert_allclose(q[0], g[f"step{s}_qpos"], atol=tol)
            np.testing.assert_allclose(q[0], d.f("qpos"), atol=tol)
            np.testing.assert_allclose(f[0], d.f("qfrc_inverse"), rtol=2e-3, atol=5e-2)
            np.testing.assert_array_equal(q[0], q[3])
    e.close()


def test_host_cpp_simulate_loop_matches_python_loop():
    """csrc/host_sim.cpp: simulate() + MjhHWInterface (C++ mirrors of mj_main.cpp:76-164 / mj_hw_interface.cpp)
    drive the split step1 / inverse / step2 API; same closed loop as the Python loop above and the oracle"""
    import ctypes as C

    from mujoco_sim_amd import capi

    g = np.load(os.path.join(G, "arm7_golden.npz"))
    m = ms.scene("arm7", 1)
    e = ms.Engine(m, 2)
    e.set_initial_qpos(np.tile(g["q0"], (2, 1))); e.reset()
    q = np.zeros(7); f = np.zeros(7); rtf = C.c_double(0)
    tgt = np.ascontiguousarray(g["target"], dtype=np.float64)
    rc = capi.load().mjh_host_run_pd(e.h, 0, capi.dptr(tgt), 200.0, 50.0, 300, capi.dptr(q), capi.dptr(f), C.byref(rtf))
    assert rc == 0
    # the reference loop reads (mj_inverse) BEFORE the controller update of the same step, i.e. exactly the
    # sequence of tests/golden/make_golden.py::arm7
    np.testing.assert_allclose(q, g["step300_qpos"], atol=2e-3)
    assert rtf.value > 0
    e.close()


def test_velocity_command_override_and_limits():
    m = ms.scene("arm7", 1)
    e = ms.Engine(m, 1)
    q0 = np.array([[2.85, 0.0, 0.0, -1.5, 0.0, 1.0, 0.0]])
    e.set_initial_qpos(q0); e.reset()
    e.set_controlled_dofs(np.ones(7, dtype=np.int32))
    d = orc.OrcData(m.ptr); d.set_qpos(q0[0]); d.call("reset"); d.ifield("controlled")[:] = 1
    dq = np.zeros(7); dq[0] = 0.5      # velocity interface drives joint 1 into its upper limit (2.8973)
    for _ in range(100):
        e.set_cmd(dq=dq); d.f("dq")[:] = dq
        e.step(1, with_inverse=True); d.step(1, 1)
    q, v, f = e.get_joint_state()
    np.testing.assert_allclose(q[0], d.f("qpos"), atol=1e-3)
    assert q[0, 0] > 2.8973 - 1e-3 and e.get_stats()[0, 1] >= 1 and d.i("nefc") >= 1


def test_mulM_and_energy_vs_oracle():
    m = ms.scene("arm7", 0)
    e = ms.Engine(m, 3)
    rng = np.random.default_rng(3)
    q = rng.uniform(-1, 1, (3, 7)); q[:, 3] = -1.5; v = rng.uniform(-1, 1, (3, 7))
    e.set_state(qpos=q, qvel=v)
    vec = rng.normal(size=(3, 7))
    res = e.mulM(vec)
    e.forward(); en = e.get_field("energy")
    for i in range(3):
        d = orc.OrcData(m.ptr); d.f("qpos")[:] = q[i]; d.f("qvel")[:] = v[i]
        d.call("forward")
        np.testing.assert_allclose(res[i], d.mul_m(vec[i]), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(en[i], d.f("energy"), rtol=1e-4, atol=1e-3)
    e.close()


def test_two_link_energy_conservation_on_device(lib):
    m = two_link_model(lib)
    e = ms.Engine(m, 1)
    e.set_state(qpos=np.array([[1.0, 0.5]]))
    E = []
    for _ in range(40):
        e.forward(); E.append(e.get_field("energy")[0].sum()); e.step(50)
    E = np.array(E)
    assert np.abs(E - E[0]).max() < 0.5
    e.close()


def test_odom_velocities_on_device(lib):
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.005, gravity=[0, 0, 0])
    bd = lib.mjh_builder_add_body(b, b"base", 0, D(0, 0, 0.5), None, 0.0)
    for nm, tp, ax in ((b"lx", 2, (1, 0, 0)), (b"ly", 2, (0, 1, 0)), (b"az", 3, (0, 0, 1))):
        lib.mjh_builder_add_joint(b, nm, bd, tp, None, D(*ax), None, 0, 0, 0, 0, 0)
    lib.mjh_builder_add_geom(b, b"g", bd, 6, D(0.2, 0.2, 0.1), None, None, None, -1, 0, 0, -1)
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    e = ms.Engine(m, 2)
    e.set_odom([0, 1, -1], [-1, -1, 2], [-1, -1, 2])
    e.set_odom_vel(np.array([[1.0, 0, 0, 0, 0, 0.5], [0.5, 0.2, 0, 0, 0, -1.0]]))
    d = orc.OrcData(m.ptr)
    d.ifield("odom_lin")[:] = [0, 1, -1]; d.ifield("odom_ang")[:] = [-1, -1, 2]; d.ifield("odom_angq")[:] = [-1, -1, 2]
    d.f("odom_vel")[:] = [1.0, 0, 0, 0, 0, 0.5]
    e.step(200); d.step(200)
    _, q, v, _ = e.get_state()
    np.testing.assert_allclose(q[0], d.f("qpos"), atol=1e-4)
    np.testing.assert_allclose(v[0], d.f("qvel"), atol=1e-5)
    e.close()


def _compare_rollout(m, q0, nsteps_list, tols, v0=None, nenv=2):
    e = ms.Engine(m, nenv)
    e.set_initial_qpos(np.tile(q0, (nenv, 1))); e.reset()
    d = orc.OrcData(m.ptr); d.set_qpos(q0); d.call("reset")
    if v0 is not None:
        e.set_state(qvel=np.tile(v0, (nenv, 1))); d.f("qvel")[:] = v0
    done = 0
    for n, tol in zip(nsteps_list, tols):
        e.step(n - done); d.step(n - done); done = n
        _, q, v, _ = e.get_state()
        st = e.get_stats()
        assert st[0, 3] == 0 and d.i("warn") == 0
        np.testing.assert_allclose(q[0], d.f("qpos"), atol=tol, err_msg=f"step {n}")
        np.testing.assert_array_equal(q[0], q[1])
    out = (e.get_stats()[0].copy(), d.i("ncon"), d.i("nefc"))
    e.close()
    return out


def test_box_pile_nv48_single_block_sweep(lib, layout_policy):
    """8 free boxes (nv = 48 > 32): exercises the full-wave single-block PGS sweep (NROW = 4) and box-box stacks"""
    layout_policy(1)          # keep the pools in LDS (116 KB at full-manifold capacity)
    m = ms.scene("boxpile", 8)
    assert m.nv == 48
    q0 = m.array("qpos0").copy()
    rng = np.random.default_rng(7)
    for k in range(8):                       # drop them closer together so they collide with each other
        q0[7*k:7*k+2] *= 0.55; q0[7*k+2] = 0.12 + 0.22 * (k // 4) + 0.01 * k
        quat = rng.normal(size=4) * 0.15 + np.array([1, 0, 0, 0]); q0[7*k+3:7*k+7] = quat / np.linalg.norm(quat)
    st, ncon, nefc = _compare_rollout(m, q0, [1, 40, 120], [1e-5, 5e-4, 2e-2])
    assert ncon >= 8 and st[0] == ncon and st[1] == nefc


def test_box_pile_nv96_lds_resident_acceleration(lib):
    """16 free boxes (nv = 96 > 64): the running acceleration of the PGS sweep lives in LDS and every block gathers /
    scatters its <= 12 dofs (NROW = 8 path); capacities set by hand (the default is every pair at full manifold)"""
    m = ms.scene("boxpile", 16)
    assert m.nv == 96
    m.c.maxcon = 96; m.c.maxefc = 96 * 4
    assert lib.mjh_query_lds_bytes(m.ptr) <= 160 * 1024
    q0 = m.array("qpos0").copy()
    rng = np.random.default_rng(11)
    for k in range(16):                      # drop them closer together so they collide with each other
        q0[7*k:7*k+2] *= 0.55; q0[7*k+2] = 0.12 + 0.22 * (k // 8) + 0.005 * k
        quat = rng.normal(size=4) * 0.12 + np.array([1, 0, 0, 0]); q0[7*k+3:7*k+7] = quat / np.linalg.norm(quat)
    st, ncon, nefc = _compare_rollout(m, q0, [1, 40, 100], [1e-5, 5e-4, 2e-2])
    assert ncon >= 16 and st[0] == ncon and st[1] == nefc


def test_c2_64_box_lattice_many_body_pools(lib):
    """BASELINE config C2: 64 free boxes (nv 384, 2080 candidate pairs) released from the 4x4x4 lattice of
    mjh_scene_boxpile.  The contact / block / Jacobian pools of such models live in per-env global memory; capacities are
    set by hand (600 contacts).  A short horizon: the oracle's dense AR is (4 ncon)^2."""
    m = ms.scene("boxpile", 64)
    assert m.nv == 384 and m.npair == 2080
    m.c.maxcon = 600; m.c.maxefc = 600 * 4
    assert lib.mjh_query_lds_bytes(m.ptr) <= 160 * 1024
    q0 = m.array("qpos0").copy()
    rng = np.random.default_rng(5)
    for k in range(64):                      # tighter lattice, slightly tilted boxes: contacts within a few steps
        q0[7*k:7*k+2] *= 0.62; q0[7*k+2] = 0.095 + 0.19 * (k // 16) + 0.002 * (k % 16)
        quat = rng.normal(size=4) * 0.05 + np.array([1, 0, 0, 0]); q0[7*k+3:7*k+7] = quat / np.linalg.norm(quat)
    st, ncon, nefc = _compare_rollout(m, q0, [1, 12, 30], [1e-5, 2e-4, 5e-3])
    assert ncon >= 100 and st[0] == ncon and st[1] == nefc


def test_cylinders_on_the_plane(lib):
    """plane - cylinder narrow phase (wheels of the reference's robots): a standing, a lying (rolling) and a tilted
    free cylinder, device vs oracle"""
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.004)
    lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, -1, -1, -1, -1)
    specs = [(b"stand", (0.06, 0.10), (0.0, 0.0, 0.12), (1, 0, 0, 0)),
             (b"roll", (0.05, 0.08), (0.4, 0.0, 0.07), (np.cos(np.pi / 4), np.sin(np.pi / 4), 0, 0)),
             (b"tilt", (0.07, 0.05), (0.0, 0.4, 0.16), (np.cos(0.3), 0, np.sin(0.3), 0))]
    for name, size, pos, quat in specs:
        bd = lib.mjh_builder_add_body(b, name, 0, D(*pos), D(*quat), 0.0)
        lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
        lib.mjh_builder_add_geom(b, None, bd, 5, D(size[0], size[1], 0), None, None, None, -1, -1, -1, -1)
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    assert m.nv == 18 and m.npair == 6      # 3 plane-cylinder + 3 cylinder-cylinder (generic convex; they stay apart here)
    q0 = m.array("qpos0").copy()
    v0 = np.zeros(m.nv); v0[6] = 0.5; v0[6 + 4] = 0.0; v0[12 + 3] = 1.0      # push the lying one along x, spin the tilted one
    st, ncon, nefc = _compare_rollout(m, q0, [1, 50, 150], [1e-5, 1e-3, 3e-2], v0=v0)
    assert ncon >= 5 and st[0] == ncon


def test_mixed_primitives_scene(lib):
    """sphere / capsule / box free bodies on the plane and on each other: every narrow-phase routine on the device"""
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.005)
    lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, -1, -1, -1, -1)
    specs = [(b"s1", 2, (0.08, 0, 0), (0.0, 0.0, 0.30)), (b"s2", 2, (0.06, 0, 0), (0.02, 0.01, 0.52)),
             (b"c1", 3, (0.04, 0.12, 0), (0.25, 0.0, 0.10)), (b"c2", 3, (0.04, 0.10, 0), (0.27, 0.03, 0.30)),
             (b"b1", 6, (0.10, 0.08, 0.05), (0.0, 0.0, 0.06)), (b"s3", 2, (0.05, 0, 0), (0.27, 0.0, 0.45))]
    for name, gt, size, pos in specs:
        bd = lib.mjh_builder_add_body(b, name, 0, D(*pos), None, 0.0)
        lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
        lib.mjh_builder_add_geom(b, None, bd, gt, D(*size), None, None, None, -1, -1, -1, -1)
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    assert m.nv == 36
    q0 = m.array("qpos0").copy()
    q0[7*2+3:7*2+7] = [np.cos(0.7), 0, np.sin(0.7), 0]      # tilt the capsules
    q0[7*3+3:7*3+7] = [np.cos(0.5), np.sin(0.5), 0, 0]
    st, ncon, nefc = _compare_rollout(m, q0, [1, 60, 160], [1e-5, 1e-3, 3e-2])
    assert ncon >= 4


def test_mjcf_model_with_equality_limits_and_contacts():
    """MJCF-loaded model: joint equality (two joints of one tree), hinge limits, gravcomp, ball-vs-arm contacts"""
    from test_mjcf_loader import ARM

    m = ms.load_mjcf(ARM)
    q0 = m.array("qpos0").copy()
    q0[0] = 0.3; q0[1] = 0.7                    # violate the equality a little: it must pull the elbow back
    q0[2:5] = [0.05, 0.0, 1.6]                   # drop the ball onto the arm
    st, ncon, nefc = _compare_rollout(m, q0, [1, 80, 250], [1e-5, 2e-3, 3e-2])
    assert nefc >= 1


def test_spawn_destroy_as_slot_toggle(s24):
    """F2: a destroyed slot neither collides nor moves; re-spawned with a pose and a twist it falls again"""
    m, e, tab, ds = s24
    e.reset(); [d.call("reset") for d in ds]
    top = 4                                           # body id of box3 (top of the column)
    e.set_slot_active(top, False, env0=2, n=3)        # destroy in envs 2,3,4
    for i in (2, 3, 4):
        ds[i].L.orc_set_slot_mask(ds[i].d, 1 << top)
    e.step(150); [d.step(150) for d in ds[:6]]
    _, q, v, _ = e.get_state()
    for i in (2, 3, 4):
        np.testing.assert_allclose(q[i, 21:28], tab["qpos"][i, 21:28], atol=1e-6)     # parked where it was
        assert (v[i, 18:24] == 0).all()
    assert q[0, 23] < tab["qpos"][0, 23] - 0.2                                       # in untouched envs it fell
    err = np.abs(q[:6] - np.array([d.f("qpos") for d in ds[:6]])).max(axis=1)
    assert (err < 2e-2).sum() >= 5, err
    # spawn again: pose + twist, as the spawn service does (mj_ros.cpp:1406-1412)
    e.set_slot_active(top, True, env0=2, n=3)
    e.set_body_pose(3, top, [0.0, 0.0, 1.2], [1, 0, 0, 0], [0, 0, -1.0, 0, 0, 0])
    e.step(60)
    _, q2, _, _ = e.get_state()
    assert q2[3, 23] < 1.0
    e.set_slot_active(top, True); e.reset()
    for d in ds:
        d.L.orc_set_slot_mask(d.d, 0)


def test_reset_and_bad_state_recovery(s24):
    m, e, tab, ds = s24
    e.reset(); e.step(30)
    e.reset([3, 5])
    t, q, v, w = e.get_state()
    np.testing.assert_allclose(q[[3, 5]], tab["qpos"][[3, 5]], atol=1e-6)
    assert (v[[3, 5]] == 0).all() and (t[[3, 5]] == 0).all() and t[0] > 0
    bad = q.copy(); bad[7, 2] = np.nan
    e.set_state(qpos=bad)
    e.step(1)
    st = e.get_stats()
    assert st[7, 3] & 4 and np.isfinite(e.get_state()[1]).all()      # mj_checkPos-style auto reset
    e.reset()


def test_error_behaviour(s24):
    m, e, tab, ds = s24
    with pytest.raises(MjhError, match="before mjh_step1"):
        e.step2()
    with pytest.raises(MjhError, match="out of bounds"):
        e.get_state(env0=10, n=100)
    with pytest.raises(MjhError):
        e.get_field("no_such_field")


def test_export_state_device_matches_getters(s24):
    import torch

    m, e, tab, ds = s24
    e.reset(); e.step(25)
    buf = torch.zeros(e.nenv * e.state_stride, dtype=torch.float32, device="cuda")
    e.export_state_device(buf.data_ptr()); e.synchronize()
    torch.cuda.synchronize()
    out = buf.cpu().numpy().reshape(e.nenv, -1)
    t, q, v, _ = e.get_state()
    np.testing.assert_allclose(out[:, 0], t, rtol=1e-6)
    np.testing.assert_allclose(out[:, 1:1 + m.nq], q, atol=0); np.testing.assert_allclose(out[:, 1 + m.nq:], v, atol=0)


# ---------------------------------------------------------------- full BASELINE size: invariants
def test_s24_full_size_invariants():
    m = ms.scene("s24")
    nenv = 4096
    e = ms.Engine(m, nenv)
    tab = e.load_s24()
    e.step(400)                                   # settle
    e.forward(); E0 = e.get_field("energy").sum(axis=1)
    e.step(100)
    e.forward(); E1 = e.get_field("energy").sum(axis=1)
    t, q, v, _ = e.get_state()
    st = e.get_stats()
    assert np.isfinite(q).all() and np.isfinite(v).all()
    assert (st[:, 3] == 0).all(), "no contact/row overflow and no bad-state resets at the chosen capacities"
    pos = q.reshape(nenv, 4, 7)
    assert np.abs(pos[:, :, :2]).max() < 0.175 + 0.02 and pos[:, :, 2].min() > 0.03      # inside the pen, above the floor
    np.testing.assert_allclose(np.linalg.norm(pos[:, :, 3:], axis=-1), 1, atol=1e-5)
    # dissipative contacts: total energy does not increase after settling (tolerate fp32 noise)
    assert np.mean(E1 <= E0 + 1e-3 * np.abs(E0)) > 0.99
    # Newton on the vertical axis of every free box: sum_b f_constraint,z = sum_b m_b (a_z + g), moving or not;
    # ties together qacc, qfrc_constraint and the per-env masses at full size
    e.forward()
    fz = e.get_field("qfrc_constraint").reshape(nenv, 4, 6)[:, :, 2]
    az = e.get_field("qacc").reshape(nenv, 4, 6)[:, :, 2]
    mb = tab["body_mass"][:, 1:]
    lhs, rhs = fz.sum(axis=1), (mb * (az + 9.81)).sum(axis=1)
    wtot = 9.81 * mb.sum(axis=1)
    assert np.median(np.abs(lhs - rhs) / wtot) < 1e-4 and np.quantile(np.abs(lhs - rhs) / wtot, 0.99) < 1e-2
    # settled piles carry their weight: normal-force sum = weight within 1 % (BASELINE.md invariant)
    speed = np.abs(v).max(axis=1)
    rest = speed < 2e-2
    print(f"resting envs (max |qvel| < 2e-2): {rest.mean():.2%}; speed quantiles {np.quantile(speed, [0.1, 0.5, 0.9])}")
    if rest.sum() > 50:
        assert np.median(np.abs(lhs - wtot)[rest] / wtot[rest]) < 1e-2
    # the "~30 contact" claim is measured, not assumed
    print(f"S24 4096 envs: mean ncon {st[:,0].mean():.1f} max {st[:,0].max()}  mean nefc {st[:,1].mean():.1f} max {st[:,1].max()}  mean iter {st[:,2].mean():.1f}")
    assert 8 <= st[:, 0].mean() <= 48
    # penetration bound on a sample of envs
    worst = 0.0
    for i in range(0, nenv, 256):
        c = e.get_contacts(i)
        if len(c["dist"]):
            worst = min(worst, c["dist"].min())
    assert worst > -8e-3
    e.close()


@pytest.mark.gpu
def test_cohort_streams_do_not_change_results():
    """mjh_set_cohorts: the envs are stepped as 1, 2 or 3 cohorts on separate HIP streams (with longest-job-first
    dispatch inside each); the trajectory must be bitwise identical, also with reads/writes between the steps."""
    import mujoco_sim_amd as ms
    m = ms.scene("s24")
    nenv = 1536
    out = []
    for nc in (1, 2, 3):
        e = ms.Engine(m, nenv); e.load_s24()
        e.set_cohorts(nc)
        assert e.cohorts == nc
        e.step(40)
        t, q, v, w = e.get_state()                       # join
        cmd = np.zeros((nenv, m.nv)); cmd[:, 2] = 0.5
        e.set_cmd(ddq=cmd)                               # write between steps
        e.step(1); e.step(1); e.step(3)                  # consecutive calls stay forked
        st = e.get_stats()
        t2, q2, v2, w2 = e.get_state()
        out.append((q, v, q2, v2, w2, st))
        e.close()
    for k in (1, 2):
        for a, b in zip(out[0], out[k]):
            assert np.array_equal(a, b)


@pytest.fixture
def layout_policy(lib):
    """mjh_set_layout_policy for one test, restored to the default afterwards"""
    def set_policy(p):
        lib.mjh_set_layout_policy(p)
    yield set_policy
    lib.mjh_set_layout_policy(0)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [1, 2], ids=["lds-resident", "global-pools"])
@pytest.mark.parametrize("name", ["pr2", "tiago", "hsrb4s", "ridgeback_panda", "pr2_world", "hsrb4s_world", "pr2_mesh", "pr2_world_mesh",
                                  "c5_pendulum_bowl_mesh", "c4_pr2_world_objects_mesh", "tiago_mesh", "hsrb4s_mesh", "armar6_mesh", "ridgeback_panda_mesh"])
def test_reference_robot_models_match_oracle(name, layout, layout_policy):
    """C4-type articulated models (the reference's pr2 / tiago / hsrb4s test assets, compiled to table fixtures by
    tests/golden/make_robot_fixtures.py; meshes skipped): 32-49 dof single trees with equality constraints, joint
    limits, friction loss, damping.  nv <= 32 takes the dual-block sweep with general (non-diagonal) M, nv > 32 the
    single-block sweep.  The computed-torque wrapper (mj_sim.cpp:1055-1063) and mj_inverse run every step."""
    import mujoco_sim_amd as ms
    from helpers import load_model_tables
    from test_robot_fixtures import robot_command
    m, z = load_model_tables(os.path.join(ROOT, "tests", "golden", f"robot_{name}.npz"))
    KEEP = [int(k) for k in z["keep"]]     # the *_mesh fixtures (PR2 with its 37 convex-hull mesh geoms) keep a short horizon
    nenv = 4
    layout_policy(layout)     # both memory layouts: pools in LDS (dual / single-block sweeps) and in global memory (three-launch step)
    e = ms.Engine(m, nenv)
    e.set_controlled_dofs(z["controlled"].astype(np.int32))
    d = orc.OrcData(m.ptr); d.ifield("controlled")[:] = z["controlled"]
    d.f("qvel")[:] = z["qvel0"]; e.set_state(qvel=np.tile(z["qvel0"], (nenv, 1)))
    for k in range(1, KEEP[-1] + 1):
        cmd = robot_command(m, k)
        e.set_cmd(ddq=np.tile(cmd, (nenv, 1)))
        e.step(1, True)
        d.f("ddq")[:] = cmd; d.step(1, 1)
        if k in KEEP:
            t, q, v, w = e.get_state()
            # all envs identical (same inputs), and equal to the committed golden / the live oracle
            assert np.array_equal(q[0], q[-1]) and np.array_equal(v[0], v[-1])
            np.testing.assert_allclose(d.f("qpos"), z[f"qpos_{k}"], atol=1e-9)          # oracle == golden
            # friction loss and grazing contacts make these trajectories fork over long horizons, so the engine is
            # re-synchronised with the oracle after every check and every 50 steps: each segment (<= 50 steps) starts from identical states
            # (tiago / hsrb4s with meshes: hulls that overlap by centimetres for good — portal refinement is
            # ill-conditioned there, fp32 and fp64 pick contact points millimetres apart)
            tol = 5e-3 if name in ("tiago_mesh", "hsrb4s_mesh", "armar6_mesh") else 4e-4
            np.testing.assert_allclose(q[0], z[f"qpos_{k}"], rtol=0, atol=tol * max(1.0, np.abs(z[f"qpos_{k}"]).max()))
            np.testing.assert_allclose(v[0], z[f"qvel_{k}"], rtol=0, atol=10 * tol * max(1.0, np.abs(z[f"qvel_{k}"]).max()))
            fi = e.get_field("qfrc_inverse")[0]
            ref = z[f"qfrc_inverse_{k}"]
            np.testing.assert_allclose(fi, ref, rtol=0, atol=5e-3 * max(1.0, np.abs(ref).max()))
            st = e.get_stats()
            assert (st[:, 3] == 0).all()
            # tiago's gripper fingers are boxes that touch at exactly zero distance (|dist| ~ 1e-8): whether such a
            # grazing pair counts as a contact is a rounding decision (its force is ~0 either way), so the row count is
            # only compared when every oracle contact is a real penetration
            if all(c["dist"] < -1e-6 for c in d.contacts()):
                assert st[0, 1] == int(z[f"nefc_{k}"])
        if k in KEEP or k % 50 == 0:
            # (also between two checks that are 100 steps apart: tiago's segment 200 -> 300 amplifies a rounding-level change of
            #  the tree solves a hundredfold — 3.8e-4 with dof-by-dof solves, 7.7e-4 level by level, 3e-6 at step 200 for both)
            e.set_state(qpos=np.tile(d.f("qpos"), (nenv, 1)), qvel=np.tile(d.f("qvel"), (nenv, 1)),
                        warmstart=np.tile(d.f("qacc_warmstart"), (nenv, 1)))
    e.close()


@pytest.mark.gpu
def test_split_api_large_batch_uses_launch_order_and_matches_fused():
    """Above 1024 envs the split entry points (mjh_step1 | mjh_inverse | mjh_step2, the reference loop's calls) also
    dispatch the envs longest-job-first; exports are indexed by env, not by launch position."""
    import mujoco_sim_amd as ms
    m = ms.scene("s24")
    nenv = 1280
    a = ms.Engine(m, nenv); a.load_s24(); b = ms.Engine(m, nenv); b.load_s24()
    a.step(60); b.step(60)
    ta, qa, va, wa = a.get_state(); tb, qb, vb, wb = b.get_state()
    assert np.array_equal(qa, qb)
    same = np.ones(nenv, dtype=bool)
    for _ in range(3):
        a.step(1, True)
        b.step1(); b.inverse(); b.step2()       # (step1 + inverse go out as ONE launch: mjh_step1 defers to the next entry point)
        sa, sb = a.get_stats(), b.get_stats()
        same &= (sa[:, 0] == sb[:, 0]) & (sa[:, 1] == sb[:, 1])
    ta, qa, va, wa = a.get_state(); tb, qb, vb, wb = b.get_state()
    # the two paths differ by an ulp in the stored quaternions (1.8e-7 after one step, measured); an env in which that moves a
    # contact across its margin forks — it is excused from the step its contact set differs (1 of 1280 here), no other
    assert same.mean() >= 0.995, same.mean()
    np.testing.assert_allclose(qa[same], qb[same], rtol=0, atol=2e-5)
    np.testing.assert_allclose(va[same], vb[same], rtol=0, atol=2e-3)
    fa = a.get_field("qfrc_inverse"); fb = b.get_field("qfrc_inverse")
    np.testing.assert_allclose(fa[same], fb[same], rtol=0, atol=2e-2 * max(1.0, np.abs(fa).max()))
    # per-env export rows (bias force = gravity on free boxes: m g on the z dof of every box, whatever the launch order)
    b.forward()
    bias = b.get_field("qfrc_bias")
    tab = m.s24_randomize(0, nenv)
    np.testing.assert_allclose(bias[:, 2], 9.81 * tab["body_mass"][:, 1], rtol=1e-5)
    a.close(); b.close()


@pytest.mark.gpu
def test_capacity_saturation_matches_oracle_drop_rule():
    """Contact / row capacities far below what the scene produces: the LDS pools are sized exactly for the capacities,
    contacts beyond maxcon are dropped in pair order, a contact whose rows do not fit drops it and every later one
    (same rule in the oracle), the overflow flags are raised and the step stays finite and in parity."""
    import mujoco_sim_amd as ms
    m = ms.scene("s24")
    nenv = 64
    full = ms.Engine(m, nenv); tab = full.load_s24(); full.step(300)
    t, q, v, w = full.get_state(); st_full = full.get_stats(); full.close()
    assert st_full[:, 0].max() > 14
    m.c.maxcon = 10; m.c.maxefc = 46          # 10 contacts, but only 46 rows: condim-4 floor contacts use 6 each
    e = ms.Engine(m, nenv)
    for k in EP:
        e.set_env_param(k, tab[k])
    e.set_initial_qpos(tab["qpos"])
    e.set_state(qpos=q, qvel=v, warmstart=w)
    e.step(1)
    st = e.get_stats()
    assert (st[:, 0] <= 10).all() and (st[:, 1] <= 46).all()
    assert (st[:, 3] & 3).any(), "the scene must overflow these capacities"
    t1, q1, v1, w1 = e.get_state()
    assert np.isfinite(q1).all() and np.isfinite(v1).all()
    for i in (0, 7, 21, 40, 63):
        d = oracle_s24(m, tab, i)
        d.set_qpos(q[i]); d.f("qvel")[:] = v[i]; d.f("qacc_warmstart")[:] = w[i]
        d.step()
        assert d.i("ncon") == st[i, 0] and d.i("nefc") == st[i, 1], (i, d.i("ncon"), d.i("nefc"), st[i])
        np.testing.assert_allclose(q1[i], d.f("qpos"), rtol=0, atol=5e-6)
        np.testing.assert_allclose(v1[i], d.f("qvel"), rtol=0, atol=2e-3)
    e.step(50)
    assert np.isfinite(e.get_state()[1]).all()
    e.close()


@pytest.mark.gpu
def test_forked_export_publishes_without_joining_the_cohorts():
    """mjh_export_state_device while the cohorts are forked: each cohort exports its range on its own stream (the
    step pipeline is not drained); the published buffer must equal the state after exactly the steps queued so far."""
    import torch
    m = ms.scene("s24")
    nenv = 2048
    a = ms.Engine(m, nenv); a.load_s24(); a.set_cohorts(2)
    b = ms.Engine(m, nenv); b.load_s24(); b.set_cohorts(1)
    bufs = [torch.zeros(nenv * a.state_stride, dtype=torch.float32, device="cuda") for _ in range(3)]
    for k in range(3):
        a.step(7)
        a.export_state_device(bufs[k].data_ptr())      # forked: no join
    a.step(5)
    a.synchronize(); torch.cuda.synchronize()
    for k in range(3):
        b.step(7)
        t, q, v, _ = b.get_state()
        out = bufs[k].cpu().numpy().reshape(nenv, -1)
        assert np.array_equal(out[:, 1:1 + m.nq], q.astype(np.float32)) and np.array_equal(out[:, 1 + m.nq:], v.astype(np.float32))
    b.step(5)
    assert np.array_equal(a.get_state()[1], b.get_state()[1])
    a.close(); b.close()


@pytest.mark.gpu
def test_pr2_with_large_capacity_falls_back_to_global_pools(lib, layout_policy):
    """A working set beyond one CU's LDS (PR2 at a 128-contact capacity: 49-dof rows) selects the many-body layout:
    pools in per-env global memory, block-at-a-time sweep with M^-1 J^T rows, equality and limit rows.  Same golden."""
    from helpers import load_model_tables
    from test_robot_fixtures import robot_command
    m, z = load_model_tables(os.path.join(ROOT, "tests", "golden", "robot_pr2.npz"))
    layout_policy(1)                                         # "LDS whenever it fits": at this capacity it does not
    small = lib.mjh_query_lds_bytes(m.ptr)
    m.c.maxcon = 128; m.c.maxefc = 6 * 128 + 200
    assert lib.mjh_query_lds_bytes(m.ptr) < small            # pools left LDS
    e = ms.Engine(m, 2)
    e.set_controlled_dofs(z["controlled"].astype(np.int32))
    for k in range(1, 101):
        e.set_cmd(ddq=np.tile(robot_command(m, k), (2, 1)))
        e.step(1, True)
        if k in (1, 10, 50, 100):
            t, q, v, w = e.get_state()
            np.testing.assert_allclose(q[0], z[f"qpos_{k}"], rtol=0, atol=4e-4 * max(1.0, np.abs(z[f"qpos_{k}"]).max()))
            ref = z[f"qfrc_inverse_{k}"]
            np.testing.assert_allclose(e.get_field("qfrc_inverse")[0], ref, rtol=0, atol=5e-3 * max(1.0, np.abs(ref).max()))
            assert e.get_stats()[0, 1] == int(z[f"nefc_{k}"])
    e.close()


@pytest.mark.gpu
def test_many_body_layout_keeps_environments_apart():
    """Global-pool layout with heterogeneous environments (different commands and initial joint positions per env), enough of
    them for cohorts and longest-job-first dispatch to be active: every sampled env must follow ITS oracle."""
    from helpers import load_model_tables
    from test_robot_fixtures import robot_command
    m, z = load_model_tables(os.path.join(ROOT, "tests", "golden", "robot_hsrb4s_world.npz"))
    nenv = 1100
    e = ms.Engine(m, nenv)
    assert e.cohorts >= 2        # (three for the three-launch layout)
    e.set_controlled_dofs(z["controlled"].astype(np.int32))
    rng = np.random.default_rng(3)
    scale = rng.uniform(0.2, 1.5, nenv)
    q0 = np.tile(m.array("qpos0"), (nenv, 1))
    jt = m.array("jnt_type"); qa = m.array("jnt_qposadr")
    hinge = [int(qa[j]) for j in range(m.njnt) if jt[j] == 3]
    q0[:, hinge[-6:]] += rng.uniform(-0.05, 0.05, (nenv, 6))
    e.set_initial_qpos(q0); e.reset()
    sample = [0, 1, 547, 548, 1099]
    ds = []
    for i in sample:
        d = orc.OrcData(m.ptr); d.set_qpos(q0[i]); d.call("reset"); d.ifield("controlled")[:] = z["controlled"]; ds.append(d)
    for k in range(1, 61):
        base = robot_command(m, k)
        e.set_cmd(ddq=scale[:, None] * base[None, :])
        e.step(1, True)
        for i, d in zip(sample, ds):
            d.f("ddq")[:] = scale[i] * base; d.step(1, 1)
    t, q, v, w = e.get_state()
    for i, d in zip(sample, ds):
        np.testing.assert_allclose(q[i], d.f("qpos"), rtol=0, atol=5e-4 * max(1.0, np.abs(d.f("qpos")).max()), err_msg=f"env {i}")
    assert not np.allclose(q[0], q[1], atol=1e-4)         # the envs really differ
    assert (e.get_stats()[:, 3] == 0).all()
    e.close()


@pytest.mark.gpu
def test_set_timestep_takes_effect_with_the_next_launch(lib):
    """m->opt.timestep is mutable in the reference (adaptive dt of simulate(), mj_main.cpp:150-163): free fall integrated
    with dt, then 2 dt, then dt again equals the semi-implicit Euler sums."""
    from helpers import free_body_model
    m = free_body_model(lib, geom_type=2, size=(0.1, 0.1, 0.1), pos=(0, 0, 10), floor=False, timestep=0.002)
    e = ms.Engine(m, 3)
    g = -9.81; z = 10.0; vz = 0.0; t = 0.0
    for dt, n in ((0.002, 40), (0.004, 25), (0.002, 10)):
        e.set_timestep(dt)
        assert abs(e.timestep - dt) < 1e-9
        e.step(n)
        for _ in range(n):
            vz += dt * g; z += dt * vz; t += dt
    tt, q, v, w = e.get_state()
    np.testing.assert_allclose(q[:, 2], z, rtol=2e-6)
    np.testing.assert_allclose(v[:, 2], vz, rtol=2e-6)
    np.testing.assert_allclose(tt, t, rtol=1e-5)
    with pytest.raises(MjhError):
        e.set_timestep(0.0)
    e.close()


@pytest.mark.gpu
def test_pinned_host_mirror_matches_getters(s24, lib):
    """F3: the asynchronous pinned mirror of an env range (joint state, body poses, geom poses) equals the blocking getters"""
    import ctypes as C
    m, e, tab, ds = s24
    e.reset(); e.step(30, True)
    env0, n = 2, 5
    h = C.c_void_p()
    assert lib.mjh_mirror_create(e.h, env0, n, C.byref(h)) == 0
    try:
        assert lib.mjh_mirror_update(h, 1 | 2 | 4) == 0
        e.step(3)                                   # more work queued behind the refresh: the mirror keeps the earlier state
        assert lib.mjh_mirror_wait(h) == 0
        def field(which):
            w = C.c_int(0)
            p = lib.mjh_mirror_field(h, which, C.byref(w))
            return np.ctypeslib.as_array(p, shape=(n, w.value)).copy()
        got = {k: field(k) for k in range(8)}
    finally:
        lib.mjh_mirror_destroy(h)
    f = ms.Engine(m, e.nenv); 
    for k in EP: f.set_env_param(k, tab[k])
    f.set_initial_qpos(tab["qpos"]); f.reset(); f.step(30, True)
    t, q, v, w = f.get_state(env0, n); qj, vj, fi = f.get_joint_state(env0, n)
    xp, xq = f.get_body_state(env0, n); gp, gm = f.get_geom_state(env0, n)
    f.close()
    assert np.array_equal(got[0].view(np.float64)[:, 0], t)          # field 0: one fp64 per env (also mjh_mirror_time)
    assert np.array_equal(got[1], q.astype(np.float32)) and np.array_equal(got[2], v.astype(np.float32))
    assert np.array_equal(got[3], fi.astype(np.float32))
    np.testing.assert_allclose(got[4], xp.reshape(n, -1), atol=1e-6); np.testing.assert_allclose(got[5], xq.reshape(n, -1), atol=1e-6)
    np.testing.assert_allclose(got[6], gp.reshape(n, -1), atol=1e-6); np.testing.assert_allclose(got[7], gm.reshape(n, -1), atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("nenv", [1, 63, 1025, 2051])
def test_odd_batch_sizes_with_cohorts_and_launch_order(nenv):
    """uneven cohort splits, env counts that are not multiples of the wave / bucket sizes: bitwise equal to one cohort"""
    m = ms.scene("s24")
    a = ms.Engine(m, nenv); a.load_s24(); a.set_cohorts(3 if nenv > 64 else 1)
    b = ms.Engine(m, nenv); b.load_s24(); b.set_cohorts(1)
    a.step(45); b.step(45)
    ta, qa, va, wa = a.get_state(); tb, qb, vb, wb = b.get_state()
    assert np.array_equal(qa, qb) and np.array_equal(va, vb) and np.array_equal(a.get_stats(), b.get_stats())
    assert np.isfinite(qa).all()
    a.close(); b.close()


def _convex_zoo(lib, with_floor):
    """a static box and five free bodies whose pairs all go through the generic convex narrow phase (or plane-x)"""
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.004)
    if with_floor:
        lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, -1, -1, -1, -1)
    lib.mjh_builder_add_geom(b, b"block", 0, 6, D(0.25, 0.2, 0.1), D(0, 0, 0.1), None, None, -1, -1, -1, -1)
    specs = [(b"cyl1", 5, (0.07, 0.06, 0)), (b"cap", 3, (0.04, 0.09, 0)), (b"ell", 4, (0.09, 0.06, 0.04)),
             (b"cyl2", 5, (0.05, 0.10, 0)), (b"sph", 2, (0.06, 0, 0))]
    for k, (name, gt, size) in enumerate(specs):
        bd = lib.mjh_builder_add_body(b, name, 0, D(0.15 * k - 0.3, 0, 0.6), None, 0.0)
        lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
        lib.mjh_builder_add_geom(b, None, bd, gt, D(*size), None, None, None, -1, -1, -1, -1)
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    return m


@pytest.mark.gpu
def test_generic_convex_contacts_match_oracle(lib):
    """cylinder-x, capsule-box, ellipsoid-x (no analytic routine -> portal refinement over support mappings): contacts of
    64 random clusters, device (fp32) against the oracle (fp64), from identical poses"""
    m = _convex_zoo(lib, with_floor=False)
    assert m.nv == 30 and m.npair == 5 + 10      # every free geom against the block and against each other
    nenv = 192
    rng = np.random.default_rng(11)
    q = np.zeros((nenv, m.nq))
    for i in range(nenv):
        for k in range(5):
            p = rng.normal(size=3); p *= rng.uniform(0.05, 0.32) / np.linalg.norm(p)
            p[2] = abs(p[2]) + 0.2 + rng.uniform(0.0, 0.12)              # above the block's top face (z = 0.2), close together
            quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
            q[i, 7*k:7*k+3] = p; q[i, 7*k+3:7*k+7] = quat
    e = ms.Engine(m, nenv)
    e.set_initial_qpos(q); e.reset(); e.forward(); e.synchronize()
    st = e.get_stats()
    npairs = nsoft = nmiss = 0
    for i in range(nenv):
        d = orc.OrcData(m.ptr); d.set_qpos(q[i]); d.call("reset"); d.call("forward")
        oc = {x["geom"]: x for x in d.contacts()}
        c = e.get_contacts(i)
        dc = {tuple(int(v) for v in g): k for k, g in enumerate(c["geom"])}
        assert len(dc) == len(c["dist"])                                   # one contact per convex pair
        for key in set(oc) | set(dc):
            if key not in oc or key not in dc:                             # only grazing contacts may be seen by one side only
                depth = -oc[key]["dist"] if key in oc else -c["dist"][dc[key]]
                assert depth < 2e-4, (i, key, depth)
                nmiss += 1
                continue
            o, k = oc[key], dc[key]
            npairs += 1
            if o["dist"] < -0.03:                                          # deep overlaps are ill-conditioned (see the oracle test)
                continue
            np.testing.assert_allclose(c["dist"][k], o["dist"], atol=3e-4, err_msg=str((i, key)))
            if np.abs(c["frame"][k][:3] - o["frame"][:3]).max() > 0.05 or np.abs(c["pos"][k] - o["pos"]).max() > 5e-3:
                nsoft += 1                                                 # exit point next to an edge of the Minkowski difference
    e.close()
    assert npairs >= 150 and nsoft <= 0.12 * npairs and nmiss <= 0.05 * npairs, (npairs, nsoft, nmiss)


@pytest.mark.gpu
def test_generic_convex_rollout(lib):
    """the same bodies dropped on the block and the floor: short-horizon trajectory parity, then both come to rest
    without sinking in (one contact per convex pair: flat-on-flat rests are soft, so only bounds are compared late)"""
    m = _convex_zoo(lib, with_floor=True)
    q0 = m.array("qpos0").copy()
    rng = np.random.default_rng(3)
    for k in range(5):
        q0[7*k:7*k+3] = [0.11 * k - 0.22, 0.03 * (k % 2), 0.32 + 0.02 * k]
        quat = rng.normal(size=4); q0[7*k+3:7*k+7] = quat / np.linalg.norm(quat)
    st, ncon, nefc = _compare_rollout(m, q0, [1, 10, 20], [1e-4, 5e-4, 2e-3])   # bodies start in contact: the contact points carry the 1e-6 portal tolerance
    e = ms.Engine(m, 2)
    e.set_initial_qpos(np.tile(q0, (2, 1))); e.reset(); e.step(1500)
    _, q, v, _ = e.get_state()
    assert e.get_stats()[0, 3] == 0
    assert np.all(q[0, 2::7] > 0.03) and np.all(q[0, 2::7] < 0.45)          # nothing fell through the floor or flew away
    assert np.abs(v[0]).max() < 30.0                                        # (the round ones keep rolling: no rolling friction at condim 3)
    e.close()


def _random_hull(rng, n, scale):
    """a random convex polyhedron (vertices + outward-oriented triangles)"""
    from scipy.spatial import ConvexHull
    pts = rng.normal(size=(n, 3)); pts /= np.linalg.norm(pts, axis=1)[:, None]; pts *= scale * rng.uniform(0.7, 1.0, size=(n, 1))
    hull = ConvexHull(pts)
    faces = []
    for simplex, eq in zip(hull.simplices, hull.equations):
        a, b, c = pts[simplex]
        faces.append(simplex if np.dot(np.cross(b - a, c - a), eq[:3]) > 0 else simplex[::-1])
    return pts, np.array(faces, dtype=np.int32)


def _mesh_zoo(lib, with_floor):
    """a static block, three free convex-mesh bodies (a box given as a mesh, two random polyhedra) and a free cylinder"""
    import ctypes as C
    from test_oracle_collision import CUBE_F, CUBE_V
    rng = np.random.default_rng(21)
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.004)
    if with_floor:
        lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, -1, -1, -1, -1)
    lib.mjh_builder_add_geom(b, b"block", 0, 6, D(0.25, 0.2, 0.1), D(0, 0, 0.1), None, None, -1, -1, -1, -1)
    meshes = [(CUBE_V * np.array([0.08, 0.06, 0.04]), CUBE_F), _random_hull(rng, 40, 0.09), _random_hull(rng, 120, 0.07)]
    for k, (v, f) in enumerate(meshes):
        v = np.ascontiguousarray(v, dtype=np.float64); f = np.ascontiguousarray(f, dtype=np.int32)
        mid = lib.mjh_builder_add_mesh(b, v.ctypes.data_as(C.POINTER(C.c_double)), len(v), f.ctypes.data_as(C.POINTER(C.c_int)), len(f), None)
        assert mid == k
        bd = lib.mjh_builder_add_body(b, b"mesh%d" % k, 0, D(0.2 * k - 0.3, 0, 0.6), None, 0.0)
        lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
        assert lib.mjh_builder_add_mesh_geom(b, None, bd, mid, None, None, None, -1, -1, -1, -1) >= 0
    bd = lib.mjh_builder_add_body(b, b"cyl", 0, D(0.3, 0, 0.6), None, 0.0)
    lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
    lib.mjh_builder_add_geom(b, None, bd, 5, D(0.05, 0.08, 0), None, None, None, -1, -1, -1, -1)
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    return m


def test_convex_mesh_contacts_match_oracle(lib):
    """mesh geoms collide as convex hulls through their vertex tables: plane-mesh (up to 4 points) and mesh-x (portal
    refinement with a vertex-scan support mapping), device against oracle from identical poses"""
    m = _mesh_zoo(lib, with_floor=True)
    assert m.c.nmesh == 3 and m.nv == 24 and m.npair == 4 + 4 + 6
    nenv = 128
    rng = np.random.default_rng(12)
    q = np.zeros((nenv, m.nq))
    for i in range(nenv):
        for k in range(4):
            if i % 2 == 0:      # a cluster above the block
                p = rng.normal(size=3); p *= rng.uniform(0.05, 0.3) / np.linalg.norm(p); p[2] = abs(p[2]) + 0.2 + rng.uniform(0.0, 0.1)
            else:               # spread out on the floor next to the block
                p = np.array([0.5 + 0.25 * k, rng.uniform(-0.1, 0.1), rng.uniform(0.02, 0.09)])
            quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
            q[i, 7*k:7*k+3] = p; q[i, 7*k+3:7*k+7] = quat
    e = ms.Engine(m, nenv)
    e.set_initial_qpos(q); e.reset(); e.forward(); e.synchronize()
    npairs = nsoft = nmiss = nplane = 0
    for i in range(nenv):
        d = orc.OrcData(m.ptr); d.set_qpos(q[i]); d.call("reset"); d.call("forward")
        oc = {}
        for x in d.contacts():
            oc.setdefault(x["geom"], []).append(x)
        c = e.get_contacts(i)
        dc = {}
        for k, g in enumerate(c["geom"]):
            dc.setdefault(tuple(int(v) for v in g), []).append(k)
        for key in set(oc) | set(dc):
            if key not in oc or key not in dc:
                depth = max(-x["dist"] for x in oc[key]) if key in oc else max(-c["dist"][k] for k in dc[key])
                assert depth < 2e-4, (i, key, depth)
                nmiss += 1
                continue
            if key[0] == 0:                                       # plane - x: the same points (the order may differ on ties)
                od = sorted(x["dist"] for x in oc[key]); dd = sorted(c["dist"][k] for k in dc[key])
                if len(od) == len(dd):
                    np.testing.assert_allclose(dd, od, atol=2e-5, err_msg=str((i, key)))
                    nplane += 1
                else:
                    assert abs(len(od) - len(dd)) == 1 and max(od[-1], dd[-1]) > -2e-4, (i, key, od, dd)      # a grazing vertex
                continue
            o, k = oc[key][0], dc[key][0]
            assert len(oc[key]) == 1 and len(dc[key]) == 1
            npairs += 1
            if o["dist"] < -0.03:
                continue
            np.testing.assert_allclose(c["dist"][k], o["dist"], atol=3e-4, err_msg=str((i, key)))
            if np.abs(c["frame"][k][:3] - o["frame"][:3]).max() > 0.05 or np.abs(c["pos"][k] - o["pos"]).max() > 5e-3:
                nsoft += 1
    e.close()
    assert npairs >= 100 and nplane >= 100 and nsoft <= 0.15 * npairs and nmiss <= 0.05 * (npairs + nplane), (npairs, nplane, nsoft, nmiss)


def test_convex_mesh_rollout(lib):
    """mesh bodies dropped on the block and the floor: short-horizon parity, then rest on the floor / the block"""
    m = _mesh_zoo(lib, with_floor=True)
    q0 = m.array("qpos0").copy()
    rng = np.random.default_rng(4)
    for k in range(4):
        q0[7*k:7*k+3] = [0.16 * k - 0.24, 0.02 * (k % 2), 0.30 + 0.03 * k]
        quat = rng.normal(size=4); q0[7*k+3:7*k+7] = quat / np.linalg.norm(quat)
    _compare_rollout(m, q0, [1, 10, 25], [1e-5, 2e-4, 3e-3])
    e = ms.Engine(m, 2)
    e.set_initial_qpos(np.tile(q0, (2, 1))); e.reset(); e.step(1500)
    _, q, v, _ = e.get_state()
    assert e.get_stats()[0, 3] == 0
    assert np.all(q[0, 2::7] > 0.02) and np.all(q[0, 2::7] < 0.45)
    assert np.abs(v[0]).max() < 5.0                                           # (a polyhedron on a single contact point keeps rocking a little; the cylinder may roll)
    e.close()


def test_spawn_pool_of_cubes_spheres_cylinders_and_meshes(lib):
    """C4's spawn/destroy service as slot toggling over a mixed pool — the object types the reference's spawn test draws
    (test/test_spawn_and_destroy.py:13-14: CUBE, SPHERE, CYLINDER, MESH; sizes 0.05 * [2, 5]) dropped onto one spot of the
    world/empty.xml floor (condim 4, friction 2 / 0.05 / 0.01) so that they pile: every new arrival meets the others
    through box-box, sphere-x and the generic convex pairs.  Each env has its own schedule; oracle mirrors it."""
    import ctypes as C
    from test_oracle_collision import CUBE_F, CUBE_V
    rng = np.random.default_rng(8)
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.005)
    lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, D(2, 0.05, 0.01), 4, -1, -1, -1)
    hull_v, hull_f = _random_hull(rng, 60, 0.12)
    meshes = [(CUBE_V * np.array([0.12, 0.09, 0.06]), CUBE_F), (hull_v, hull_f)]
    mids = []
    for v, f in meshes:
        v = np.ascontiguousarray(v, dtype=np.float64); f = np.ascontiguousarray(f, dtype=np.int32)
        mids.append(lib.mjh_builder_add_mesh(b, v.ctypes.data_as(C.POINTER(C.c_double)), len(v), f.ctypes.data_as(C.POINTER(C.c_int)), len(f), None))
    pool = [("cube", 6, (0.10, 0.10, 0.10)), ("sphere", 2, (0.11, 0, 0)), ("cyl", 5, (0.09, 0.12, 0)), ("mesh", 0, None),
            ("cube", 6, (0.15, 0.08, 0.06)), ("sphere", 2, (0.07, 0, 0)), ("cyl", 5, (0.12, 0.05, 0)), ("mesh", 1, None)]
    for k, (kind, gt, size) in enumerate(pool):
        bd = lib.mjh_builder_add_body(b, b"object_%d" % k, 0, D(3.0 + 0.6 * k, 3.0, 0.3), None, 0.0)     # parked out of the way
        lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
        if kind == "mesh":
            assert lib.mjh_builder_add_mesh_geom(b, None, bd, mids[gt], None, None, None, -1, -1, -1, -1) >= 0
        else:
            lib.mjh_builder_add_geom(b, None, bd, gt, D(*size), None, None, None, -1, -1, -1, -1)
    lib.mjh_builder_set_capacity(b, 48, 48 * 6)
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    nslot = len(pool)
    assert m.nv == 6 * nslot and m.npair == nslot + nslot * (nslot - 1) // 2
    nenv = 6
    e = ms.Engine(m, nenv)
    ds = [orc.OrcData(m.ptr) for _ in range(nenv)]
    e.reset(); [d.call("reset") for d in ds]
    for k in range(nslot):                         # everything starts destroyed
        e.set_slot_active(k + 1, False)
    mask = [(1 << (nslot + 1)) - 2] * nenv
    for i, d in enumerate(ds):
        d.L.orc_set_slot_mask(d.d, mask[i])
    order = [rng.permutation(nslot) for _ in range(nenv)]
    worst = 0.0
    for rnd in range(nslot + 2):
        for i, d in enumerate(ds):
            if rnd < nslot:                         # spawn the next object of this env's schedule above the pile, with a twist
                k = int(order[i][rnd])
                pos = np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), 0.55 + 0.05 * rnd])
                quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
                vel = np.array([0, 0, -0.5, *rng.normal(size=3)])
                e.set_slot_active(k + 1, True, env0=i, n=1); e.set_body_pose(i, k + 1, pos, quat, vel)
                mask[i] &= ~(1 << (k + 1)); d.L.orc_set_slot_mask(d.d, mask[i])
                d.f("qpos")[7*k:7*k+3] = pos; d.f("qpos")[7*k+3:7*k+7] = quat; d.f("qvel")[6*k:6*k+6] = vel
            else:                                   # destroy the first two arrivals again
                k = int(order[i][rnd - nslot])
                e.set_slot_active(k + 1, False, env0=i, n=1)
                mask[i] |= 1 << (k + 1); d.L.orc_set_slot_mask(d.d, mask[i])
        e.step(24); [d.step(24) for d in ds]
        _, q, v, _ = e.get_state()
        st = e.get_stats()
        assert (st[:, 3] == 0).all() and all(d.i("warn") == 0 for d in ds)
        oq = np.array([d.f("qpos") for d in ds])
        err = np.abs(q - oq).max(axis=1)
        worst = max(worst, float(np.sort(err)[-3]))          # impacts in a pile of round things are chaotic: an env or two per round may fork
        assert (err < 2e-2).sum() >= nenv - 2 and err.max() < 0.5, (rnd, err)
        for i, d in enumerate(ds):                             # destroyed objects stay where they were parked / left
            for k in range(nslot):
                if mask[i] >> (k + 1) & 1:
                    assert (v[i, 6*k:6*k+6] == 0).all()
        # every segment starts from identical states
        e.set_state(qpos=oq, qvel=np.array([d.f("qvel") for d in ds]), warmstart=np.array([d.f("qacc_warmstart") for d in ds]))
    assert max(d.i("ncon") for d in ds) >= 8
    e.close()


def _noslip_scene(lib, nbox, noslip, floss_joint=False):
    """boxes resting on a 0.3 rad incline (friction holds them), optionally a slider with friction loss"""
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.005)
    o = ms.capi.Option(); lib.mjh_builder_get_option(b, o); o.noslip_iterations = noslip; lib.mjh_builder_set_option(b, o)
    ang = 0.3
    tilt = D(np.cos(ang / 2), 0, np.sin(ang / 2), 0)
    lib.mjh_builder_add_geom(b, b"ramp", 0, 0, D(0, 0, 0.05), None, tilt, None, -1, -1, -1, -1)
    n = np.array([np.sin(ang), 0, np.cos(ang)]); t = np.array([0.0, 1.0, 0.0])
    for k in range(nbox):
        h = 0.06 + 0.01 * (k % 3)
        p = n * (h - 5e-4) + t * (0.3 * k - 0.15 * (nbox - 1))
        bd = lib.mjh_builder_add_body(b, b"box%d" % k, 0, D(*p), tilt, 0.0)
        lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
        lib.mjh_builder_add_geom(b, None, bd, 6, D(0.08, 0.1, h), None, None, None, 3 + (k % 2), -1, -1, -1)   # condim 3 and 4
    if floss_joint:      # a weight on a vertical slider held by dry joint friction (friction-loss row in the noslip pass)
        bd = lib.mjh_builder_add_body(b, b"slider", 0, D(1.0, 0, 0.5), None, 0.0)
        lib.mjh_builder_add_joint(b, b"slide", bd, 2, None, D(0, 0, 1), None, 0.0, 0.0, 0.0, 30.0, 0.0)
        lib.mjh_builder_add_geom(b, None, bd, 2, D(0.05, 0, 0), None, None, None, -1, 0, 0, -1)
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    return m


@pytest.mark.parametrize("nbox,policy", [(2, 1), (6, 1), (6, 2), (12, 2), (20, 2)], ids=["dual", "single-block", "global-pools", "many-body", "four-row"])
def test_noslip_sweeps_stop_the_creep_and_match_oracle(lib, layout_policy, nbox, policy):
    """option noslip_iterations (model/ontology/scene.xml:2-3): friction-only sweeps without the regulariser after the
    main PGS.  Boxes held by friction on an incline creep at ~1.5 mm/s with the soft constraint; with noslip they stay.
    All sweep implementations (dual-block nv <= 32, single-block, many-body in global pools)."""
    layout_policy(policy)
    res = {}
    for noslip in (0, 5):
        m = _noslip_scene(lib, nbox, noslip, floss_joint=True)
        assert m.opt.noslip_iterations == noslip
        e = ms.Engine(m, 2); e.reset()
        d = orc.OrcData(m.ptr); d.call("reset")
        e.step(200); d.step(200)
        _, q1, _, _ = e.get_state()
        oq1 = d.f("qpos").copy()
        e.step(200); d.step(200)
        _, q2, v2, _ = e.get_state()
        st = e.get_stats()
        assert st[0, 3] == 0 and d.i("warn") == 0 and st[0, 0] == d.i("ncon") == 4 * nbox
        np.testing.assert_allclose(q2[0], d.f("qpos"), atol=2e-4)                     # device == oracle (with or without noslip)
        np.testing.assert_array_equal(q2[0], q2[1])
        creep = np.abs(q2[0] - q1[0])[:7 * nbox].reshape(nbox, 7)[:, :3].max() / (200 * 0.005)
        ocreep = np.abs(d.f("qpos") - oq1)[:7 * nbox].reshape(nbox, 7)[:, :3].max() / (200 * 0.005)
        res[noslip] = (creep, ocreep)
        # the slider (5 N of weight against 30 N of dry joint friction) sags with the soft friction row, not with noslip
        assert (abs(q2[0, 7 * nbox]) < 1e-5) if noslip else (q2[0, 7 * nbox] < -5e-3)
        e.close()
    assert res[0][0] > 5e-4 and res[0][1] > 5e-4, res        # soft contacts creep down the incline ...
    assert res[5][0] < 2e-5 and res[5][1] < 2e-5, res        # ... noslip holds them


def test_state_transplant_after_a_model_change(lib):
    """A18: the reference recompiles the model on spawn / destroy and carries the state over by body NAME
    (add_old_state, mj_sim.cpp:465-558).  Old model: three named boxes mid-fall; new model: one box destroyed, one added,
    bodies declared in another order.  The survivors continue exactly where they were."""
    def build(names, z0=0.5):
        b = lib.mjh_builder_create()
        set_opt(lib, b, timestep=0.005)
        lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, -1, -1, -1, -1)
        for name, x in names:
            bd = lib.mjh_builder_add_body(b, name, 0, D(x, 0, z0), None, 0.0)
            lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
            lib.mjh_builder_add_geom(b, None, bd, 6, D(0.05, 0.06, 0.07), None, None, None, -1, -1, -1, -1)
        m = ms.Model(lib.mjh_builder_compile(b), lib); lib.mjh_builder_destroy(b)
        return m
    old = build([(b"object_0", 0.0), (b"object_1", 0.6), (b"object_2", 1.2)])
    new = build([(b"object_7", 3.0), (b"object_2", -1.0), (b"object_0", -2.0)], z0=0.9)       # object_1 destroyed, object_7 spawned
    nenv = 3
    ea, eb = ms.Engine(old, nenv), ms.Engine(new, nenv)
    rng = np.random.default_rng(1)
    q = np.tile(old.array("qpos0"), (nenv, 1)); v = rng.normal(size=(nenv, old.nv)) * 0.3
    for i in range(nenv):
        for k in range(3):
            x = rng.normal(size=4); q[i, 7*k+3:7*k+7] = x / np.linalg.norm(x)
    ea.set_initial_qpos(q); ea.reset(); ea.set_state(qvel=v); ea.step(40)
    ta, qa, va, wa = ea.get_state()
    assert eb.transplant_state_from(ea, full_qpos=True) == 2                       # object_0 and object_2 matched
    tb, qb, vb, wb = eb.get_state()
    np.testing.assert_array_equal(tb, ta)
    # new-model slots: object_7 -> 0, object_2 -> 1, object_0 -> 2
    np.testing.assert_array_equal(qb[:, 7:14], qa[:, 14:21]); np.testing.assert_array_equal(qb[:, 14:21], qa[:, 0:7])
    np.testing.assert_array_equal(vb[:, 6:12], va[:, 12:18]); np.testing.assert_array_equal(wb[:, 12:18], wa[:, 0:6])
    np.testing.assert_allclose(qb[:, 0:3], [[3.0, 0, 0.9]] * nenv)                   # the spawned one keeps its baked pose
    ea.step(120); eb.step(120)
    _, qa2, va2, _ = ea.get_state(); _, qb2, vb2, _ = eb.get_state()
    np.testing.assert_allclose(qb2[:, 7:14], qa2[:, 14:21], atol=2e-4); np.testing.assert_allclose(qb2[:, 14:21], qa2[:, 0:7], atol=2e-4)
    assert (qb2[:, 2] < 0.2).all()                                                  # and the new box fell to the floor
    # literal mode: the reference copies body_jntnum qpos scalars, i.e. only x of a free body (mj_sim.cpp:510-513)
    ec = ms.Engine(new, nenv)
    assert ec.transplant_state_from(ea, full_qpos=False) == 2
    _, qc, vc, _ = ec.get_state()
    np.testing.assert_array_equal(qc[:, 7], qa2[:, 14]); np.testing.assert_allclose(qc[:, 8:14], [[0, 0.9, 1, 0, 0, 0]] * nenv)
    np.testing.assert_array_equal(vc[:, 6:12], va2[:, 12:18])
    for e in (ea, eb, ec):
        e.close()


def test_plane_contacts_of_round_geoms_match_oracle(lib):
    """analytic plane pairs on the device against the oracle from identical random poses: ellipsoid (support point),
    cylinder (rim points), capsule, sphere — and the rest pose of an ellipsoid on the floor"""
    m = _convex_zoo(lib, with_floor=True)
    nenv = 48
    rng = np.random.default_rng(31)
    q = np.zeros((nenv, m.nq))
    for i in range(nenv):
        for k in range(5):
            quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
            q[i, 7*k:7*k+3] = [0.6 + 0.4 * k, rng.uniform(-0.1, 0.1), rng.uniform(0.03, 0.1)]      # next to the block, touching the floor
            q[i, 7*k+3:7*k+7] = quat
    e = ms.Engine(m, nenv)
    e.set_initial_qpos(q); e.reset(); e.forward(); e.synchronize()
    nsame = npts = 0
    for i in range(nenv):
        d = orc.OrcData(m.ptr); d.set_qpos(q[i]); d.call("reset"); d.call("forward")
        oc = d.contacts(); c = e.get_contacts(i)
        if len(oc) != len(c["dist"]):
            assert abs(len(oc) - len(c["dist"])) <= 1            # a grazing rim point
            continue
        nsame += 1
        if oc:
            assert [tuple(g) for g in c["geom"]] == [x["geom"] for x in oc]
            np.testing.assert_allclose(c["dist"], [x["dist"] for x in oc], atol=2e-6)
            np.testing.assert_allclose(c["pos"], [x["pos"] for x in oc], atol=2e-5)
            npts += len(oc)
    assert nsame >= nenv - 4 and npts >= 3 * nenv
    # an ellipsoid dropped flat rests at its smallest half-axis
    q0 = m.array("qpos0").copy()
    for k in range(5):
        q0[7*k:7*k+3] = [0.6 + 0.4 * k, 0, 0.2]
    q0[7*2:7*2+3] = [1.4, 0, 0.06]
    e2 = ms.Engine(m, 1); e2.set_initial_qpos(q0[None, :]); e2.reset(); e2.step(600)
    _, qq, vv, _ = e2.get_state()
    assert abs(qq[0, 7*2+2] - 0.04) < 2e-3 and np.abs(vv[0, 12:18]).max() < 5e-2
    e.close(); e2.close()


@pytest.mark.parametrize("scene,copies", [("arm7", 4), ("pendulum", 3)])
def test_sub_wave_packing_matches_the_oracle_per_instance(scene, copies):
    """mjh_model_replicate: `copies` environments per wavefront (instances of the moving trees in one model, sharing the
    static geometry, never colliding with each other).  Every instance must follow ITS oracle."""
    m = ms.scene(scene, 0) if scene == "arm7" else ms.scene(scene)
    r = m.replicate(copies)
    assert r.nv == copies * m.nv and r.ntree == copies * m.ntree and r.npair == copies * m.npair
    nw = 3                                    # wavefronts
    rng = np.random.default_rng(5)
    q0 = np.tile(m.array("qpos0"), (nw * copies, 1)); v0 = rng.normal(size=(nw * copies, m.nv)) * 0.4
    if scene == "arm7":
        q0 += rng.uniform(-0.2, 0.2, size=q0.shape)
    e = ms.Engine(r, nw)
    e.set_initial_qpos(q0.reshape(nw, -1)); e.reset(); e.set_state(qvel=v0.reshape(nw, -1))
    ds = []
    for i in range(nw * copies):
        d = orc.OrcData(m.ptr); d.set_qpos(q0[i]); d.call("reset"); d.f("qvel")[:] = v0[i]; ds.append(d)
    for n in (1, 100, 400):
        done = int(round(ds[0].f("time")[0] / m.opt.timestep))
        e.step(n - done, True); [d.step(n - done, 1) for d in ds]
        _, q, v, _ = e.get_state()
        q = q.reshape(nw * copies, m.nq); v = v.reshape(nw * copies, m.nv)
        tol = {1: 1e-5, 100: 2e-3, 400: 2e-2}[n]     # (the arm sags into its limits: when a limit row switches on is a rounding matter)
        np.testing.assert_allclose(q, [d.f("qpos") for d in ds], atol=tol, err_msg=f"step {n}")
        np.testing.assert_allclose(v, [d.f("qvel") for d in ds], atol=10 * tol, err_msg=f"step {n}")
    fi = e.get_field("qfrc_inverse").reshape(nw * copies, m.nv)
    ref = np.array([d.f("qfrc_inverse") for d in ds])
    np.testing.assert_allclose(fi, ref, atol=2e-2 * max(1.0, np.abs(ref).max()))
    assert (e.get_stats()[:, 3] == 0).all()
    e.close()
