"""Round 4 GPU tests: the window sweep (csrc/window_pgs.h — mjh_step of small free-body models as assemble launch + mjh_window_kernel,
projected Gauss-Seidel in mj_solPGS's row order over windows of 16 consecutive rows, four envs per wavefront) against the fused
kernel's contact-patch sweep and against the oracle; S24D (the "30-contact" reading of the metric: 140 rows per env, windows streamed
beyond the register-resident ones) teacher-forced like S24; the independent check of the project's own box-box manifold."""
import os

import numpy as np
import pytest

import mujoco_sim_amd as ms
import orc
from helpers import oracle_s24
from test_gpu_teacher_forced import teacher_forced, summarize, S24_TOL_Q, S24_TOL_V

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _engine(m, nenv, window, load=None):
    lib = ms.capi.load()
    lib.mjh_set_window_solver(1 if window else 0)
    try:
        e = ms.Engine(m, nenv)
    finally:
        lib.mjh_set_window_solver(1)
    assert e.window_solver() == (1 if window else 0) and e.solver_order() == 2 and e.patch_sweep() == 1
    tab = load(e) if load else e.load_s24()
    return e, tab


def test_window_sweep_equals_the_fused_patch_sweep_up_to_rounding():
    """same constraint order, same clamps, same stopping rule, different grouping of the arithmetic (windows of 16 consecutive rows with
    a dense J over the 24 dofs, against patches of one body pair): one step from identical states agrees to fp32 rounding — qpos 1e-6,
    qvel 2e-5 relative, as each does against the oracle — with identical contact / row counts and (nearly always) identical sweep
    counts; 200 free-running steps from reset stay together until a contact set forks"""
    m = ms.scene("s24")
    nenv = 64
    a, tab = _engine(m, nenv, True)
    b, _ = _engine(m, nenv, False)
    a.step(150); b.step(150)                    # fall and settle, each on its own
    worst_q = worst_v = 0.0; same_sweeps = []; n = 0
    for k in range(80):
        t, q, v, w = a.get_state()
        b.set_state(qpos=q, qvel=v, time=t, warmstart=w)
        a.step(1); b.step(1)
        _, qa, va, wa = a.get_state(); _, qb, vb, wb = b.get_state()
        sa, sb = a.get_stats(), b.get_stats()
        assert np.array_equal(sa[:, :2], sb[:, :2]) and np.array_equal(sa[:, 3] & 7, sb[:, 3] & 7), "same contacts, rows and flags from the same state"
        rq = np.abs(qa - qb).max(axis=1) / np.maximum(1, np.abs(qb).max(axis=1)); rv = np.abs(va - vb).max(axis=1) / np.maximum(1, np.abs(vb).max(axis=1))
        worst_q = max(worst_q, rq.max()); worst_v = max(worst_v, rv.max())
        same_sweeps.append(sa[:, 2] == sb[:, 2]); n += nenv
    same = float(np.mean(same_sweeps))
    print(f"WINDOW-vs-PATCH {n} env-steps: qpos rel max {worst_q:.2e}, qvel rel max {worst_v:.2e}, same sweep count {same:.4f}")
    assert worst_q <= S24_TOL_Q and worst_v <= S24_TOL_V and same >= 0.98
    st = a.get_stats()
    assert st[:, 0].mean() >= 12 and (st[:, 3] & 7 == 0).all()
    a.close(); b.close()


def test_window_sweep_results_do_not_depend_on_which_envs_share_a_wavefront():
    """four envs per wavefront, grouped by the launch order (longest job first, renewed every few steps) and by the cohort split:
    every env's arithmetic stays inside its 16-lane row, so 1 cohort / 3 cohorts / another batch size give the same bits"""
    m = ms.scene("s24")
    outs = []
    for nenv, nc in ((1280, 1), (1280, 3), (1283, 2)):
        e, _ = _engine(m, nenv, True)
        e.set_cohorts(nc)
        e.step(70)
        t, q, v, w = e.get_state(); st = e.get_stats()
        outs.append((q[:1280], v[:1280], w[:1280], st[:1280, :3]))
        e.close()
    for k in (1, 2):
        for x, y in zip(outs[0], outs[k]):
            assert np.array_equal(x, y)


def _s24d(nenv):
    pen = 0.175
    m = ms.scene("s24pen", pen, 64)
    e, tab = _engine(m, nenv, True)
    q = tab["qpos"].reshape(nenv, 4, 7)
    for i in range(nenv):
        rng = np.random.default_rng(0x524D0000 + i)
        for k in range(4):
            yaw = rng.uniform(-0.3, 0.3)
            q[i, k] = [(-1 if k & 1 else 1) * pen / 2, (-1 if k & 2 else 1) * pen / 2, 0.16 + 0.02 * k, np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]
    e.set_initial_qpos(tab["qpos"]); e.reset()
    return m, e, tab


def test_s24d_teacher_forced_with_streamed_windows():
    """S24D = bench.py's `s24d`: the four boxes released flat, ~33 contacts / ~140 rows per env — nine windows, the ninth and later ones
    streamed from memory every sweep.  Oracle settles 400 steps; 100 teacher-forced steps at S24's tolerances, no capacity flag."""
    nenv = 12
    m, e, tab = _s24d(nenv)
    ds = [oracle_s24(m, tab, i) for i in range(nenv)]
    for d in ds:
        d.step(400)
    r = teacher_forced(e, ds, 100)
    s = summarize("s24d/default=mj_solPGS-row-order(window sweep)", r)
    a = r["agree"].astype(bool)
    assert r["ncon"].mean() >= 24 and r["nefc"].max() > 128, "the scene must exercise the streamed windows (more than 8 x 16 rows)"
    assert s["agree_fraction"] >= 0.95, s
    assert r["eq"][a].max() <= S24_TOL_Q and r["ev"][a].max() <= S24_TOL_V, s
    st = e.get_stats()
    assert (st[:, 3] & 3 == 0).all()
    e.close()


def test_window_sweep_handles_empty_sparse_and_reset_environments():
    """envs without any contact (in free fall) finish in the assemble launch; a NaN state is reset by the window kernel's mj_checkAcc
    path or the assemble launch's mj_checkPos; time advances by dt for every env either way"""
    m = ms.scene("s24")
    nenv = 8
    e, tab = _engine(m, nenv, True)
    q0 = tab["qpos"].copy()
    q0[0].reshape(4, 7)[:, 2] += 50.0                  # env 0: far above the floor, no contacts for a long time
    e.set_initial_qpos(q0); e.reset()
    d0 = oracle_s24(m, {k: (q0 if k == "qpos" else v) for k, v in tab.items()}, 0)
    e.step(30); d0.step(30)
    t, q, v, w = e.get_state(); st = e.get_stats()
    np.testing.assert_allclose(t, 30 * m.opt.timestep, rtol=1e-12)
    assert st[0, 0] == 0 and st[0, 2] == 0
    np.testing.assert_allclose(q[0], d0.f("qpos"), atol=2e-5); np.testing.assert_allclose(v[0], d0.f("qvel"), atol=2e-5)
    v[3, :] = np.nan
    e.set_state(qvel=v)
    e.step(2)
    t2, q2, v2, _ = e.get_state(); st2 = e.get_stats()
    assert np.isfinite(q2).all() and np.isfinite(v2).all() and (st2[3, 3] & 4) and not (st2[[0, 1, 2, 4, 5, 6, 7], 3] & 4).any()
    np.testing.assert_allclose(t2, 32 * m.opt.timestep, rtol=1e-12)
    e.close()


def test_device_box_box_manifold_passes_the_independent_geometry_check():
    """VERDICT r03 next #6a: the device's own box-box manifold (dev_collide.h c_box_box, reached through mjh_get_contacts) on more
    than 10 000 random overlapping box pairs — free boxes against each other and against the pen's wall boxes, per-env sizes —
    checked with tests/boxgeom.py (numpy; nothing shared with the routine): contacts iff no axis separates, the normal is the
    least-overlap candidate and points from the first geom to the second, every point sits midway between the two surfaces with
    `dist` the gap of the surfaces along the normal through it (1e-5), the point set = the clipped incident face."""
    import boxgeom as bg
    m = ms.scene("s24")
    nenv = 2560
    e = ms.Engine(m, nenv)
    tab = e.load_s24()
    rng = np.random.default_rng(4)
    q = np.zeros((nenv, 4, 7))
    for i in range(nenv):
        # two loose clusters of two boxes, well above the floor; poses as in boxgeom.random_pairs: arbitrary, stacked, twisted
        for c in range(2):
            base = np.array([rng.uniform(-0.06, 0.06), rng.uniform(-0.06, 0.06), 0.45 + 0.5 * c])
            R = bg.random_rotation(rng)
            mode = rng.integers(3)
            R2 = R @ bg.random_rotation(rng, [np.pi, 10.0 ** rng.uniform(-4, -1), 0.3][mode])
            s1, s2 = tab["geom_size"][i].reshape(-1, 3)[5 + 2 * c], tab["geom_size"][i].reshape(-1, 3)[6 + 2 * c]
            k = rng.integers(3)
            off = rng.uniform(-0.5, 0.5, 3) * (s1 + s2); off[k] = (s1[k] + s2[k]) * (1 - 10.0 ** rng.uniform(-4, -0.7))
            for b, (pp, RR) in enumerate(((base, R), (base + R @ off, R2))):
                w = np.sqrt(max(0.0, 1 + np.trace(RR))) / 2
                qq = np.array([w, (RR[2, 1] - RR[1, 2]) / (4 * w), (RR[0, 2] - RR[2, 0]) / (4 * w), (RR[1, 0] - RR[0, 1]) / (4 * w)]) if w > 1e-3 else np.array([0, 1.0, 0, 0])
                q[i, 2 * c + b, :3] = pp; q[i, 2 * c + b, 3:] = qq / np.linalg.norm(qq)
    e.set_state(qpos=q.reshape(nenv, -1), qvel=np.zeros((nenv, m.nv)))
    e.forward(); e.synchronize()
    gp, gm = e.get_geom_state()
    st = e.get_stats()
    gtype = m.array("geom_type")
    boxes = [g for g in range(m.ngeom) if gtype[g] == 6]
    npairs = ncontacts = nover = 0
    failures = []
    for i in range(nenv):
        if st[i, 3] & 1:
            nover += 1; continue            # (contact capacity exceeded: the list is cut, not a manifold question)
        c = e.get_contacts(i)
        sizes = tab["geom_size"][i].reshape(-1, 3)
        by = {}
        for k in range(len(c["dist"])):
            by.setdefault(tuple(c["geom"][k]), []).append(k)
        for a in range(len(boxes)):
            for b in range(a + 1, len(boxes)):
                g1, g2 = boxes[a], boxes[b]
                if g1 < 5 and g2 < 5:
                    continue                # wall against wall: not a pair of the scene
                ks = by.get((g1, g2), [])
                b1 = (gp[i, g1], gm[i, g1].reshape(3, 3), sizes[g1]); b2 = (gp[i, g2], gm[i, g2].reshape(3, 3), sizes[g2])
                # pairs the broad phase drops (bounding spheres apart) have no contacts: the checker agrees or reports them
                n = c["frame"][ks[0], :3] if ks else np.zeros(3)
                for k in ks[1:]:
                    assert np.allclose(c["frame"][k, :3], n, atol=1e-6)
                bad = bg.check_contacts(b1, b2, 0.0, c["dist"][ks], c["pos"][ks], n, tol=1e-5, count_tol=1e-4)
                if ks:
                    npairs += 1; ncontacts += len(ks)
                if bad:
                    failures.append((i, g1, g2, len(ks), bad[:3]))
    assert nover < nenv // 50, f"{nover} envs over the contact capacity: thin the scene"
    assert npairs >= 10000, f"only {npairs} touching pairs"
    print(f"BOXBOX-INDEPENDENT device: {npairs} touching box pairs, {ncontacts} contacts, {len(failures)} violations, {nover} envs skipped (capacity)")
    kinds = {}
    for f in failures:
        for msg in f[4]:
            key = "".join(ch for ch in msg if not (ch.isdigit() or ch in ".e+-"))
            kinds[key] = kinds.get(key, 0) + 1
    assert not failures, (kinds, failures[:5])
    e.close()


def _loop_models():
    """(label, model, engine set-up) of articulated models in the LDS-resident layout — the instances that carry the in-kernel step loop"""
    import os
    from mujoco_sim_amd.tables import load_model_tables
    from helpers import hinge_pendulum_model
    lib = ms.capi.load()
    arm = ms.scene("arm7", 1).replicate(4)

    def arm_setup(e, rng):
        e.set_controlled_dofs(np.ones(arm.nv, dtype=np.int32))
        e.set_pd_controller(200.0, 50.0)
        lo, hi = ms.scene("arm7", 1).array("jnt_range").reshape(-1, 2).T
        e.set_pd_target(rng.uniform(np.tile(lo, 4), np.tile(hi, 4), size=(e.nenv, arm.nv)))
    pend = hinge_pendulum_model(lib, damping=0.3)

    def pend_setup(e, rng):
        e.set_state(qpos=rng.uniform(-1, 1, size=(e.nenv, 1)), qvel=rng.uniform(-2, 2, size=(e.nenv, 1)))
    gold = os.path.join(os.path.dirname(__file__), "golden")
    c5, z = load_model_tables(os.path.join(gold, "robot_c5_pendulum_bowl_mesh.npz"))

    def c5_setup(e, rng):
        e.set_controlled_dofs(z["controlled"].astype(np.int32))
        if "qvel0" in z:
            e.set_state(qvel=z["qvel0"][None, :] * rng.uniform(0.5, 1.5, size=(e.nenv, 1)))
    return [("arm7 x4 with the in-engine PD law", arm, arm_setup), ("damped pendulum", pend, pend_setup), ("C5 pendulum world + bowl (mesh contacts)", c5, c5_setup)]


@pytest.mark.parametrize("with_inverse", [False, True], ids=["step", "step+inverse"])
def test_in_kernel_step_loop_equals_one_launch_per_step(with_inverse):
    """mjh_step(e, n) with the step loop inside the kernel (VERDICT r03 missing #3; SURVEY §7.3 'one launch per n steps') against n
    launches of one step: bitwise equal state, warm start, clock and statistics — with a host command consumed by the first step of a
    call, the PD law evaluated in every step, and calls that are no multiple of the launch's step count"""
    for label, m, setup in _loop_models():
        out = []
        for spl in (1, 8, 3):
            e = ms.Engine(m, 1536)
            e.set_cohorts(3)
            setup(e, np.random.default_rng(11))
            e.set_steps_per_launch(spl)
            assert e.steps_per_launch == spl, label
            e.step(17, with_inverse)
            cmd = np.zeros((e.nenv, m.nv)); cmd[:, 0] = 0.7
            e.set_cmd(ddq=cmd)                                  # consumed by the first of the next 10 steps only
            e.step(10, with_inverse); e.step(1, with_inverse); e.step(5, with_inverse)
            t, q, v, w = e.get_state()
            f = e.get_field("qfrc_inverse") if with_inverse else np.zeros(1)
            out.append((t, q, v, w, e.get_stats()[:, :3], f))
            assert np.isfinite(q).all() and np.abs(v).max() > 0, label
            e.close()
        for k in (1, 2):
            for a, b in zip(out[0], out[k]):
                assert np.array_equal(a, b), f"{label}: {(8, 3)[k - 1]} steps per launch differ from one launch per step"
    # free-body models (window chain / patch sweep) and the many-body chain step once per launch, whatever is asked
    e = ms.Engine(ms.scene("s24"), 64); e.set_steps_per_launch(8)
    assert e.steps_per_launch == 1
    e.close()


@pytest.mark.parametrize("host_threads", [1, 0], ids=["thread-per-shard", "one-host-thread"])
def test_c5_as_eight_shards_on_one_device_equals_the_plain_engine(host_threads):
    """the multi-GPU config (multi_mujoco_sim.launch: C5) through the C host's group — 4096 envs as EIGHT shards
    of 512, all on device 0 (the one device there is; peer-copy transport: RCCL refuses duplicate devices), per-env spin, the state
    slice published every 3 steps and the consumer releasing it — against ONE engine over all 4096 envs: env order and bitwise state"""
    import os
    from mujoco_sim_amd.tables import load_model_tables
    lib = ms.capi.load()
    m, z = load_model_tables(os.path.join(os.path.dirname(__file__), "golden", "robot_c5_pendulum_bowl_mesh.npz"))
    nenv, nshard = 4096, 8
    spin = z["qvel0"][None, :] * np.random.default_rng(0xC5).uniform(0.5, 1.5, size=(nenv, 1))
    lib.mjh_group_set_transport(1); lib.mjh_group_set_host_threads(host_threads)
    try:
        g = ms.Group(m, nenv, [0] * nshard)
    finally:
        lib.mjh_group_set_transport(0); lib.mjh_group_set_host_threads(1)
    # (round 5: a host thread per shard issues that shard's launches, exports and peer copies; the caller's thread only posts and waits)
    assert lib.mjh_group_host_threads(g.h) == (nshard if host_threads else 0)
    single = ms.Engine(m, nenv)
    single.set_controlled_dofs(z["controlled"].astype(np.int32)); single.set_state(qvel=spin)
    assert [n for _, n in g.ranges] == [nenv // nshard] * nshard
    for (e0, n), e in zip(g.ranges, g.engines):
        e.set_controlled_dofs(z["controlled"].astype(np.int32)); e.set_state(qvel=spin[e0:e0 + n])
    for k in range(20):
        g.step(3, True); single.step(3, True)               # the stretch between two publishes: one launch per cohort and shard
        g.publish_device()
        for r in range(nshard):
            g.wait_publish(r); g.release_publish(r)         # a consumer on every shard's stream
    pub = g.publish()
    t, q, v, _ = single.get_state()
    ref = np.concatenate([t[:, None], q, v], axis=1).astype(np.float32)
    assert np.array_equal(pub, ref), np.abs(pub - ref).max()
    assert np.abs(v).max() > 0.1 and len(np.unique(np.round(v, 5), axis=0)) > nenv // 2      # the envs do differ
    g.close(); single.close()


def test_32_row_windows_for_the_envs_with_many_rows_equal_the_16_row_form_up_to_rounding():
    """window_pgs.h wn_run32: envs with more than 96 constraint rows (the ones a cohort's step waits for) are swept in 32-row windows, two
    per wavefront — same rows, same order, same stopping rule, one dot and one transpose-reduce per 32 rows.  Against the same engine
    with the section switched off (MJH_WINDOW32=0): envs with at most 96 rows are bitwise equal (their form did not change), the others
    agree to fp32 rounding with identical sweep counts almost everywhere; and the 32-row envs against the oracle like every other env"""
    import os
    m = ms.scene("s24")
    nenv = 2048
    os.environ["MJH_WINDOW64"] = "0"            # (round 5: S24's envs beyond 96 rows take the 64-row form by default; this test keeps them in the 32-row one)
    try:
        a = ms.Engine(m, nenv); tab = a.load_s24()
        os.environ["MJH_WINDOW32"] = "0"
        try:
            b = ms.Engine(m, nenv); b.load_s24()
        finally:
            del os.environ["MJH_WINDOW32"]
    finally:
        del os.environ["MJH_WINDOW64"]
    a.step(300); a.synchronize()
    heavy_seen = 0; worst_q = worst_v = 0.0; same_it = []
    for k in range(40):
        t, q, v, w = a.get_state()
        b.set_state(qpos=q, qvel=v, time=t, warmstart=w)
        a.step(1); b.step(1)
        _, qa, va, _ = a.get_state(); _, qb, vb, _ = b.get_state()
        sa, sb = a.get_stats(), b.get_stats()
        assert np.array_equal(sa[:, :2], sb[:, :2])                       # same contacts, same rows
        heavy = (sa[:, 1] > 96) & (sa[:, 1] <= 128)
        assert np.array_equal(qa[~heavy], qb[~heavy]) and np.array_equal(va[~heavy], vb[~heavy]) and np.array_equal(sa[~heavy, 2], sb[~heavy, 2])
        if heavy.any():
            heavy_seen += int(heavy.sum())
            worst_q = max(worst_q, float((np.abs(qa[heavy] - qb[heavy]).max(1) / np.maximum(1, np.abs(qb[heavy]).max(1))).max()))
            worst_v = max(worst_v, float((np.abs(va[heavy] - vb[heavy]).max(1) / np.maximum(1, np.abs(vb[heavy]).max(1))).max()))
            same_it.append(float((sa[heavy, 2] == sb[heavy, 2]).mean()))
    print(f"WINDOW32: {heavy_seen} env-steps in 32-row windows of {40 * nenv}: qpos {worst_q:.2e} qvel {worst_v:.2e} against the 16-row form, same sweep count {np.mean(same_it):.3f}")
    assert heavy_seen >= 400 and worst_q <= S24_TOL_Q and worst_v <= S24_TOL_V and np.mean(same_it) >= 0.9
    # against the oracle: one step from the device's state, the envs in 32-row windows
    t, q, v, w = a.get_state(); st0 = a.get_stats()
    a.step(1); _, q1, v1, _ = a.get_state(); st = a.get_stats()
    heavy = np.nonzero((st[:, 1] > 96) & (st[:, 1] <= 128))[0][:24]
    assert len(heavy) >= 8
    for i in heavy:
        d = oracle_s24(m, tab, int(i))
        d.f("qpos")[:] = q[i]; d.f("qvel")[:] = v[i]; d.f("qacc_warmstart")[:] = w[i]; d.f("qacc")[:] = w[i]; d.f("time")[0] = t[i]
        d.step(1)
        if d.i("ncon") != st[i, 0] or d.i("nefc") != st[i, 1]:
            continue
        assert np.abs(q1[i] - d.f("qpos")).max() / max(1.0, np.abs(d.f("qpos")).max()) <= S24_TOL_Q
        assert np.abs(v1[i] - d.f("qvel")).max() / max(1.0, np.abs(d.f("qvel")).max()) <= S24_TOL_V
    a.close(); b.close()


def test_split_api_hands_over_through_the_window_chain():
    """The reference's loop (mj_main.cpp:82-112) on a window model: mjh_step1 [+ mjh_inverse] IS the chain's assemble launch and leaves
    the rows for mjh_step2, which is the window kernel alone.  Same trajectories as with the hand-over switched off
    (MJH_SPLIT_HANDOVER=0 is read once per process, so: against the fused mjh_step, to rounding — the split path renormalises the
    quaternions once more — and bitwise against itself when a getter / a command sits between the halves); state read between the
    halves is what mj_step1 leaves (nothing integrated); a state change between the halves drops the hand-over and the step is still right"""
    m = ms.scene("s24")
    nenv = 1536
    def fresh():
        e = ms.Engine(m, nenv); e.load_s24(); e.set_cohorts(3); return e
    a, b, c = fresh(), fresh(), fresh()
    for k in range(120):
        a.step(1, True)
        b.step1(); b.inverse(); b.step2()
        # the literal loop: a read and a command between the halves
        c.step1(); c.inverse()
        t0, q0, v0, _ = c.get_state(0, 4)
        qj, vj, fj = c.get_joint_state(0, 1)
        c.set_cmd(ddq=np.zeros((1, m.nv)), env0=0)
        c.step2()
        if k in (0, 60):
            t1, q1, _, _ = c.get_state(0, 4)
            assert np.allclose(t1, t0 + 0.005) and np.abs(q1 - q0).max() < 0.05      # the read saw the state BEFORE the integration
    _, qa, va, wa = a.get_state(); _, qb, vb, wb = b.get_state(); _, qc, vc, wc = c.get_state()
    assert np.array_equal(qb, qc) and np.array_equal(vb, vc)                              # getters / commands in between change nothing
    fa = a.get_field("qfrc_inverse"); fb = b.get_field("qfrc_inverse")
    sa, sb = a.get_stats(), b.get_stats()
    same = (sa[:, 0] == sb[:, 0]) & (sa[:, 1] == sb[:, 1])
    assert same.mean() > 0.9
    err_q = np.abs(qa - qb).max(1)[same]; err_v = np.abs(va - vb).max(1)[same]
    print(f"SPLIT-HANDOVER: 120 steps, {same.mean():.3f} of the envs with the same contact sets: qpos {np.median(err_q):.2e} (median) {err_q.max():.2e} (max), qvel {np.median(err_v):.2e} / {err_v.max():.2e}")
    assert np.median(err_q) < 1e-5 and np.isfinite(qb).all()
    assert np.abs(fa - fb)[same].max() < 5e-2 * max(1.0, np.abs(fa).max())
    # a state change between the halves: the hand-over is dropped, mjh_step2 runs the whole chain from the new state
    d = fresh(); d.step(50)
    t, q, v, w = d.get_state()
    d.step1(); d.inverse()
    q2 = q.copy(); q2[:, 2] += 0.01
    d.set_state(qpos=q2, qvel=v, time=t, warmstart=w)
    d.step2()
    e2 = fresh(); e2.step(50); e2.set_state(qpos=q2, qvel=v, time=t, warmstart=w); e2.step1(); e2.step2()
    _, qd, vd, _ = d.get_state(); _, qe, ve, _ = e2.get_state()
    assert np.abs(qd - qe).max() < 1e-4 and np.abs(qd[:, 2] - q2[:, 2]).max() < 0.01
    for x in (a, b, c, d, e2):
        x.close()


def test_per_step_read_write_through_host_mapped_staging_equals_the_copy_path():
    """MjHWInterface::read / write of a few envs per step (mj_hw_interface.cpp:59-91) go through host-mapped staging — one small kernel
    per call, mjh_set_cmd without waiting for the device, a ring of 8 slots — and whole-batch calls through strided copies: same
    values on the device and back.  Arm7 (C3's model) with a PD effort command per env: engine A is commanded env by env (40 calls per
    step: the ring wraps five times without a synchronisation in between) with both ddq and dq of some envs, engine B with two
    whole-batch calls; read back in small ranges and as a whole."""
    g = np.load(os.path.join(G, "arm7_golden.npz"))
    m = ms.scene("arm7", 1)
    nenv = 2048                                        # 2048 x 7 values: beyond the staging slot, the copy path
    rng = np.random.default_rng(4)
    q0 = np.tile(g["q0"], (nenv, 1)) + rng.uniform(-0.05, 0.05, size=(nenv, m.nq))
    def fresh():
        e = ms.Engine(m, nenv); e.set_initial_qpos(q0); e.reset(); e.set_controlled_dofs(np.ones(7, dtype=np.int32)); e.set_cohorts(3); return e
    a, b = fresh(), fresh()
    envs = np.sort(rng.choice(nenv, 40, replace=False))
    for s in range(30):
        qa, va, _ = a.get_joint_state(); qb, vb, _ = b.get_joint_state()
        assert np.array_equal(qa, qb) and np.array_equal(va, vb)
        ddq = np.zeros((nenv, 7)); dq = np.zeros((nenv, 7))
        ddq[envs] = 200.0 * (g["target"] - qa[envs]) - 50.0 * va[envs]
        dq[envs[::4], 6] = 0.3                         # a velocity command on the last joint of every fourth of them
        for i in envs:
            if i in envs[::4]: a.set_cmd(ddq=ddq[i:i+1], dq=dq[i:i+1], env0=int(i))
            else: a.set_cmd(ddq=ddq[i:i+1], env0=int(i))
        b.set_cmd(ddq=ddq, dq=dq)
        if s % 2: a.step(1, True); b.step(1, True)
        else:
            for x in (a, b): x.step1(); x.inverse()
            if s == 0:      # between the halves: the velocity command has overridden qvel of the controlled dof (mj_sim.cpp:1066-1070)
                for i in envs[::4]:
                    assert a.get_joint_state(int(i), 1)[1][0, 6] == np.float32(0.3) == b.get_joint_state(int(i), 1)[1][0, 6]
                assert np.abs(a.get_joint_state(int(envs[1]), 1)[1][0, 6]) < 0.05
            for x in (a, b): x.step2()
        # small ranges (staging) against the whole batch (copies), on either engine
        k = int(envs[s % len(envs)]); k = min(k, nenv - 3)
        q3, v3, f3 = a.get_joint_state(k, 3); qw, vw, fw = b.get_joint_state()
        assert np.array_equal(q3, qw[k:k+3]) and np.array_equal(v3, vw[k:k+3]) and np.array_equal(f3, fw[k:k+3])
        q1, v1, f1 = b.get_joint_state(k, 1)
        assert np.array_equal(q1, qw[k:k+1]) and np.array_equal(f1, fw[k:k+1])
    qa, va, fa = a.get_joint_state(); qb, vb, fb = b.get_joint_state()
    assert np.array_equal(qa, qb) and np.array_equal(va, vb) and np.array_equal(fa, fb)
    moved = np.abs(qa[envs] - q0[envs]).max(1)
    assert moved.min() > 1e-3 and np.isfinite(qa).all()          # the commanded envs did follow their commands
    a.close(); b.close()


_TIER_SCRIPT = r'''
import sys, os, hashlib
sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, sys.argv[1])
import numpy as np
import test_gpu_round4 as T
m, e, tab = T._s24d(768)
e.set_cohorts(3); e.step(260); e.synchronize()
_, q, v, w = e.get_state(); st = e.get_stats()
print("TIER", hashlib.sha256(q.tobytes() + v.tobytes() + w.tobytes() + st[:, :3].tobytes()).hexdigest(), int(st[:, 1].max()), int((st[:, 1] > 96).sum()))
'''


def test_window_tiers_hold_the_same_values_wherever_a_window_waits():
    """mjh_window_kernel keeps six windows of an env in registers; the later ones wait in LDS (as many as the launch was given) or in the
    env's global slice.  The host picks the LDS tier's size from an UNSYNCHRONISED hint (the cohort's largest row count of some steps
    ago, engine.hip launch_on) — allowed only because the choice changes where a window waits, never what is computed.  S24D (up to 16
    windows per env) with the LDS tier forced off (MJH_WN_NL=0: everything beyond the registers in global memory), forced to 3 and to
    its full size, and chosen by the hint (default): the same bits after 260 steps.  MJH_WN_NL is read once per process: subprocesses."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for nl in ("default", "0", "3", "10"):
        env = dict(os.environ)
        env.pop("MJH_WN_NL", None)
        env["MJH_WINDOW64"] = "0"          # (the 64-row section keeps tiles in the LDS tier and is switched off with it: this test is about the 16-row form's tiers)
        if nl != "default":
            env["MJH_WN_NL"] = nl
        r = subprocess.run([sys.executable, "-c", _TIER_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("TIER ")]
        assert r.returncode == 0 and lines, r.stderr[-2000:]
        out[nl] = lines[-1].split()
    print("WINDOW-TIERS:", {k: (v[1][:12], v[2], v[3]) for k, v in out.items()})
    assert int(out["default"][2]) > 128 and int(out["default"][3]) > 100        # envs beyond the register-resident windows and the 32-row section
    assert len({v[1] for v in out.values()}) == 1, out
