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


# ------------------------------------------------------------------ BASELINE-size runs of C2 / C3 / C4 with invariants
def _robot(name):
    from mujoco_sim_amd.tables import load_model_tables
    return load_model_tables(os.path.join(G, f"robot_{name}.npz"))


def test_c2_full_size_invariants():
    """C2 as SURVEY.md §8-d D3 states it — 64 boxes with per-env random sizes / orientations / jitter, 4096 envs — settled
    200 steps + 60 more: finite, no capacity overflow, no bad-state reset, unit quaternions, nothing below the floor,
    Newton's second law on the vertical axis of every env (ties qacc, qfrc_constraint and the per-env masses together),
    total energy not increasing once the boxes have landed"""
    m = ms.scene("boxpile", 64); m.c.maxcon = 600; m.c.maxefc = 2400
    nenv = 4096
    e = ms.Engine(m, nenv)
    tab = e.load_tables(ms.boxes_randomize(m, 0, nenv, jitter=0.01))
    assert np.abs(tab["geom_size"][0] - tab["geom_size"][1]).max() > 1e-3          # per-env sizes really differ
    e.step(200)
    e.forward(); E0 = e.get_field("energy").sum(axis=1)
    e.step(60)
    e.forward(); E1 = e.get_field("energy").sum(axis=1)
    t, q, v, _ = e.get_state(); st = e.get_stats()
    assert np.isfinite(q).all() and np.isfinite(v).all()
    assert (st[:, 3] == 0).all(), f"flags {np.unique(st[:, 3])}: overflow / reset at capacity 600"
    pos = q.reshape(nenv, 64, 7)
    np.testing.assert_allclose(np.linalg.norm(pos[:, :, 3:], axis=-1), 1, atol=1e-5)
    assert pos[:, :, 2].min() > 0.02                                           # smallest half-extent 0.05, penetration of millimetres
    assert np.mean(E1 <= E0 + 1e-3 * np.abs(E0)) > 0.98
    fz = e.get_field("qfrc_constraint").reshape(nenv, 64, 6)[:, :, 2]; az = e.get_field("qacc").reshape(nenv, 64, 6)[:, :, 2]
    mb = tab["body_mass"][:, 1:]
    lhs, rhs = fz.sum(axis=1), (mb * (az + 9.81)).sum(axis=1); wtot = 9.81 * mb.sum(axis=1)
    rel = np.abs(lhs - rhs) / wtot
    assert np.median(rel) < 2e-4 and np.quantile(rel, 0.99) < 2e-2, (np.median(rel), np.quantile(rel, 0.99))
    print(f"C2 4096 envs: mean ncon {st[:, 0].mean():.0f} max {st[:, 0].max()}, mean nefc {st[:, 1].mean():.0f} max {st[:, 1].max()}, mean sweeps {st[:, 2].mean():.0f}")
    assert 100 < st[:, 0].mean() < 600
    e.close()


def test_c3_full_size_invariants():
    """C3 at 8192 envs: the in-engine PD law drives every arm to its own random in-range target; afterwards every joint is
    inside its limits (up to the solref penetration), velocities are small, qfrc_inverse is finite and — the arms being at
    rest under gravity compensation + computed torque — close to the applied torque"""
    m = ms.scene("arm7", 1)
    nenv = 8192
    e = ms.Engine(m, nenv)
    e.set_controlled_dofs(np.ones(m.nv, dtype=np.int32))
    rng = np.random.default_rng(11)
    lo, hi = m.array("jnt_range").reshape(-1, 2).T
    target = rng.uniform(lo + 0.05, hi - 0.05, size=(nenv, m.nv))
    e.set_pd_controller(200.0, 50.0); e.set_pd_target(target)
    e.step(600, True)
    q, v, f = e.get_joint_state(); st = e.get_stats()
    assert np.isfinite(q).all() and np.isfinite(v).all() and np.isfinite(f).all()
    assert (st[:, 3] == 0).all()
    assert (q > lo - 2e-2).all() and (q < hi + 2e-2).all()
    assert np.abs(v).max() < 0.2, np.abs(v).max()
    e.step1(); e.inverse()                       # split API: qfrc_passive / qfrc_applied / qfrc_inverse of THIS step are readable
    # Where the arms come to rest: the wrapper compensates gravity twice (gravcomp = 1 in qfrc_passive AND qfrc_bias added on
    # the controlled dofs, mj_sim.cpp:301-310,1058-1063 — reproduced literally), so the PD law settles at the offset where
    # M ddq = -qfrc_passive, not at the target.  Checked with the engine's own mj_mulM and qfrc_passive on every env.
    ddq = 200.0 * (target - q) - 50.0 * v
    lhs = e.mulM(ddq); rhs = -e.get_field("qfrc_passive")
    scale = np.abs(rhs).max(axis=1, keepdims=True) + 1e-3
    assert np.quantile(np.abs(lhs - rhs) / scale, 0.99) < 0.05, np.quantile(np.abs(lhs - rhs) / scale, 0.99)
    assert np.quantile(np.abs(q - target), 0.99) < 0.6
    # mj_inverse's contract (mj_hw_interface.cpp:61-69) on every env: with the limits inactive, qfrc_inverse of the step equals
    # the torque the controller applied, qfrc_applied = M ddq + bias (split API: the hand-over vectors are then readable)
    fa, fi = e.get_field("qfrc_applied"), e.get_field("qfrc_inverse")
    e.step2()
    free = (st[:, 1] == 0)
    assert free.mean() > 0.9
    np.testing.assert_allclose(fi[free], fa[free], rtol=0, atol=2e-2 * max(1.0, np.abs(fa).max()))
    e.close()


def test_c4_full_size_invariants_with_runtime_spawn_and_destroy():
    """C4 at 2048 envs: PR2 (37 mesh geoms as hulls) on the reference floor + the object pool, slots spawned / destroyed at
    run time in a different pattern per env (reference shape: test/test_spawn_and_destroy_pr2.py:25-42,69-80): finite, no
    overflow / reset, the robot keeps standing (base height), destroyed slots stay frozen, spawned objects end on the floor"""
    m, z = _robot("c4_pr2_world_objects_mesh")
    lib = ms.capi.load()
    names = [lib.mjh_id2name(m.ptr, 0, b).decode() for b in range(m.c.nbody)]
    slots = [b for b, n in enumerate(names) if n.startswith("object_")]
    nenv = 2048
    e = ms.Engine(m, nenv)
    e.set_controlled_dofs(z["controlled"].astype(np.int32))
    for b in slots:
        e.set_slot_active(b, False)
    rng = np.random.default_rng(4)
    alive = np.zeros((nenv, len(slots)), dtype=bool)
    base_z0 = None
    jq = m.array("jnt_qposadr"); bj = m.array("body_jntadr")
    for rnd in range(6):
        for i in rng.choice(nenv, nenv // 16, replace=False):
            k = int(rng.integers(len(slots)))
            if alive[i, k]:
                e.set_slot_active(slots[k], False, env0=int(i), n=1); alive[i, k] = False
            else:
                a, r = rng.uniform(-np.pi, np.pi), rng.uniform(0.9, 1.5)
                e.set_slot_active(slots[k], True, env0=int(i), n=1)
                e.set_body_pose(int(i), slots[k], [r * np.sin(a), r * np.cos(a), 1.0], [1, 0, 0, 0], [0, 0, -0.5, 0.3, 0.2, 0.1])
                alive[i, k] = True
        e.step(100, True)
        t, q, v, _ = e.get_state(); st = e.get_stats()
        assert np.isfinite(q).all() and np.isfinite(v).all(), rnd
        assert (st[:, 3] == 0).all(), (rnd, np.unique(st[:, 3]))
        if base_z0 is None:
            base_z0 = q[:, 2].copy()
        assert np.abs(q[:, 2] - base_z0).max() < 0.02                       # the PR2 keeps standing on its casters
        for k, b in enumerate(slots):
            qa = jq[bj[b]]; da = m.array("body_dofadr")[b]
            dead = ~alive[:, k]
            assert (v[dead][:, da:da + 6] == 0).all()                        # destroyed slots are frozen
            if rnd >= 2 and alive[:, k].any():
                zs = q[alive[:, k], qa + 2]
                assert zs.min() > 0.0 and np.median(zs) < 0.5                 # spawned objects fell onto the floor, none through it
    print(f"C4 2048 envs: objects alive per env {alive.sum(1).mean():.2f}, mean ncon {st[:, 0].mean():.1f} max {st[:, 0].max()}, mean nefc {st[:, 1].mean():.1f}")
    e.close()


@pytest.mark.parametrize("layout", [1, 2], ids=["lds-resident", "global-pools"])
def test_c4_pr2_spawn_destroy_schedule_matches_the_oracle(layout, lib):
    """C4 under the oracle: on the PR2 + world + object-pool fixture, objects are spawned (pose + twist, mj_ros.cpp:1406-1412)
    and destroyed per env on a schedule (reference shape: test/test_spawn_and_destroy_pr2.py:25-42,69-80), mirrored in the
    oracle through orc_set_slot_mask; segments of 25 steps, each starting from identical states (contacts of round objects
    dropped beside a 49-dof robot fork over long horizons).  Both memory layouts."""
    from test_robot_fixtures import robot_command
    m, z = _robot("c4_pr2_world_objects_mesh")
    lib.mjh_set_layout_policy(layout)
    try:
        names = [lib.mjh_id2name(m.ptr, 0, b).decode() for b in range(m.c.nbody)]
        slots = [b for b, n in enumerate(names) if n.startswith("object_")]
        sbase = m.c.nbody - 32 if m.c.nbody > 32 else 0
        nenv = 4
        e = ms.Engine(m, nenv)
        e.set_controlled_dofs(z["controlled"].astype(np.int32))
        ds = [orc.OrcData(m.ptr) for _ in range(nenv)]
        for d in ds:
            d.ifield("controlled")[:] = z["controlled"]
        mask = [0] * nenv
        for b in slots:
            e.set_slot_active(b, False)
            for i in range(nenv):
                mask[i] |= 1 << (b - sbase)
        for i, d in enumerate(ds):
            d.L.orc_set_slot_mask(d.d, mask[i])
        rng = np.random.default_rng(21)
        jq, bj, bd = m.array("jnt_qposadr"), m.array("body_jntadr"), m.array("body_dofadr")
        nrobot = int(min(bd[b] for b in slots))                        # dofs in front of the object pool
        order = [rng.permutation(len(slots)) for _ in range(nenv)]
        worst = 0.0; step = 0; forked = 0
        for rnd in range(7):
            for i, d in enumerate(ds):
                if rnd < 5:                                           # spawn next to the robot, with a twist
                    b = slots[int(order[i][rnd])]
                    a, r = rng.uniform(-np.pi, np.pi), rng.uniform(0.9, 1.3)
                    pos = np.array([r * np.sin(a), r * np.cos(a), 0.6]); quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
                    vel = np.array([0.1, -0.1, -0.3, *(0.5 * rng.normal(size=3))])
                    e.set_slot_active(b, True, env0=i, n=1); e.set_body_pose(i, b, pos, quat, vel)
                    mask[i] &= ~(1 << (b - sbase)); d.L.orc_set_slot_mask(d.d, mask[i])
                    qa, da = jq[bj[b]], bd[b]
                    d.f("qpos")[qa:qa + 3] = pos; d.f("qpos")[qa + 3:qa + 7] = quat; d.f("qvel")[da:da + 6] = vel
                else:                                                 # destroy the first arrivals again
                    b = slots[int(order[i][rnd - 5])]
                    e.set_slot_active(b, False, env0=i, n=1)
                    mask[i] |= 1 << (b - sbase); d.L.orc_set_slot_mask(d.d, mask[i])
            same = np.ones(nenv, dtype=bool)          # the env's contact-set history equals the oracle's over this segment
            for k in range(25):
                step += 1
                cmd = robot_command(m, step)
                e.set_cmd(ddq=np.tile(cmd, (nenv, 1))); e.step(1, True)
                for d in ds:
                    d.f("ddq")[:] = cmd; d.step(1, 1)
                sk = e.get_stats()
                same &= (sk[:, 0] == np.array([d.i("ncon") for d in ds])) & (sk[:, 1] == np.array([d.i("nefc") for d in ds]))
            _, q, v, _ = e.get_state(); st = e.get_stats()
            assert (st[:, 3] == 0).all() and all(d.i("warn") == 0 for d in ds), (rnd, st[:, 3])
            oq = np.array([d.f("qpos") for d in ds]); ov = np.array([d.f("qvel") for d in ds])
            err = np.abs(q - oq).max(axis=1)
            worst = max(worst, float(err[same].max())) if same.any() else worst
            forked += int((~same).sum())
            # an env is held to the tolerance as long as it sees the oracle's contacts; one whose contact history differs in the
            # segment (a round object touching down a step earlier or later) is only required to stay in the neighbourhood
            assert (err[same] < 5e-3).all() and err.max() < 0.5, (rnd, err, same)
            fi = e.get_field("qfrc_inverse")
            for i, d in enumerate(ds):
                if same[i]:
                    # what read() hands to ros_control: the ROBOT's joints (the pool objects' free dofs carry impact forces whose
                    # inverse is a difference of large numbers)
                    ref = d.f("qfrc_inverse")[:nrobot]
                    np.testing.assert_allclose(fi[i][:nrobot], ref, rtol=0, atol=2e-2 * max(1.0, np.abs(ref).max()))
                for b in slots:                                       # destroyed slots: frozen on both sides
                    if mask[i] >> (b - sbase) & 1:
                        assert (v[i, bd[b]:bd[b] + 6] == 0).all() and (ov[i, bd[b]:bd[b] + 6] == 0).all()
            e.set_state(qpos=oq, qvel=ov, warmstart=np.array([d.f("qacc_warmstart") for d in ds]))
        assert max(d.i("ncon") for d in ds) >= 16
        assert forked <= 3, forked                    # of 28 (env, segment) pairs
        print(f"C4 spawn/destroy schedule vs oracle (layout {layout}): worst |dqpos| over 7 segments {worst:.2e}; {forked} of 28 (env, segment) pairs saw a different contact history")
        e.close()
    finally:
        lib.mjh_set_layout_policy(0)


# ------------------------------------------------------------------ multi-GPU from the C host (mjh_group_*)
def _group_vs_single(devices, nenv, transport, lib):
    m = ms.scene("s24")
    lib.mjh_group_set_transport(transport)
    try:
        g = ms.Group(m, nenv, devices)
    finally:
        lib.mjh_group_set_transport(0)
    single = ms.Engine(m, nenv); single.load_s24()
    assert sum(n for _, n in g.ranges) == nenv and g.ranges[0][0] == 0
    for (e0, n), e in zip(g.ranges, g.engines):
        assert e.nenv == n
        e.load_s24(env_offset=e0)                       # every shard draws ITS envs' boxes: env ids are global
    for e0 in (0, nenv - 1):
        r, l = g.locate(e0)
        assert g.ranges[r][0] + l == e0
    for k in range(4):
        g.step(15, k % 2 == 1); single.step(15, k % 2 == 1)
        pub = g.publish()
        t, q, v, _ = single.get_state()
        ref = np.concatenate([t[:, None], q, v], axis=1).astype(np.float32)
        assert pub.shape == ref.shape
        assert np.array_equal(pub, ref), (k, np.abs(pub - ref).max())          # env order + bitwise state across the shards
    # the split API over the group = the fused one
    g.step1(); g.inverse(); g.step2(); single.step(1, True)
    assert np.array_equal(g.publish()[:, 1:], np.concatenate([single.get_state()[1], single.get_state()[2]], axis=1).astype(np.float32))
    used = g.uses_rccl
    g.close(); single.close()
    return used


def test_group_of_one_device_equals_the_plain_engine_and_gathers_through_rccl(lib):
    """E2: mjh_group_create / step / publish on a world of ONE device (all this box has) is bitwise the plain engine, and the
    publish goes through ncclAllGather when RCCL can be loaded (communicator of size 1: same code path as on 8 GPUs)"""
    used = _group_vs_single([0], 2100, 0, lib)          # 2100 envs: two cohorts per engine, the forked export path
    print("group publish transport:", "RCCL ncclAllGather" if used else "peer copies (RCCL not loadable here)")


def test_group_of_two_shards_keeps_env_order_with_uneven_shares(lib):
    """two shards (both on device 0: RCCL refuses duplicate devices, so this takes the peer-copy transport) with an ODD env
    count: shares 24 + 23, the gathered slots carry padding behind the smaller share and are compacted into env order"""
    used = _group_vs_single([0, 0], 47, 1, lib)
    assert not used


def test_host_simulate_over_a_group_equals_the_single_engine_loop(lib):
    """host_sim.cpp: simulate() with MjhSim::group set — every shard stepped with the split API, the ROS surface attached to a
    GLOBAL env that lives on the second shard, the state slice all-gathered every 3 steps (60 Hz at dt = 5 ms) — gives the
    attached env the trajectory of the single-engine loop, and the published slice is the state of ALL envs in env order"""
    import ctypes as C
    from mujoco_sim_amd import capi
    g = np.load(os.path.join(G, "arm7_golden.npz"))
    m = ms.scene("arm7", 1)
    nenv, att = 5, 3
    tgt = np.ascontiguousarray(g["target"], dtype=np.float64)
    single = ms.Engine(m, nenv); single.set_initial_qpos(np.tile(g["q0"], (nenv, 1))); single.reset()
    q1 = np.zeros(7); f1 = np.zeros(7); rtf = C.c_double(0)
    assert lib.mjh_host_run_pd(single.h, att, capi.dptr(tgt), 200.0, 50.0, 300, capi.dptr(q1), capi.dptr(f1), C.byref(rtf)) == 0
    lib.mjh_group_set_transport(1)
    try:
        grp = ms.Group(m, nenv, [0, 0])
    finally:
        lib.mjh_group_set_transport(0)
    for (e0, n), e in zip(grp.ranges, grp.engines):
        e.set_initial_qpos(np.tile(g["q0"], (n, 1))); e.reset()
    q2 = np.zeros(7); f2 = np.zeros(7); pub = np.zeros((nenv, grp.stride), dtype=np.float32)
    assert lib.mjh_host_run_pd_group(grp.h, att, capi.dptr(tgt), 200.0, 50.0, 300, 3, capi.dptr(q2), capi.dptr(f2), pub.ctypes.data_as(C.POINTER(C.c_float))) == 0
    assert np.array_equal(q1, q2) and np.array_equal(f1, f2)
    np.testing.assert_allclose(q2, g["step300_qpos"], atol=2e-3)
    t, q, v, _ = single.get_state()
    assert np.array_equal(pub, np.concatenate([t[:, None], q, v], axis=1).astype(np.float32))
    assert np.abs(pub[att, 1:8] - pub[0, 1:8]).max() > 1e-3          # only the attached env was commanded
    grp.close(); single.close()


def test_simulate_real_time_spin_and_adaptive_timestep(lib):
    """A17 (mj_main.cpp:115-163) executed, not only compiled: (i) a cheap simulation is held at real time by the wall-clock spin
    (RTF <= 1: 100 steps of 5 ms take >= 0.5 s of wall time); (ii) a simulation that cannot keep up — timestep 20 us, far below
    the cost of one host round trip — doubles its timestep while it lags by > 1 ms, never beyond max_time_step, and halves it
    back when it has caught up"""
    import ctypes as C
    from mujoco_sim_amd import capi
    m = ms.scene("arm7", 1)
    tgt = np.zeros(7); tgt[3] = -1.0
    st = np.zeros(6)
    e = ms.Engine(m, 1)
    assert lib.mjh_host_run_realtime(e.h, 0, capi.dptr(tgt), 200.0, 50.0, 100, 0.005, capi.dptr(st)) == 0
    sim_time, wall, rtf, final_dt, steps, changes = st
    assert abs(sim_time - 0.5) < 1e-9 and wall >= 0.5 - 1e-3, (sim_time, wall)
    assert sim_time / wall <= 1.0 + 2e-3 and wall < 0.6                 # spun, not slept past: RTF just below 1
    assert final_dt == 0.005 and changes == 0                           # in sync: the timestep is left alone
    assert abs(e.get_state()[0][0] - 0.5) < 1e-9
    e.close()
    # (ii) too slow for real time at the configured step
    m2 = ms.scene("arm7", 1); m2.c.opt.timestep = 2.0e-5
    e = ms.Engine(m2, 1)
    assert lib.mjh_host_run_realtime(e.h, 0, capi.dptr(tgt), 200.0, 50.0, 3000, 1.0e-3, capi.dptr(st)) == 0
    sim_time, wall, rtf, final_dt, steps, changes = st
    assert changes >= 4, st                                             # doubled several times (and possibly halved back)
    assert 2.0e-5 <= final_dt <= 2.0e-3 + 1e-12, final_dt               # the reference tests dt < max before doubling: at most 2 x max
    ratio = np.log2(final_dt / 2.0e-5)
    assert abs(ratio - round(ratio)) < 1e-9                             # only ever multiplied / divided by two
    assert sim_time > 3000 * 2.0e-5 * 1.5                               # it did integrate with larger steps
    assert abs(e.get_state()[0][0] - sim_time) < 1e-9                   # the device clock followed every change of dt
    assert sim_time / wall <= 1.0 + 1e-2
    assert abs(e.timestep - final_dt) < 1e-15
    e.close()


# ------------------------------------------------------------------ xfrc_applied, sensors, mocap + weld / connect (B1, F4, F1)
def _feature_scene(lib):
    """floor + a free base box carrying an upper body on a limited slide joint (force + torque sensors at a rotated site on it) +
    a free box welded (torquescale 0.9) to its mocap clone + a free box connected to the base: every new row type at once"""
    from helpers import D, set_opt
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.002)
    lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, -1, -1, -1, -1)
    base = lib.mjh_builder_add_body(b, b"base", 0, D(0, 0, 0.0995), None, 0.0)
    lib.mjh_builder_add_joint(b, b"base_free", base, 0, None, None, None, 0, 0, 0, 0, 0)
    lib.mjh_builder_add_geom(b, b"bg", base, 6, D(0.2, 0.2, 0.1), None, None, None, -1, -1, -1, -1)
    up = lib.mjh_builder_add_body(b, b"upper", base, D(0, 0, 0.3), None, 0.0)
    lib.mjh_builder_add_joint(b, b"slide", up, 2, None, D(0, 0, 1), D(0.0, 0.5), 0, 0, 0, 0, 0)
    lib.mjh_builder_set_inertial(b, up, 1.5, D(0.02, 0, 0.01), None, D(0.01, 0.02, 0.015))
    s1 = lib.mjh_builder_add_site(b, b"s_up", up, D(0.01, 0.02, -0.03), D(0.9, 0.1, -0.3, 0.2))
    s0 = lib.mjh_builder_add_site(b, b"s_base", base, D(0, 0, 0.05), None)
    for k, (t, s) in enumerate(((4, s1), (5, s1), (4, s0), (5, s0))):
        lib.mjh_builder_add_sensor(b, b"sens%d" % k, t, s)
    cube = lib.mjh_builder_add_body(b, b"cube", 0, D(1.0, 0, 0.6), None, 0.0)
    lib.mjh_builder_add_joint(b, b"cube_free", cube, 0, None, None, None, 0, 0, 0, 0, 0)
    lib.mjh_builder_add_geom(b, b"cg", cube, 6, D(0.1, 0.1, 0.1), None, None, None, -1, -1, -1, -1)
    ref = lib.mjh_builder_add_body(b, b"cube_ref", 0, D(1.0, 0, 0.6), None, 0.0)
    lib.mjh_builder_add_geom(b, b"rg", ref, 6, D(0.1, 0.1, 0.1), None, None, None, -1, 0, 0, -1)
    lib.mjh_builder_set_mocap(b, ref)
    lib.mjh_builder_add_eq_weld(b, cube, ref, None, 0.9)
    bob = lib.mjh_builder_add_body(b, b"bob", 0, D(0.0, 0.5, 0.5), None, 0.0)
    lib.mjh_builder_add_joint(b, b"bob_free", bob, 0, None, None, None, 0, 0, 0, 0, 0)
    lib.mjh_builder_add_geom(b, b"bb", bob, 2, D(0.08, 0, 0), None, None, None, -1, -1, -1, -1)
    lib.mjh_builder_add_eq_connect(b, bob, base, D(0.0, -0.25, 0.0))
    lib.mjh_builder_set_capacity(b, 24, 24 * 4 + 12)
    m = ms.Model(lib.mjh_builder_compile(b), lib); lib.mjh_builder_destroy(b)
    return m


@pytest.mark.parametrize("layout", [1, 2], ids=["lds-resident", "global-pools"])
def test_sensors_mocap_weld_connect_and_xfrc_match_the_oracle(layout, lib):
    m = _feature_scene(lib)
    assert (m.nsensor, m.nmocap, m.neq) == (4, 1, 2)
    lib.mjh_set_layout_policy(layout)
    try:
        nenv = 6
        e = ms.Engine(m, nenv)
    finally:
        lib.mjh_set_layout_policy(0)
    ds = [orc.OrcData(m.ptr) for _ in range(nenv)]
    rng = np.random.default_rng(3)
    tp = np.array([1.0, 0, 0.6]) + rng.uniform(-0.3, 0.3, size=(nenv, 3))
    tq = rng.normal(size=(nenv, 4)); tq[:, 0] += 3.0; tq /= np.linalg.norm(tq, axis=1, keepdims=True)
    xf = np.zeros((nenv, m.nbody, 6))
    up, bob = m.name2id(0, "upper"), m.name2id(0, "bob")
    xf[:, up, 2] = rng.uniform(0, 6, nenv); xf[:, up, 3:] = rng.uniform(-0.2, 0.2, (nenv, 3)); xf[:, bob, :3] = rng.uniform(-1, 1, (nenv, 3))
    e.reset(); e.set_mocap_pose(0, tp, tq); e.set_xfrc_applied(xf)
    np.testing.assert_allclose(e.get_xfrc_applied(), xf, atol=1e-6)
    for i, d in enumerate(ds):
        d.call("reset"); d.f("mocap_pos")[:] = tp[i]; d.f("mocap_quat")[:] = tq[i]; d.f("xfrc_applied")[:] = xf[i].reshape(-1)
    for seg in range(6):
        e.step(50, True); [d.step(50, 1) for d in ds]
        _, q, v, _ = e.get_state(); st = e.get_stats(); sd = e.get_sensordata()
        oq = np.array([d.f("qpos") for d in ds]); ov = np.array([d.f("qvel") for d in ds]); osd = np.array([d.f("sensordata") for d in ds])
        assert (st[:, 3] == 0).all() and all(d.i("warn") == 0 for d in ds)
        assert (st[:, 1] == np.array([d.i("nefc") for d in ds])).all(), (seg, st[:, 1], [d.i("nefc") for d in ds])
        np.testing.assert_allclose(q, oq, atol=2e-3, err_msg=f"segment {seg}")
        np.testing.assert_allclose(v, ov, atol=2e-2, err_msg=f"segment {seg}")
        scale = max(1.0, np.abs(osd).max())
        np.testing.assert_allclose(sd, osd, atol=2e-2 * scale, err_msg=f"sensordata, segment {seg}")
        e.set_state(qpos=oq, qvel=ov, warmstart=np.array([d.f("qacc_warmstart") for d in ds]))
    # the welded cube arrived at ITS env's mocap pose, the upper body's force sensor sees weight minus the pull
    cq = m.array("jnt_qposadr")[m.array("body_jntadr")[m.name2id(0, "cube")]]
    np.testing.assert_allclose(oq[:, cq:cq + 3], tp + np.array([0, 0, 0.0]), atol=5e-3)
    assert np.abs(osd).max() > 1.0
    e.close()


def test_xfrc_applied_survives_a_model_change_and_zero_xfrc_changes_nothing(lib):
    """B1: add_old_state() copies xfrc_applied per body across a recompile (mj_sim.cpp:496-500); an all-zero xfrc array routes
    the S24 kernel through the feature-carrying instance without changing the trajectory beyond fp32 rounding"""
    m = ms.scene("s24")
    a = ms.Engine(m, 8); a.load_s24(); b = ms.Engine(m, 8); b.load_s24()
    b.set_xfrc_applied(np.zeros((8, m.nbody, 6)))
    a.step(120); b.step(120)
    np.testing.assert_allclose(a.get_state()[1], b.get_state()[1], atol=2e-3)
    xf = np.zeros((8, m.nbody, 6)); xf[:, 2, 2] = 3.0; xf[:, 4, 3] = 0.1
    b.set_xfrc_applied(xf)
    c = ms.Engine(m, 8); c.load_s24()
    assert c.transplant_state_from(b) == 4
    np.testing.assert_allclose(c.get_xfrc_applied(), xf, atol=1e-7)
    b.step(1); c.step(1)
    assert np.array_equal(b.get_state()[1], c.get_state()[1])
    a.close(); b.close(); c.close()


def test_batched_spawn_destroy_equals_the_per_object_calls():
    """mjh_spawn_objects / mjh_destroy_objects (one call per service request, lists of (env, body)) leave exactly the state the
    per-object mjh_set_slot_active + mjh_set_body_pose calls do"""
    m = ms.scene("s24")
    a = ms.Engine(m, 32); a.load_s24(); b = ms.Engine(m, 32); b.load_s24()
    rng = np.random.default_rng(9)
    a.step(40); b.step(40)
    envs = rng.choice(32, 12, replace=False); bodies = rng.integers(1, 5, 12)
    a.destroy_objects(envs, bodies)
    for e_, b_ in zip(envs, bodies):
        b.set_slot_active(int(b_), False, env0=int(e_), n=1)
    a.step(30); b.step(30)
    pos = rng.uniform(-0.05, 0.05, (12, 3)) + np.array([0, 0, 1.2]); quat = rng.normal(size=(12, 4)); quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    vel = rng.normal(size=(12, 6)) * 0.3
    a.spawn_objects(envs, bodies, pos, quat, vel)
    for k, (e_, b_) in enumerate(zip(envs, bodies)):
        b.set_slot_active(int(b_), True, env0=int(e_), n=1); b.set_body_pose(int(e_), int(b_), pos[k], quat[k], vel[k])
    a.step(60); b.step(60)
    for x, y in zip(a.get_state(), b.get_state()):
        assert np.array_equal(x, y)
    assert np.array_equal(a.get_stats(), b.get_stats())
    a.close(); b.close()


# ----------------------------------------------------------------------------- contact-patch sweep (csrc/patch_pgs.h)
@pytest.mark.gpu
def test_solver_order_query_matches_the_model_class():
    """mjh_solver_order: 2 (mj_solPGS's row order) by default for every model; mjh_patch_sweep: 1 for small free-body models (S24,
    cube pools), 0 for articulated ones; mjh_set_pgs_row_order(0) brings the legacy orders back (1: patches, 0: pairs / groups)"""
    e = ms.Engine(ms.scene("s24"), 4); assert e.solver_order() == 2 and e.patch_sweep() == 1 and e.pgs_schedule() == 1; e.close()
    e = ms.Engine(ms.scene("arm7", 1), 4); assert e.solver_order() == 2 and e.patch_sweep() == 0; e.close()
    e = ms.Engine(ms.scene("pendulum"), 4); assert e.solver_order() == 2 and e.patch_sweep() == 0; e.close()
    lib = ms.capi.load()
    lib.mjh_set_pgs_row_order(0)
    try:
        e = ms.Engine(ms.scene("s24"), 4); assert e.solver_order() == 1 and e.pgs_schedule() == 0; e.close()
        e = ms.Engine(ms.scene("arm7", 1), 4); assert e.solver_order() == 0; e.close()
    finally:
        lib.mjh_set_pgs_row_order(1)


@pytest.mark.gpu
def test_patch_sweep_with_condim_1_3_4_contacts_matches_oracle(lib):
    """five free bodies (nv = 30) with frictionless (condim 1), condim-3 and condim-4 geoms stacked in a corner: patches mix 1-, 4-
    and 6-row contacts, one- and two-body patches, partially filled patches; device (patch sweep) vs oracle (patch order)"""
    from test_gpu_parity import _compare_rollout
    from helpers import D, set_opt
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.004)
    lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, 4, -1, -1, -1)           # condim 4 like empty.xml's floor
    lib.mjh_builder_add_geom(b, b"wall", 0, 6, D(0.02, 0.6, 0.3), D(-0.25, 0, 0.3), None, None, 3, -1, -1, -1)
    specs = [(b"box_a", 6, (0.10, 0.10, 0.05), (0.0, 0.0, 0.06), 3), (b"box_b", 6, (0.08, 0.08, 0.05), (0.02, 0.01, 0.18), 4),
             (b"ball", 2, (0.06, 0, 0), (0.03, -0.02, 0.31), 1), (b"caps", 3, (0.04, 0.10, 0), (-0.12, 0.0, 0.12), 3),
             (b"box_c", 6, (0.05, 0.12, 0.04), (0.16, 0.02, 0.05), 1)]
    for name, gt, size, pos, condim in specs:
        bd = lib.mjh_builder_add_body(b, name, 0, D(*pos), None, 0.0)
        lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
        lib.mjh_builder_add_geom(b, None, bd, gt, D(*size), None, None, None, condim, -1, -1, -1)
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    assert m.nv == 30
    m.c.maxcon = 48; m.c.maxefc = 48 * 6          # (the default capacity is every pair at full manifold: beyond the patch sweep's 64 contacts)
    e = ms.Engine(m, 2); assert e.solver_order() == 2 and e.patch_sweep() == 1; e.close()
    q0 = m.array("qpos0").copy()
    q0[7*3+3:7*3+7] = [np.cos(0.6), 0, np.sin(0.6), 0]          # lean the capsule against the wall
    v0 = np.zeros(m.nv); v0[0] = -0.3; v0[6 + 1] = 0.2
    st, ncon, nefc = _compare_rollout(m, q0, [1, 50, 150], [1e-5, 1e-3, 3e-2], v0=v0)
    assert ncon >= 8 and st[0] == ncon and st[1] == nefc and nefc > ncon     # rows: a mix of 1, 4 and 6 per contact


@pytest.mark.gpu
def test_patch_pool_overflow_drops_patches_and_raises_the_capacity_flag():
    """the patch pool takes the LDS span that is dead when the sweeps start and no more: patches beyond it are dropped with the
    capacity flag, like contacts beyond maxcon.  Forced here with a pool of 600 floats (MJH_PATCH_POOL_FLOATS): settled S24 piles
    need 1500-3000.  (The fused kernel's patch sweep: mjh_set_window_solver(0) — the window sweep of mjh_step has no pool.)"""
    m = ms.scene("s24")
    full = ms.Engine(m, 64); tab = full.load_s24(); full.step(300)
    t, q, v, w = full.get_state(); assert (full.get_stats()[:, 3] & 2).sum() == 0
    os.environ["MJH_PATCH_POOL_FLOATS"] = "600"
    ms.capi.load().mjh_set_window_solver(0)
    try:
        e = ms.Engine(m, 64)
    finally:
        del os.environ["MJH_PATCH_POOL_FLOATS"]
        ms.capi.load().mjh_set_window_solver(1)
    assert e.window_solver() == 0
    for k in EP:
        e.set_env_param(k, tab[k])
    e.set_initial_qpos(tab["qpos"]); e.set_state(qpos=q, qvel=v, warmstart=w)
    e.step(1)
    st = e.get_stats()
    assert (st[:, 3] & 2).sum() > 32, "most settled piles need more than 600 floats"
    q1 = e.get_state()[1]
    assert np.isfinite(q1).all()
    e.step(100)                                    # boxes sink where their contacts were dropped, nothing worse
    assert np.isfinite(e.get_state()[1]).all()
    e.close(); full.close()


@pytest.mark.gpu
def test_free_bodies_with_gravity_compensation_and_cartesian_forces_match_oracle(lib):
    """free-body models form bias forces, gravity compensation (~disable_gravity: mj_sim.cpp:301-310) and xfrc_applied in closed form
    (no spatial inertias / motion axes): a compensated box hovers, a half-compensated one falls slowly onto the floor, a third one is
    pushed and twisted through xfrc_applied; device vs oracle"""
    from helpers import D, set_opt
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.004)
    lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, -1, -1, -1, -1)
    for name, pos, gc in [(b"hover", (0.0, 0.0, 0.5), 1.0), (b"slow", (0.5, 0.0, 0.3), 0.5), (b"pushed", (-0.5, 0.0, 0.2), 0.0)]:
        bd = lib.mjh_builder_add_body(b, name, 0, D(*pos), D(0.9, 0.1, 0.3, 0.2), gc)
        lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
        lib.mjh_builder_add_geom(b, None, bd, 6, D(0.08, 0.06, 0.05), None, None, None, -1, -1, -1, -1)
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    assert m.nv == 18
    m.c.maxcon = 32; m.c.maxefc = 32 * 6
    e = ms.Engine(m, 2); assert e.solver_order() == 2 and e.patch_sweep() == 1
    d = orc.OrcData(m.ptr); d.call("reset")
    v0 = np.zeros(m.nv); v0[3:6] = [0.5, -0.3, 0.2]; v0[6 + 3: 6 + 6] = [0.1, 0.4, -0.2]
    e.set_state(qvel=np.tile(v0, (2, 1))); d.f("qvel")[:] = v0
    xf = np.zeros((2, 6 * m.nbody)); xf[:, 6 * 3: 6 * 3 + 6] = [1.5, 0.0, 0.4, 0.02, -0.03, 0.05]      # force + torque on "pushed"
    e.set_xfrc_applied(xf); d.f("xfrc_applied")[:] = xf[0]
    done = 0
    for n, tol in ((1, 1e-5), (50, 1e-3), (150, 2e-2)):
        e.step(n - done); d.step(n - done); done = n
        q = e.get_state()[1]
        np.testing.assert_allclose(q[0], d.f("qpos"), atol=tol, err_msg=f"step {n}")
        np.testing.assert_array_equal(q[0], q[1])
    assert abs(d.f("qpos")[2] - 0.5) < 1e-6          # the compensated box has not moved vertically
    assert d.i("ncon") >= 1
    e.close()
