"""GPU tests of round 3's solver forms against the oracle (through the C ABI): the dense row-space solver of articulated models
(csrc/dense_pgs.h: AR = J M^-1 J^T on the matrix cores, Gauss-Seidel as column updates) incl. its per-env fall-back to the block
solver, and the sixteen-blocks-per-wave-step form of free-body piles incl. single-row (connect) blocks."""
import os

import numpy as np
import pytest

import mujoco_sim_amd as ms
import orc
from conftest import ROOT
from helpers import D

pytestmark = pytest.mark.gpu


def _robot(name):
    from mujoco_sim_amd.tables import load_model_tables
    return load_model_tables(os.path.join(ROOT, "tests", "golden", f"robot_{name}.npz"))


def _rollout_against_oracle(m, z, nsteps, nenv=3, active=None):
    from test_robot_fixtures import robot_command
    e = ms.Engine(m, nenv)
    e.set_controlled_dofs(z["controlled"].astype(np.int32))
    d = orc.OrcData(m.ptr); d.ifield("controlled")[:] = z["controlled"]
    d.f("qvel")[:] = z["qvel0"]; e.set_state(qvel=np.tile(z["qvel0"], (nenv, 1)))
    worst_q = worst_v = 0.0
    for k in range(1, nsteps + 1):
        cmd = robot_command(m, k)
        e.set_cmd(ddq=np.tile(cmd, (nenv, 1))); e.step(1, True)
        d.f("ddq")[:] = cmd; d.step(1, 1)
        if k % 10 == 0:
            _, q, v, _ = e.get_state()
            assert np.array_equal(q[0], q[-1])
            worst_q = max(worst_q, np.abs(q[0] - d.f("qpos")).max()); worst_v = max(worst_v, np.abs(v[0] - d.f("qvel")).max())
            e.set_state(qpos=np.tile(d.f("qpos"), (nenv, 1)), qvel=np.tile(d.f("qvel"), (nenv, 1)), warmstart=np.tile(d.f("qacc_warmstart"), (nenv, 1)))
    st = e.get_stats(); dense = e.dense_solver()
    e.close()
    return worst_q, worst_v, st, d, dense


def test_dense_solver_runs_for_the_robots_and_can_be_switched_off(lib):
    """the same PR2-on-the-floor rollout with the dense solver (default) and with MJH_DENSE=0 (block solver): both within the robot
    tolerance of the oracle, same row counts; mjh_dense_solver() tells which one an engine runs"""
    m, z = _robot("pr2_world")
    lib.mjh_set_layout_policy(2)
    try:
        qa, va, sta, da, dense_a = _rollout_against_oracle(m, z, 60)
        os.environ["MJH_DENSE"] = "0"
        try:
            qb, vb, stb, db, dense_b = _rollout_against_oracle(m, z, 60)
        finally:
            del os.environ["MJH_DENSE"]
    finally:
        lib.mjh_set_layout_policy(0)
    assert dense_a == 1 and dense_b == 0, "articulated model in the many-body layout: the dense solver is the default, MJH_DENSE=0 turns it off"
    assert sta[0, 1] == da.i("nefc") and stb[0, 1] == db.i("nefc") and da.i("nefc") > 20
    assert qa < 4e-4 and qb < 4e-4 and va < 4e-3 and vb < 4e-3, (qa, qb, va, vb)
    print(f"PR2 on the floor, 60 steps re-synchronised every 10: dense {qa:.2e} / {va:.2e}, block solver {qb:.2e} / {vb:.2e} (qpos / qvel)")


def test_envs_beyond_the_dense_capacity_keep_the_block_solver(lib):
    """MJH_DENSE_CAP=64 on the C4 fixture (PR2 + world + objects: ~100-230 rows): every env exceeds the capacity, the build launch
    leaves them to the block solver (meta[7] = 0) inside the same chain of launches — results as before"""
    m, z = _robot("c4_pr2_world_objects_mesh")
    os.environ["MJH_DENSE_CAP"] = "64"
    try:
        q, v, st, d, dense = _rollout_against_oracle(m, z, 40)
    finally:
        del os.environ["MJH_DENSE_CAP"]
    assert dense == 1 and d.i("nefc") > 64 and st[0, 1] == d.i("nefc")
    assert q < 4e-4 and v < 4e-3, (q, v)


def _pile_with_connects(lib, nbox=20, nconnect=4):
    """free boxes resting / landing on the floor in a loose grid, a few of them tied to a neighbour by <connect> equalities:
    free-body model (diagonal M), > 64 blocks, groups of up to 16, single-row blocks in front of the contact blocks"""
    b = lib.mjh_builder_create()
    lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, D(2, 0.05, 0.01), 4, -1, -1, -1)
    ids = []
    rng = np.random.default_rng(12)
    for k in range(nbox):
        x, y = 0.32 * (k % 5), 0.32 * (k // 5)
        bd = lib.mjh_builder_add_body(b, b"box%d" % k, 0, D(x, y, 0.11 + 0.004 * k), None, 0.0)
        lib.mjh_builder_add_joint(b, b"free%d" % k, bd, 0, None, None, None, 0, 0, 0, 0, 0)
        sz = rng.uniform(0.07, 0.1, 3)
        lib.mjh_builder_add_geom(b, b"g%d" % k, bd, 6, D(*sz), None, None, None, -1, -1, -1, -1)
        ids.append(bd)
    for k in range(nconnect):
        lib.mjh_builder_add_eq_connect(b, ids[2 * k], ids[2 * k + 1], D(0.16, 0.0, 0.02))
    lib.mjh_builder_set_capacity(b, 6 * nbox, 6 * nbox * 4 + 3 * nconnect)
    m = ms.Model(lib.mjh_builder_compile(b), lib); lib.mjh_builder_destroy(b)
    return m


def test_quad_sweep_with_single_row_blocks_matches_the_oracle(lib):
    m = _pile_with_connects(lib)
    assert m.nv == 120 and m.neq == 4
    nenv = 3
    e = ms.Engine(m, nenv)
    assert e.solver_order() == 2 and e.patch_sweep() == 0 and e.lds_bytes <= 160 * 1024
    d = orc.OrcData(m.ptr)
    rng = np.random.default_rng(5)
    v0 = rng.normal(size=m.nv) * 0.2
    e.set_state(qvel=np.tile(v0, (nenv, 1))); d.f("qvel")[:] = v0
    saw_many = False
    for k in range(1, 141):                                          # (dt 0.002: the boxes land after ~40 steps)
        e.step(1); d.step(1)
        if k % 10 == 0:
            _, q, v, _ = e.get_state(); st = e.get_stats()
            assert np.array_equal(q[0], q[-1]) and (st[:, 3] == 0).all() and d.i("warn") == 0
            assert st[0, 1] == d.i("nefc"), (k, st[0], d.i("nefc"))
            saw_many |= d.i("ncon") + 12 > 64                       # (> 64 blocks: the grouped order and the quad sweep)
            np.testing.assert_allclose(q[0], d.f("qpos"), atol=3e-4, err_msg=f"step {k}")
            np.testing.assert_allclose(v[0], d.f("qvel"), atol=3e-3, err_msg=f"step {k}")
            e.set_state(qpos=np.tile(d.f("qpos"), (nenv, 1)), qvel=np.tile(d.f("qvel"), (nenv, 1)), warmstart=np.tile(d.f("qacc_warmstart"), (nenv, 1)))
    assert saw_many, "the scene must reach the many-block regime"
    e.close()


# ---------------------------------------------------------------- BASELINE size, compared with the oracle on sampled envs
def _one_step_on_samples(e, make_oracle, sample, with_inverse=False):
    """the engine holds its OWN settled state at full size; the sampled envs' state goes to the oracle, both step once, compare"""
    _, q, v, w = e.get_state()
    t = e.get_state()[0]
    ds = []
    for i in sample:
        d = make_oracle(i)
        d.f("qpos")[:] = q[i]; d.f("qvel")[:] = v[i]; d.f("qacc_warmstart")[:] = w[i]; d.f("qacc")[:] = w[i]; d.f("time")[0] = t[i]
        ds.append(d)
    e.step(1, with_inverse)
    for d in ds:
        d.step(1, int(with_inverse))
    _, q1, v1, _ = e.get_state(); st = e.get_stats()
    eq, ev, agree = [], [], []
    for k, i in enumerate(sample):
        d = ds[k]
        agree.append(st[i, 0] == d.i("ncon") and st[i, 1] == d.i("nefc"))
        eq.append(np.abs(q1[i] - d.f("qpos")).max() / max(1.0, np.abs(d.f("qpos")).max()))
        ev.append(np.abs(v1[i] - d.f("qvel")).max() / max(1.0, np.abs(d.f("qvel")).max()))
    return np.array(eq), np.array(ev), np.array(agree), ds


def test_s24_at_4096_envs_one_step_parity_on_sampled_envs():
    """BASELINE size: 4096 S24 envs settle 400 steps ON THE DEVICE (what bench.py times), then 64 envs spread over the batch are
    handed to the oracle and both advance one step, five times over (state re-read from the device each time)"""
    from helpers import oracle_s24
    m = ms.scene("s24")
    nenv = 4096
    e = ms.Engine(m, nenv)
    tab = e.load_s24()
    e.step(400)
    sample = list(range(0, nenv, 64))
    worst_q = worst_v = 0.0; agreeing = total = 0
    for rep in range(5):
        eq, ev, ag, _ = _one_step_on_samples(e, lambda i: oracle_s24(m, tab, i), sample)
        worst_q = max(worst_q, eq[ag].max()); worst_v = max(worst_v, ev[ag].max()); agreeing += int(ag.sum()); total += len(ag)
        e.step(7)
    print(f"S24 4096 envs, {total} sampled env-steps: contact sets agree {agreeing / total:.3f}, qpos {worst_q:.2e}, qvel {worst_v:.2e}")
    assert agreeing / total >= 0.97 and worst_q <= 1e-6 and worst_v <= 2e-5
    e.close()


def test_c2_at_4096_envs_one_step_parity_on_sampled_envs():
    from mujoco_sim_amd.engine import EP
    m = ms.scene("boxpile", 64); m.c.maxcon = 600; m.c.maxefc = 2400
    nenv = 4096
    e = ms.Engine(m, nenv)
    tab = e.load_tables(ms.boxes_randomize(m, 0, nenv, jitter=0.01))
    e.step(200)

    def make(i):
        d = orc.OrcData(m.ptr)
        for k, wh in EP.items():
            d.set_env_param(wh, tab[k][i])
        return d
    sample = [5 + 273 * k for k in range(15)] + [4090]        # 16 envs spread over the batch (VERDICT r03 #6b: was 4)
    worst_q = worst_v = 0.0; agreeing = total = 0
    for rep in range(2):
        eq, ev, ag, ds = _one_step_on_samples(e, make, sample)
        if ag.any():
            worst_q = max(worst_q, eq[ag].max()); worst_v = max(worst_v, ev[ag].max())
        agreeing += int(ag.sum()); total += len(ag)
        e.step(5)
    print(f"C2 4096 envs, {total} sampled env-steps at ~{ds[0].i('ncon')} contacts: contact sets agree {agreeing / total:.3f}, qpos {worst_q:.2e}, qvel {worst_v:.2e}")
    assert agreeing / total >= 0.9 and worst_q <= 1e-6 and worst_v <= 1e-5
    e.close()


def test_c4_at_2048_envs_one_step_parity_on_sampled_envs():
    """PR2 + world + (empty) object pool at BASELINE size, mj_inverse every step, commands as the fixtures use them: the dense
    solver's path under the oracle at full size"""
    from test_robot_fixtures import robot_command
    m, z = _robot("c4_pr2_world_objects_mesh")
    nenv = 2048
    e = ms.Engine(m, nenv)
    assert e.dense_solver() == 1
    e.set_controlled_dofs(z["controlled"].astype(np.int32))
    lib = m.lib
    names = [lib.mjh_id2name(m.ptr, 0, b).decode() for b in range(m.c.nbody)]
    slots = [b for b, n in enumerate(names) if n.startswith("object_")]
    for b in slots:
        e.set_slot_active(b, False)
    sbase = m.c.nbody - 32 if m.c.nbody > 32 else 0
    mask = 0
    for b in slots:
        mask |= 1 << (b - sbase)
    rng = np.random.default_rng(4)
    for k in range(1, 121):
        if k % 10 == 1:
            e.set_cmd(ddq=np.tile(robot_command(m, k), (nenv, 1)) * rng.uniform(0.5, 1.5, size=(nenv, 1)))
        e.step(1, True)

    def make(i):
        d = orc.OrcData(m.ptr); d.ifield("controlled")[:] = z["controlled"]
        orc.lib().orc_set_slot_mask(d.d, mask)
        return d
    sample = [66 * k for k in range(31)] + [2047]            # 32 envs spread over the batch (VERDICT r03 #6b: was 4, three of which had to agree)
    eq, ev, ag, ds = _one_step_on_samples(e, make, sample, with_inverse=True)
    print(f"C4 2048 envs, {len(sample)} sampled: nefc {min(d.i('nefc') for d in ds)}..{max(d.i('nefc') for d in ds)}, contact sets agree {ag.mean():.3f}, qpos {eq[ag].max():.2e}, qvel {ev[ag].max():.2e}; "
          f"not agreeing: qpos {eq[~ag].max() if (~ag).any() else 0:.2e}")
    assert ag.mean() >= 0.9 and eq[ag].max() <= 2e-6 and ev[ag].max() <= 1e-4
    assert eq.max() < 1e-3                                     # (an env whose contact set differs by a point is still the same robot pose)
    e.close()


def test_literal_loop_on_cohort_streams_equals_the_single_stream_loop():
    """The reference's loop body (mj_main.cpp:82-112: mj_step1 -> read() -> update -> write() -> mj_step2) through the split entry
    points.  With cohorts the launches go out per cohort stream and a read / write of ONE environment only waits for that
    environment's cohort (engine.hip: launch_lpt, range_stream) — the other cohorts keep running.  Commands to envs of different
    cohorts, a range that spans two cohorts (joins them) and full-range calls in between: bitwise the same trajectories as the
    same calls on one stream."""
    m = ms.scene("arm7", 1)
    nenv = 1536                                   # 512 envs per cohort
    rng = np.random.default_rng(5)
    q0 = 0.3 * rng.normal(size=(nenv, m.nq))
    out = []
    for cohorts in (1, 3):
        e = ms.Engine(m, nenv); e.set_cohorts(cohorts)
        e.set_controlled_dofs(np.ones(m.nv, dtype=np.int32))
        e.set_state(qpos=q0)
        r2 = np.random.default_rng(6)
        reads = []
        for k in range(40):
            e.step1(); e.inverse()
            for env in (0, 700, 1535):                                    # one env of each cohort
                q, v, f = e.get_joint_state(env, 1)
                reads.append(np.concatenate([q.ravel(), v.ravel(), f.ravel()]))
                e.set_cmd(ddq=(2.0 * r2.normal(size=(1, m.nv)) - 5.0 * v), dq=None, env0=env)
            if k % 7 == 3:
                e.set_cmd(ddq=r2.normal(size=(40, m.nv)), dq=None, env0=500)   # envs 500 .. 539: two cohorts
            if k % 11 == 5:
                reads.append(e.get_field("qfrc_inverse")[::97].ravel())      # a full-range getter in the middle of the loop
            e.step2()
            if k % 13 == 6:
                e.step(2, True)                                               # the fused step mixed in
        t, q, v, w = e.get_state()
        out.append((q.copy(), v.copy(), np.concatenate(reads)))
        e.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert np.abs(out[0][1]).max() > 1e-2
