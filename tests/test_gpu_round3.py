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
    assert e.solver_order() == 0 and e.lds_bytes <= 160 * 1024
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
