"""Teacher-forced GPU parity in the regime the bench TIMES (round 3): the oracle settles the scene (S24: 400 steps, C2: 200 — the
settle phases of bench.py), then for >= 100 consecutive steps the device state (qpos, qvel, warm start, time) is SET from the
oracle, both step once through the reference loop body (mj_step1 + mj_step2, src/mj_main.cpp:83,108), and qpos / qvel are
compared for EVERY environment whose contact set agrees (same ncon and nefc in that step); the fraction that agrees is asserted
as well.  No env is excused: the tolerance is the one the measured distribution supports (printed, recorded in BASELINE.md §3).

Three arms per scene: the oracle in the DEVICE's Gauss-Seidel order (patch / group order: what the kernels implement by default);
the oracle in plain constraint-row order (`orc_set_pgs_row_order(1)` = mj_solPGS's order) against the device in ITS order — the
device-vs-MuJoCo-order gap as a GPU-side number, per round, instead of a CPU-only study; and BOTH in mj_solPGS's row order
(`mjh_set_pgs_row_order(1)`: the order is a choice of the engine, a user who needs the reference's iterates can have them).
"""
import json
import os

import numpy as np
import pytest

import mujoco_sim_amd as ms
import orc
from conftest import ROOT
from helpers import oracle_s24
from mujoco_sim_amd.engine import EP

pytestmark = pytest.mark.gpu
OUT = os.path.join(ROOT, "gpurun_out")


def _rel(a, b):
    """per env: max |a - b| / max(1, max |b|)   ("relative on qpos / qvel")"""
    return np.abs(a - b).max(axis=1) / np.maximum(1.0, np.abs(b).max(axis=1))


def teacher_forced(e, ds, nsteps, with_inverse=False):
    """-> dict of [nsteps, nenv] arrays: rel. error of qpos, qvel, qacc after ONE step from the oracle's state; agree; ncon; nefc"""
    n = len(ds)
    out = {k: np.zeros((nsteps, n)) for k in ("eq", "ev", "ea", "agree", "ncon", "nefc", "iter")}
    for k in range(nsteps):
        e.set_state(qpos=np.array([d.f("qpos") for d in ds]), qvel=np.array([d.f("qvel") for d in ds]),
                    time=np.array([d.f("time")[0] for d in ds]), warmstart=np.array([d.f("qacc_warmstart") for d in ds]))
        e.step(1, with_inverse)
        for d in ds:
            d.step(1, int(with_inverse))
        _, q, v, w = e.get_state(); st = e.get_stats()
        qo = np.array([d.f("qpos") for d in ds]); vo = np.array([d.f("qvel") for d in ds]); ao = np.array([d.f("qacc") for d in ds])
        out["eq"][k] = _rel(q, qo); out["ev"][k] = _rel(v, vo); out["ea"][k] = _rel(w, ao)
        on = np.array([d.i("ncon") for d in ds]); oe = np.array([d.i("nefc") for d in ds])
        out["agree"][k] = (st[:, 0] == on) & (st[:, 1] == oe) & (st[:, 3] & 7 == 0)
        out["ncon"][k] = on; out["nefc"][k] = oe; out["iter"][k] = [d.i("solver_iter") for d in ds]
    return out


def summarize(tag, r):
    a = r["agree"].astype(bool)
    qs = [0.5, 0.9, 0.99, 1.0]
    s = {"tag": tag, "env_steps": int(a.size), "agree_fraction": float(a.mean()), "mean_ncon": float(r["ncon"].mean()), "mean_nefc": float(r["nefc"].mean()),
         "mean_sweeps": float(r["iter"].mean()),
         "qpos_rel_quantiles_50_90_99_max": [float(x) for x in np.quantile(r["eq"][a], qs)],
         "qvel_rel_quantiles_50_90_99_max": [float(x) for x in np.quantile(r["ev"][a], qs)],
         "qacc_rel_quantiles_50_90_99_max": [float(x) for x in np.quantile(r["ea"][a], qs)],
         "disagreeing_qpos_rel_max": float(r["eq"][~a].max()) if (~a).any() else None}
    print("TEACHER-FORCED", json.dumps(s))
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, "teacher_forced.jsonl"), "a") as f:
            f.write(json.dumps(s) + "\n")
    except OSError:
        pass
    return s


# ---------------------------------------------------------------- S24 (the metric's scene)
@pytest.fixture(scope="module")
def s24_settled():
    """32 S24 envs settled 400 steps by the oracle (device order); every arm starts from copies of these states"""
    m = ms.scene("s24")
    nenv = 32
    e = ms.Engine(m, nenv)
    tab = e.load_s24()
    ds = [oracle_s24(m, tab, i) for i in range(nenv)]
    for d in ds:
        d.step(400)
    state = [(d.f("qpos").copy(), d.f("qvel").copy(), d.f("qacc_warmstart").copy(), d.f("time")[0]) for d in ds]
    yield m, e, tab, ds, state
    e.close()


def _restore(ds, state):
    for d, (q, v, w, t) in zip(ds, state):
        d.f("qpos")[:] = q; d.f("qvel")[:] = v; d.f("qacc_warmstart")[:] = w; d.f("qacc")[:] = w; d.f("time")[0] = t


# tolerances: set from the distributions measured on the MI355X; one step from identical states
# measured (r03a, 3840 env-steps): qpos max 1.6e-7, qvel 99 % 3.8e-6 / max 1.04e-5, qacc max 8.4e-4; 99.9 % of the env-steps agree
S24_TOL_Q, S24_TOL_V = 1e-6, 2e-5
# measured (r03a, 3200 env-steps): qpos max 1.8e-5, qvel median 2.6e-5 / 99 % 5.3e-3 / max 9.6e-3 — the Gauss-Seidel ORDER effect at the cap
S24_ROW_TOL_Q, S24_ROW_TOL_V = 5e-5, 2e-2


def test_s24_teacher_forced_over_the_timed_regime(s24_settled):
    """device order on both sides: every agreeing env-step within tolerance, and nearly all of them agree"""
    m, e, tab, ds, state = s24_settled
    _restore(ds, state)
    assert e.solver_order() == 1                                   # the patch sweep: what bench.py's S24 line runs
    r = teacher_forced(e, ds, 120)
    s = summarize("s24/device-order", r)
    a = r["agree"].astype(bool)
    assert r["ncon"].mean() >= 12, "the window must sit in the settled, contact-rich regime the bench times"
    assert s["agree_fraction"] >= 0.97, s
    assert r["eq"][a].max() <= S24_TOL_Q and r["ev"][a].max() <= S24_TOL_V, s
    # an env whose contact set differs in a step (a contact at |dist| ~ 1e-7 of the margin) is still close after that one step
    if (~a).any():
        assert r["eq"][~a].max() <= 1e-3, s


def test_s24_teacher_forced_with_mj_inverse_every_step(s24_settled):
    m, e, tab, ds, state = s24_settled
    _restore(ds, state)
    r = teacher_forced(e, ds, 100, with_inverse=True)
    s = summarize("s24/device-order+inverse", r)
    a = r["agree"].astype(bool)
    assert s["agree_fraction"] >= 0.97 and r["eq"][a].max() <= S24_TOL_Q and r["ev"][a].max() <= S24_TOL_V, s


def test_s24_teacher_forced_against_mj_solpgs_row_order(s24_settled):
    """the oracle visits the rows in plain constraint order, as mj_solPGS does; the device keeps its patch order.  Both are
    Gauss-Seidel on the same problem stopped at the same 100-sweep cap: the gap is the order effect, measured on the GPU"""
    m, e, tab, ds, state = s24_settled
    _restore(ds, state)
    L = orc.lib()
    L.orc_set_pgs_row_order(1)
    try:
        r = teacher_forced(e, ds, 100)
    finally:
        L.orc_set_pgs_row_order(0)
    s = summarize("s24/mj_solPGS-row-order", r)
    a = r["agree"].astype(bool)
    assert s["agree_fraction"] >= 0.97, s
    assert r["eq"][a].max() <= S24_ROW_TOL_Q and r["ev"][a].max() <= S24_ROW_TOL_V, s


def _row_order_engine(make):
    """an engine created under mjh_set_pgs_row_order(1): Gauss-Seidel in mj_solPGS's own row order on the DEVICE"""
    from mujoco_sim_amd import capi
    lib = capi.load()
    lib.mjh_set_pgs_row_order(1)
    try:
        e = make()
    finally:
        lib.mjh_set_pgs_row_order(0)
    assert e.solver_order() == 2
    return e


def test_s24_device_in_mj_solpgs_row_order_matches_the_oracle_in_row_order(s24_settled):
    """the order is a CHOICE of the engine, not a property of the kernels: with mjh_set_pgs_row_order(1) the device walks the rows
    as mj_solPGS does (one block after the other, nothing side by side), and then agrees with the oracle in that order as closely
    as it does in its own order — same tolerances as the first test"""
    m, e0, tab, ds, state = s24_settled
    _restore(ds, state)

    def make():
        e = ms.Engine(m, len(ds)); e.load_s24(); return e
    e = _row_order_engine(make)
    L = orc.lib()
    L.orc_set_pgs_row_order(1)
    try:
        r = teacher_forced(e, ds, 100)
    finally:
        L.orc_set_pgs_row_order(0)
        e.close()
    s = summarize("s24/both-in-mj_solPGS-row-order", r)
    a = r["agree"].astype(bool)
    assert s["agree_fraction"] >= 0.97, s
    assert r["eq"][a].max() <= S24_TOL_Q and r["ev"][a].max() <= S24_TOL_V, s


# ---------------------------------------------------------------- C2 (64-box pile, D3-exact)
@pytest.fixture(scope="module")
def c2_settled():
    m = ms.scene("boxpile", 64); m.c.maxcon = 600; m.c.maxefc = 2400
    nenv = 4
    e = ms.Engine(m, nenv)
    tab = e.load_tables(ms.boxes_randomize(m, 0, nenv, jitter=0.01))
    ds = []
    for i in range(nenv):
        d = orc.OrcData(m.ptr)
        for k, wh in EP.items():
            d.set_env_param(wh, tab[k][i])
        d.set_qpos(tab["qpos"][i]); d.call("reset")
        d.step(200)                                                 # bench.py's C2 settle phase, in the oracle
        ds.append(d)
    state = [(d.f("qpos").copy(), d.f("qvel").copy(), d.f("qacc_warmstart").copy(), d.f("time")[0]) for d in ds]
    yield m, e, tab, ds, state
    e.close()


# measured (r03a, 400 env-steps at ~200 contacts / 1000 rows): qpos max 1.4e-7, qvel max 2.6e-6, qacc max 1.7e-5; 99 % agree
C2_TOL_Q, C2_TOL_V = 1e-6, 1e-5
# measured (r03a, 160 env-steps): qpos max 1.3e-5, qvel max 1.04e-3
C2_ROW_TOL_Q, C2_ROW_TOL_V = 5e-5, 4e-3


def test_c2_teacher_forced_over_the_timed_regime(c2_settled):
    m, e, tab, ds, state = c2_settled
    _restore(ds, state)
    assert e.solver_order() == 0
    r = teacher_forced(e, ds, 100)
    s = summarize("c2/device-order", r)
    a = r["agree"].astype(bool)
    assert r["ncon"].mean() >= 100
    assert s["agree_fraction"] >= 0.9, s
    assert r["eq"][a].max() <= C2_TOL_Q and r["ev"][a].max() <= C2_TOL_V, s
    if (~a).any():
        assert r["eq"][~a].max() <= 1e-3, s


def test_c2_teacher_forced_against_mj_solpgs_row_order(c2_settled):
    m, e, tab, ds, state = c2_settled
    _restore(ds, state)
    L = orc.lib()
    L.orc_set_pgs_row_order(1)
    try:
        r = teacher_forced(e, ds, 40)
    finally:
        L.orc_set_pgs_row_order(0)
    s = summarize("c2/mj_solPGS-row-order", r)
    a = r["agree"].astype(bool)
    assert s["agree_fraction"] >= 0.9, s
    assert r["eq"][a].max() <= C2_ROW_TOL_Q and r["ev"][a].max() <= C2_ROW_TOL_V, s


def test_c2_device_in_mj_solpgs_row_order_matches_the_oracle_in_row_order(c2_settled):
    m, e0, tab, ds, state = c2_settled
    _restore(ds, state)

    def make():
        e = ms.Engine(m, len(ds)); e.load_tables(tab); return e
    e = _row_order_engine(make)
    L = orc.lib()
    L.orc_set_pgs_row_order(1)
    try:
        r = teacher_forced(e, ds, 40)
    finally:
        L.orc_set_pgs_row_order(0)
        e.close()
    s = summarize("c2/both-in-mj_solPGS-row-order", r)
    a = r["agree"].astype(bool)
    assert s["agree_fraction"] >= 0.9, s
    # measured (r03, 160 env-steps): qpos 99 % 1.07e-7, qvel 99 % 1.5e-6 — the device-order figures — and ONE env-step at 3.1e-6 / 2.1e-4:
    # a box pair whose six-point manifold appears in that step (tools/c2_row_order_probe.py: same counts, qacc of that one body
    # differs); with every block visited in contact order, the order of the points inside a manifold is part of the iterate
    assert np.quantile(r["eq"][a], 0.99) <= C2_TOL_Q and np.quantile(r["ev"][a], 0.99) <= C2_TOL_V, s
    assert r["eq"][a].max() <= C2_ROW_TOL_Q and r["ev"][a].max() <= C2_ROW_TOL_V, s


# ---------------------------------------------------------------- C5 at its per-GPU size
def test_c5_full_size_invariants():
    """C5 (multi_mujoco_sim.launch scene: pendulum.xml world + static bowl.xml, 37 mesh geoms) at 4096 envs — the per-GPU share of
    BASELINE's 32768 over 8 GPUs — with per-env initial spin, mj_inverse every step: finite, no overflow / reset, unit quaternions,
    the three ball joints keep the bodies on their spheres (joint anchors are exact in minimal coordinates: |xpos - anchor| is the
    model's), damping 0.5 and gravity -0.1 dissipate: kinetic + potential energy never increases, and the inverse dynamics of the
    unforced system returns ~0 generalized force (qfrc_inverse = qfrc_applied = 0) wherever nothing touches."""
    from mujoco_sim_amd.tables import load_model_tables
    m, z = load_model_tables(os.path.join(ROOT, "tests", "golden", "robot_c5_pendulum_bowl_mesh.npz"))
    nenv = 4096
    e = ms.Engine(m, nenv)
    e.set_controlled_dofs(z["controlled"].astype(np.int32))
    rng = np.random.default_rng(0xC5)
    v0 = z["qvel0"][None, :] * rng.uniform(0.5, 1.5, size=(nenv, 1))
    e.set_state(qvel=v0)
    e.forward(); E0 = e.get_field("energy").sum(axis=1)
    Es = [E0]
    for _ in range(4):
        e.step(50, True)
        e.forward(); Es.append(e.get_field("energy").sum(axis=1))
    t, q, v, _ = e.get_state(); st = e.get_stats()
    assert np.isfinite(q).all() and np.isfinite(v).all()
    assert (st[:, 3] == 0).all(), f"flags {np.unique(st[:, 3])}"
    np.testing.assert_allclose(t, 200 * m.opt.timestep, rtol=1e-12)
    jt = m.array("jnt_type"); qa = m.array("jnt_qposadr")
    for j in range(m.njnt):
        if jt[j] == 1:                                               # ball joints: unit quaternions
            np.testing.assert_allclose(np.linalg.norm(q[:, qa[j]:qa[j] + 4], axis=1), 1, atol=1e-5)
    Es = np.array(Es)
    drift = (Es[1:] - Es[:-1]) / np.maximum(1e-6, np.abs(Es[:-1]) + 1e-3)
    assert np.quantile(drift, 0.999) < 1e-3, "damped pendulum: energy must not grow"
    assert np.mean(Es[-1] < Es[0]) > 0.99
    # eight envs spread over the batch against the oracle run with THEIR spin (a smooth scene: free-running 200 steps, 1e-4),
    # qfrc_inverse included (mj_hw_interface.cpp:61: computed mid-step from the previous step's qacc, so it is NOT ~0 here)
    fi = e.get_field("qfrc_inverse")
    checked = 0
    for i in range(0, nenv, nenv // 8):
        d = orc.OrcData(m.ptr); d.ifield("controlled")[:] = z["controlled"]
        d.f("qvel")[:] = v0[i]
        d.step(200, 1)
        if d.i("ncon") or st[i, 0]:
            continue                                                 # (touching the bowl: contact scenes are covered by the fixture test)
        np.testing.assert_allclose(q[i], d.f("qpos"), atol=1e-4); np.testing.assert_allclose(v[i], d.f("qvel"), atol=1e-4)
        np.testing.assert_allclose(fi[i], d.f("qfrc_inverse"), atol=1e-4 * max(1.0, np.abs(d.f("qfrc_inverse")).max()))
        checked += 1
    assert checked >= 4
    # the envs differ (per-env spin) and a few of them touch the bowl
    assert np.unique(np.round(q[:, 0], 6)).size > nenv // 2
    print(f"C5 4096 envs: {int((st[:, 0] > 0).sum())} envs in contact, max ncon {st[:, 0].max()}, energy {Es[0].mean():.4f} -> {Es[-1].mean():.4f}")
    e.close()
