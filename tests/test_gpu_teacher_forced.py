"""Teacher-forced GPU parity in the regime the bench TIMES: the oracle settles the scene (S24: 400 steps, C2: 200 — the settle
phases of bench.py), then for >= 100 consecutive steps the device state (qpos, qvel, warm start, time) is SET from the oracle, both
step once through the reference loop body (mj_step1 + mj_step2, src/mj_main.cpp:83,108), and qpos / qvel are compared for EVERY
environment whose contact set agrees (same ncon and nefc in that step); the fraction that agrees is asserted as well.  No env is
excused: the tolerance is the one the measured distribution supports (printed, recorded in BASELINE.md §3).

Round 4: the DEFAULT engine runs mj_solPGS's own constraint-row order (`mjh_solver_order() == 2`), blocks without a common body side
by side under a precedence-preserving list schedule.  Arms per scene: (1) default engine against the oracle in row order — the
headline; (2) the same with mj_inverse; (3) the default engine against an engine that walks the same order strictly one patch / block
after the other (`mjh_set_pgs_row_order(2)`): BIT-IDENTICAL state, sweep counts included (`np.array_equal`), teacher-forced and
free-running; (4) the legacy reordering schedules (`mjh_set_pgs_row_order(0)`: patch / group first fit) against the oracle in the same
legacy order, and against the row-order oracle (what the old default cost in exactness: the order effect at the sweep cap).
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


def _same_contacts(dc, oc):
    """the narrow phase handed the solver the same problem: same geom pairs in the same order, normals within 1e-4, points within
    1e-5 (fp32 against fp64 from the same state: normally 1e-7; an edge-edge contact between NEARLY PARALLEL box edges — the normal
    is the normalised cross product of two fp32 unit vectors, sin(angle) down to 1e-3 — is conditioned 1e3 times worse, and the one
    such env-step in 3840 moved qvel by 3.7e-4 with both solvers converging to their own fixed point: tools/tf_replay.py)"""
    if len(oc) != len(dc["dist"]):
        return False
    if len(oc) == 0:
        return True
    og = np.array([c["geom"] for c in oc]); of = np.array([c["frame"][:3] for c in oc]); op = np.array([c["pos"] for c in oc])
    return bool((og == dc["geom"]).all() and np.abs(of - dc["frame"][:, :3]).max() <= 1e-4 and np.abs(op - dc["pos"]).max() <= 1e-5)


def teacher_forced(e, ds, nsteps, with_inverse=False):
    """-> dict of [nsteps, nenv] arrays: rel. error of qpos, qvel, qacc after ONE step from the oracle's state; agree; ncon; nefc.
    agree: same ncon / nefc, no capacity flag, AND the same contact records (device snapshot at the state just set, before the step,
    against the oracle's contacts of that step: _same_contacts)"""
    n = len(ds)
    out = {k: np.zeros((nsteps, n)) for k in ("eq", "ev", "ea", "agree", "ncon", "nefc", "iter", "diter", "samecon")}
    for k in range(nsteps):
        e.set_state(qpos=np.array([d.f("qpos") for d in ds]), qvel=np.array([d.f("qvel") for d in ds]),
                    time=np.array([d.f("time")[0] for d in ds]), warmstart=np.array([d.f("qacc_warmstart") for d in ds]))
        dcs = [e.get_contacts(i) for i in range(n)]
        e.step(1, with_inverse)
        for d in ds:
            d.step(1, int(with_inverse))
        _, q, v, w = e.get_state(); st = e.get_stats()
        qo = np.array([d.f("qpos") for d in ds]); vo = np.array([d.f("qvel") for d in ds]); ao = np.array([d.f("qacc") for d in ds])
        out["eq"][k] = _rel(q, qo); out["ev"][k] = _rel(v, vo); out["ea"][k] = _rel(w, ao)
        on = np.array([d.i("ncon") for d in ds]); oe = np.array([d.i("nefc") for d in ds])
        same = np.array([_same_contacts(dcs[i], ds[i].contacts()) for i in range(n)])
        out["samecon"][k] = same
        out["agree"][k] = (st[:, 0] == on) & (st[:, 1] == oe) & (st[:, 3] & 7 == 0) & same
        out["ncon"][k] = on; out["nefc"][k] = oe; out["iter"][k] = [d.i("solver_iter") for d in ds]; out["diter"][k] = st[:, 2]
    return out


def summarize(tag, r):
    a = r["agree"].astype(bool)
    qs = [0.5, 0.9, 0.99, 1.0]
    s = {"tag": tag, "env_steps": int(a.size), "agree_fraction": float(a.mean()), "same_contact_records_fraction": float(r["samecon"].mean()), "mean_ncon": float(r["ncon"].mean()), "mean_nefc": float(r["nefc"].mean()),
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
    """32 S24 envs settled 400 steps by the oracle (row order); every arm starts from copies of these states"""
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


def _engine_in_order(mode, make):
    """an engine created under mjh_set_pgs_row_order(mode): 0 legacy reordering schedules, 1 row order list-scheduled (default),
    2 row order strictly sequential"""
    from mujoco_sim_amd import capi
    lib = capi.load()
    lib.mjh_set_pgs_row_order(mode)
    try:
        e = make()
    finally:
        lib.mjh_set_pgs_row_order(1)
    assert e.pgs_schedule() == mode
    return e


class _oracle_order:
    def __init__(self, mode): self.mode = mode
    def __enter__(self): orc.lib().orc_set_pgs_row_order(self.mode)
    def __exit__(self, *a): orc.lib().orc_set_pgs_row_order(1)


# tolerances: set from the distributions measured on the MI355X; one step from identical states
# measured (r03a, 3840 env-steps, both sides in one order): qpos max 1.6e-7, qvel 99 % 3.8e-6 / max 1.04e-5, qacc max 8.4e-4; 99.9 % agree
S24_TOL_Q, S24_TOL_V = 1e-6, 2e-5
# measured (r03a, 3200 env-steps): qpos max 1.8e-5, qvel median 2.6e-5 / 99 % 5.3e-3 / max 9.6e-3 — the Gauss-Seidel ORDER effect at the cap
S24_ROW_TOL_Q, S24_ROW_TOL_V = 5e-5, 2e-2


def test_s24_teacher_forced_over_the_timed_regime(s24_settled):
    """the default engine IS in mj_solPGS's row order: every agreeing env-step within the same-order tolerance against the oracle's
    plain row-order sweep, and nearly all of them agree"""
    m, e, tab, ds, state = s24_settled
    _restore(ds, state)
    assert e.solver_order() == 2 and e.pgs_schedule() == 1        # row order, list-scheduled patches: what bench.py's S24 line runs
    r = teacher_forced(e, ds, 120)
    s = summarize("s24/default=mj_solPGS-row-order", r)
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
    s = summarize("s24/default+inverse", r)
    a = r["agree"].astype(bool)
    assert s["agree_fraction"] >= 0.97 and r["eq"][a].max() <= S24_TOL_Q and r["ev"][a].max() <= S24_TOL_V, s


def _bitwise_pair(ea, eb, ds, nforced, nfree):
    """both engines teacher-forced from the oracle's states for nforced steps, then free-running nfree steps from the last one:
    every array of the state and the solver statistics must be EQUAL"""
    for k in range(nforced):
        for e in (ea, eb):
            e.set_state(qpos=np.array([d.f("qpos") for d in ds]), qvel=np.array([d.f("qvel") for d in ds]),
                        time=np.array([d.f("time")[0] for d in ds]), warmstart=np.array([d.f("qacc_warmstart") for d in ds]))
            e.step(1, False)
        for d in ds:
            d.step(1, 0)
        sa, sb = ea.get_state(), eb.get_state()
        for x, y in zip(sa, sb):
            assert np.array_equal(x, y), f"teacher-forced step {k}: max diff {np.abs(np.asarray(x) - np.asarray(y)).max()}"
        assert np.array_equal(ea.get_stats(), eb.get_stats()), f"teacher-forced step {k}: ncon / nefc / sweeps / flags differ"
    sweeps = []
    for k in range(nfree):
        ea.step(1, False); eb.step(1, False)
        sa, sb = ea.get_state(), eb.get_state()
        for x, y in zip(sa, sb):
            assert np.array_equal(x, y), f"free-running step {k}"
        st = ea.get_stats()
        assert np.array_equal(st, eb.get_stats())
        sweeps.append(st[:, 2].copy())
    return np.array(sweeps)


def test_s24_list_schedule_is_bit_identical_to_the_sequential_row_order_sweep(s24_settled):
    """patches without a common body commute exactly, and the list schedule never swaps two patches that share one: the default
    engine (up to four patches side by side) and an engine walking the same constraint order one patch per step produce the same
    bits — state, warm start, sweep counts — over 60 teacher-forced and 150 free-running steps of 32 settled piles"""
    m, e, tab, ds, state = s24_settled
    _restore(ds, state)

    def make():
        x = ms.Engine(m, len(ds)); x.load_s24(); return x
    eseq = _engine_in_order(2, make)
    try:
        sw = _bitwise_pair(e, eseq, ds, 60, 150)
    finally:
        eseq.close()
    print(f"S24 list schedule == sequential row order, bitwise; sweeps per step: mean {sw.mean():.1f}, at the cap {np.mean(sw >= 100):.2f}, below {np.mean(sw < 100):.2f}")
    assert (sw < 100).any() and (sw >= 100).any(), "the window must cover both early-stopped and capped solves (the convergence test is part of the claim)"


def test_s24_legacy_patch_order_matches_the_oracle_in_that_order_and_the_order_effect_is_measured(s24_settled):
    """mjh_set_pgs_row_order(0): the round-3 default (contacts regrouped by body pair, first-fit steps) is still an engine option and
    still agrees with the oracle walking the SAME legacy order; against the row-order oracle it shows the Gauss-Seidel order effect at
    the 100-sweep cap (qvel up to 1e-2 per step) — the gap the default engine no longer has"""
    m, e0, tab, ds, state = s24_settled

    def make():
        x = ms.Engine(m, len(ds)); x.load_s24(); return x
    e = _engine_in_order(0, make)
    assert e.solver_order() == 1
    try:
        _restore(ds, state)
        with _oracle_order(0):
            r = teacher_forced(e, ds, 60)
        s = summarize("s24/legacy-patch-order-both", r)
        a = r["agree"].astype(bool)
        assert s["agree_fraction"] >= 0.97 and r["eq"][a].max() <= S24_TOL_Q and r["ev"][a].max() <= S24_TOL_V, s
        _restore(ds, state)
        r = teacher_forced(e, ds, 60)
        s = summarize("s24/legacy-patch-order-vs-row-order-oracle", r)
        a = r["agree"].astype(bool)
        assert s["agree_fraction"] >= 0.97, s
        assert r["eq"][a].max() <= S24_ROW_TOL_Q and r["ev"][a].max() <= S24_ROW_TOL_V, s
        assert r["ev"][a].max() > S24_TOL_V, "the legacy order must differ measurably from the row order at the cap (else this arm tests nothing)"
    finally:
        e.close()


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
    """default engine (row order, up to 16 independent blocks per wave-step) against the oracle's plain row-order sweep"""
    m, e, tab, ds, state = c2_settled
    _restore(ds, state)
    assert e.solver_order() == 2 and e.pgs_schedule() == 1
    r = teacher_forced(e, ds, 100)
    s = summarize("c2/default=mj_solPGS-row-order", r)
    a = r["agree"].astype(bool)
    assert r["ncon"].mean() >= 100
    assert s["agree_fraction"] >= 0.9, s
    # measured (r03, both sides in row order, 160 env-steps): qpos 99 % 1.07e-7, qvel 99 % 1.5e-6 and ONE env-step at 3.1e-6 / 2.1e-4:
    # a box pair whose six-point manifold appears in that step (tools/c2_row_order_probe.py: same counts, qacc of that one body
    # differs); with every block visited in contact order, the order of the points inside a manifold is part of the iterate
    assert np.quantile(r["eq"][a], 0.99) <= C2_TOL_Q and np.quantile(r["ev"][a], 0.99) <= C2_TOL_V, s
    assert r["eq"][a].max() <= C2_ROW_TOL_Q and r["ev"][a].max() <= C2_ROW_TOL_V, s
    if (~a).any():
        assert r["eq"][~a].max() <= 1e-3, s


def test_c2_list_schedule_is_bit_identical_to_the_sequential_row_order_sweep(c2_settled):
    m, e, tab, ds, state = c2_settled
    _restore(ds, state)

    def make():
        x = ms.Engine(m, len(ds)); x.load_tables(tab); return x
    eseq = _engine_in_order(2, make)
    try:
        sw = _bitwise_pair(e, eseq, ds, 12, 30)
    finally:
        eseq.close()
    print(f"C2 list schedule == sequential row order, bitwise; sweeps per step: mean {sw.mean():.1f}")


def test_c2_legacy_group_order_matches_the_oracle_in_that_order(c2_settled):
    m, e0, tab, ds, state = c2_settled

    def make():
        x = ms.Engine(m, len(ds)); x.load_tables(tab); return x
    e = _engine_in_order(0, make)
    assert e.solver_order() == 0
    try:
        _restore(ds, state)
        with _oracle_order(0):
            r = teacher_forced(e, ds, 40)
        s = summarize("c2/legacy-group-order-both", r)
        a = r["agree"].astype(bool)
        assert s["agree_fraction"] >= 0.9 and r["eq"][a].max() <= C2_TOL_Q and r["ev"][a].max() <= C2_TOL_V, s
        _restore(ds, state)
        r = teacher_forced(e, ds, 20)
        s = summarize("c2/legacy-group-order-vs-row-order-oracle", r)
        a = r["agree"].astype(bool)
        assert s["agree_fraction"] >= 0.9 and r["eq"][a].max() <= C2_ROW_TOL_Q and r["ev"][a].max() <= C2_ROW_TOL_V, s
    finally:
        e.close()


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
    # ... and the envs that DO touch the bowl (mesh-convex contacts, VERDICT r04 next #6d): up to 12 of them teacher-forced one step from
    # the device's state — keep stepping until enough of the batch is in contact (more pendulums reach the bowl as time goes on)
    for _ in range(6):
        if (st[:, 0] > 0).sum() >= 12:
            break
        e.step(50, True); st = e.get_stats()
    t, q, v, w = e.get_state(); st = e.get_stats()
    touching = np.nonzero(st[:, 0] > 0)[0][:12]
    assert len(touching) >= 8, f"only {len(touching)} of {nenv} envs touch the bowl"
    e.step(1, True)
    _, q1, v1, w1 = e.get_state(); st1 = e.get_stats()
    tq = tv = 0.0; same = 0
    for i in touching:
        d = orc.OrcData(m.ptr); d.ifield("controlled")[:] = z["controlled"]
        d.f("qpos")[:] = q[i]; d.f("qvel")[:] = v[i]; d.f("qacc_warmstart")[:] = w[i]; d.f("qacc")[:] = w[i]; d.f("time")[0] = t[i]
        d.step(1, 1)
        if d.i("ncon") != st1[i, 0] or d.i("nefc") != st1[i, 1]:
            continue
        same += 1
        tq = max(tq, float(np.abs(q1[i] - d.f("qpos")).max())); tv = max(tv, float(np.abs(v1[i] - d.f("qvel")).max() / max(1.0, np.abs(d.f("qvel")).max())))
    print(f"C5 touching envs, one teacher-forced step: {same} of {len(touching)} with the oracle's contact / row counts, qpos {tq:.2e} qvel {tv:.2e}")
    assert same >= 6 and tq <= 2e-6 and tv <= 5e-5
    # the envs differ (per-env spin) and a few of them touch the bowl
    assert np.unique(np.round(q[:, 0], 6)).size > nenv // 2
    print(f"C5 4096 envs: {int((st[:, 0] > 0).sum())} envs in contact, max ncon {st[:, 0].max()}, energy {Es[0].mean():.4f} -> {Es[-1].mean():.4f}")
    e.close()
