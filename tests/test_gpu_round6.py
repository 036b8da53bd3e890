"""Round 6 GPU tests: the window hand-over's own capacity rule, graph re-capture while steps are in flight, and
the N-rank RCCL publish path executed on ONE device through the recording stand-in of tests/nccl_stub."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import mujoco_sim_amd as ms
import orc
from conftest import ROOT
from helpers import oracle_s24
from test_gpu_round5 import _s24d_seeds, S24D_HEAVY_SEEDS

pytestmark = pytest.mark.gpu


def test_window_hand_over_drops_whole_contacts_beyond_its_capacity(monkeypatch):
    """ADVICE r05 (engine.hip:395 / window_emit): a window model whose rows can exceed the window capacity (16 rows x win_maxw; maxefc of a
    128-contact model is 512 .. 768, the capacity 384) used to have its rows cut at the capacity — in the middle of a friction pyramid, which
    leaves a net tangential force.  Now the hand-over keeps the longest prefix of whole blocks that fits: the oracle's own rule for maxefc
    (a contact whose rows do not fit drops it and every later one).  Checked with the capacity lowered to 6 windows = 96 rows
    (MJH_WINDOW_MAXW; S24D's settled piles carry 100 - 250 rows) against the oracle built with maxefc = 96: one teacher-forced step agrees to
    round-off on every env, the capacity flag is raised exactly where rows were dropped, and the kept rows are a multiple of whole contacts."""
    seeds = S24D_HEAVY_SEEDS[:6] + list(range(26))
    m, e, tab = _s24d_seeds(seeds)
    e.step(450)
    t, q, v, w = e.get_state(); st0 = e.get_stats()
    assert (st0[:, 3] & 3 == 0).all() and (st0[:, 1] > 96).sum() >= 16, "the sample must hold envs beyond 96 rows"
    e.close()
    monkeypatch.setenv("MJH_WINDOW_MAXW", "6")
    m2, e2, _ = _s24d_seeds(seeds)
    monkeypatch.delenv("MJH_WINDOW_MAXW")
    mo = ms.scene("s24pen", 0.175, 96); mo.c.maxefc = 96           # the oracle's row capacity = the hand-over's
    ds = [oracle_s24(mo, tab, i) for i in range(len(seeds))]
    for i, d in enumerate(ds):
        d.set_qpos(q[i]); d.f("qvel")[:] = v[i]; d.f("qacc_warmstart")[:] = w[i]; d.f("time")[0] = t[i]
    e2.set_state(qpos=q, qvel=v, time=t, warmstart=w)
    e2.step(1)
    for d in ds:
        d.step(1)
    _, q2, v2, w2 = e2.get_state(); st = e2.get_stats()
    over = st0[:, 1] > 96
    onefc = np.array([d.i("nefc") for d in ds]); owarn = np.array([d.i("warn") for d in ds])
    assert ((st[:, 3] & 2) != 0).tolist() == over.tolist(), "capacity flag exactly on the envs whose rows were dropped"
    assert ((owarn & 2) != 0).tolist() == over.tolist()
    assert (onefc <= 96).all() and (onefc[over] >= 96 - 5).all(), "the oracle kept whole contacts up to its capacity"
    vo = np.array([d.f("qvel") for d in ds]); qo = np.array([d.f("qpos") for d in ds])
    ev = np.abs(v2 - vo).max(1) / np.maximum(1.0, np.abs(vo).max(1)); eq = np.abs(q2 - qo).max(1)
    print(f"WINDOW-CLAMP envs over capacity {int(over.sum())} of {len(seeds)}: qvel err over / within capacity {ev[over].max():.2e} / {ev[~over].max():.2e}, qpos {eq.max():.2e}")
    # (a row set cut inside a pyramid differs from the oracle's by the force of the half-kept contact: 1e-2 .. 1e-1 in qvel on these piles)
    assert ev.max() <= 5e-4 and eq.max() <= 1e-6, (ev.max(), eq.max())
    e2.close()


def test_graph_recapture_while_the_previous_graph_is_still_running():
    """ADVICE r05 (engine.hip:937): a change of what a captured chain holds by value (here: xfrc_applied allocated, then the cohort count)
    re-captures the graph while the launches of the old one may still be running on the cohort streams — the retired exec is destroyed only
    after its stream has drained.  Same results as the plain launches, bitwise, with re-captures forced every few steps and no synchronisation
    in between."""
    lib = ms.capi.load()
    outs = []
    for graph in (2, 0):
        lib.mjh_set_chain_graph(graph)
        try:
            m = ms.scene("s24"); nenv = 1536
            e = ms.Engine(m, nenv); e.load_s24(); e.set_cohorts(3)
            for k in range(24):
                e.step(5)                                   # asynchronous: nothing waits for these launches ...
                e.set_cohorts(2 + k % 3)                    # ... before the env ranges change and every slot is captured again
            assert e.launches_per_step == (1 if graph else 2), "what the LAST step issued per cohort-step"
            outs.append(e.get_state())
            e.close()
        finally:
            lib.mjh_set_chain_graph(1)
    for x, y in zip(*outs):
        assert np.array_equal(x, y)


def _stub():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_rccl_sequence import build_stub
    return build_stub()


_GROUP_SCRIPT = r"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import mujoco_sim_amd as ms
from mujoco_sim_amd.tables import load_model_tables
lib = ms.capi.load()
stub = C.CDLL(os.environ["MJH_RCCL_LIB"])
m, z = load_model_tables(os.path.join({root!r}, "tests", "golden", "robot_c5_pendulum_bowl_mesh.npz"))
nenv, nshard = 4096, 8
spin = z["qvel0"][None, :] * np.random.default_rng(0xC5).uniform(0.5, 1.5, size=(nenv, 1))
res = {{}}
pubs = {{}}
for label, transport, threads in (("rccl-stub/thread-per-rank", 2, 1), ("rccl-stub/grouped", 2, 0), ("peer-copies", 1, 1)):
    stub.stub_reset()
    lib.mjh_group_set_transport(transport); lib.mjh_group_set_host_threads(threads)
    g = ms.Group(m, nenv, [0] * nshard)
    lib.mjh_group_set_transport(0); lib.mjh_group_set_host_threads(1)
    for (e0, n), e in zip(g.ranges, g.engines):
        e.set_controlled_dofs(z["controlled"].astype(np.int32)); e.set_state(qvel=spin[e0:e0 + n])
    for k in range(12):
        g.step(3, True)
        try:
            g.publish_device()
        except Exception as ex:
            he = (C.c_int * 2)(); stub.stub_hip_error(he)
            raise RuntimeError(f"{{label}}: {{ex}}; stub's first failing HIP call: code {{he[0]}} where {{he[1]}}")
        for r in range(nshard):
            g.wait_publish(r); g.release_publish(r)
    pubs[label] = g.publish()
    cnt = (C.c_int * 8)(); stub.stub_counters(cnt)
    res[label] = {{"uses_rccl": bool(g.uses_rccl), "threads": int(lib.mjh_group_host_threads(g.h)), "allgathers": int(stub.stub_nlog()), "counters": list(cnt)}}
    g.close()
ref = pubs["peer-copies"]
res["equal"] = {{k: bool(np.array_equal(v, ref)) for k, v in pubs.items()}}
res["spread"] = float(np.abs(ref[:, -9:]).max())
print(json.dumps(res))
"""


def test_eight_ranks_through_the_rccl_code_path_on_one_device_equal_the_peer_copy_transport():
    """the RCCL path of mjh_group_publish has only ever run with ONE rank (RCCL refuses the same device twice and no
    multi-GPU node is available).  With tests/nccl_stub as the library (MJH_RCCL_LIB; NCCL_STUB_COPY=1: the stand-in performs the
    all-gather as device copies with the call's stream semantics) the group runs C5 as EIGHT ranks on device 0 through exactly the code an
    8-GPU node runs — ncclCommInitAll over 8 devices, then per publish either 8 per-thread ncclAllGather calls or one grouped call — and
    the gathered state equals the peer-copy transport's bit for bit, consumers waiting for and releasing every publish."""
    env = dict(os.environ); env["MJH_RCCL_LIB"] = _stub(); env["NCCL_STUB_COPY"] = "1"
    r = subprocess.run([sys.executable, "-c", _GROUP_SCRIPT.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    a, b, c = res["rccl-stub/thread-per-rank"], res["rccl-stub/grouped"], res["peer-copies"]
    assert a["uses_rccl"] and b["uses_rccl"] and not c["uses_rccl"]
    assert a["threads"] == 8 and b["threads"] == 0
    assert a["counters"][6] == 1, "the stand-in performed the copies"
    assert a["allgathers"] == b["allgathers"] == 8 * 13 and c["allgathers"] == 0          # 12 publish_device + the final publish, 8 ranks each
    assert (a["counters"][3], a["counters"][4]) == (0, 0) and (b["counters"][3], b["counters"][4]) == (13, 13), "per-thread: no group; one thread: one group per publish"
    assert a["counters"][0] == b["counters"][0] == 1 and a["counters"][5] == 8, "one ncclCommInitAll over 8 ranks"
    assert res["equal"] == {"rccl-stub/thread-per-rank": True, "rccl-stub/grouped": True, "peer-copies": True}
    assert res["spread"] > 0.1


def test_device_against_the_oracle_built_on_the_independent_model():
    """VERDICT r05 weak #1b closed on the GPU side: the device steps the PRODUCT's compiled S24 / arm7 models, the oracle runs on a struct whose
    physics tables come from tests/indep_model.py (a second, numpy construction from the scene descriptions — no model_builder.cpp in it): the
    smoke test's comparison, with the compiler taken out of the oracle's side."""
    import ctypes as C
    from indep_model import compile_scene, scene_arm7, scene_s24
    from test_independent_model import _independent_model
    from mujoco_sim_amd.engine import EP
    # S24: 16 envs from reset, 40 steps (boxes land), per-env sizes / masses on both sides
    m = ms.scene("s24"); T = compile_scene(scene_s24())
    c2, keep = _independent_model(m, T)
    nenv, nsteps = 16, 40
    e = ms.Engine(m, nenv); tab = e.load_s24()
    e.step(nsteps); _, q, v, _ = e.get_state(); st = e.get_stats()
    worst = 0.0
    for i in range(nenv):
        d = orc.OrcData(C.pointer(c2))
        for k, wh in EP.items():
            d.set_env_param(wh, tab[k][i])
        d.set_qpos(tab["qpos"][i]); d.call("reset"); d.step(nsteps)
        worst = max(worst, float(np.abs(q[i] - d.f("qpos")).max()), float(np.abs(v[i] - d.f("qvel")).max()))
    print(f"INDEP-MODEL-GPU s24: {nenv} envs x {nsteps} steps, max |HIP fp32 - oracle fp64 on the independent model| = {worst:.2e}, contacts {st[:, 0].mean():.1f}")
    assert st[:, 0].mean() >= 1 and worst < 5e-3
    e.close()
    # arm7 (C3's model): computed-torque controller on every dof, 200 steps
    m = ms.scene("arm7", 1); T = compile_scene(scene_arm7(1))
    c2, keep = _independent_model(m, T)
    e = ms.Engine(m, 8); e.set_controlled_dofs(np.ones(m.nv, dtype=np.int32))
    ddq = np.tile(0.5 * np.sin(np.arange(m.nv)), (8, 1)) * np.linspace(0.5, 1.5, 8)[:, None]
    e.set_cmd(ddq=ddq); e.step(200, True); _, q, v, _ = e.get_state()
    worst = 0.0
    for i in range(8):
        d = orc.OrcData(C.pointer(c2)); d.call("reset"); d.ifield("controlled")[:] = 1; d.f("ddq")[:] = ddq[i]
        d.step(200, 1)
        worst = max(worst, float(np.abs(q[i] - d.f("qpos")).max()), float(np.abs(v[i] - d.f("qvel")).max()))
    print(f"INDEP-MODEL-GPU arm7: 8 envs x 200 steps, max |HIP fp32 - oracle fp64 on the independent model| = {worst:.2e}")
    assert worst < 2e-3
    e.close()


@pytest.mark.parametrize("name", ["s24", "s24d"])
def test_every_env_of_the_metrics_batch_one_step_against_the_oracle(name):
    """BASELINE size, not a sample: ALL 4096 envs of the metric's scene (S24, and S24D in its round-6 window forms), settled 400 steps on the
    device on three cohorts exactly as bench.py times them, then at three points of the timed regime every env's state goes to the oracle and
    both advance one step (the oracle on the host's thread team, 512 envs at a time): 12 288 env-steps each.  Contact sets must agree on >= 97 %
    of the env-steps (counts; where the counts agree and the result does not, the contact RECORDS are compared as well: fp32 and fp64 geometry
    keep different points of a manifold on a handful of env-steps); on those, qpos / qvel within the teacher-forced tolerances (qvel: 99 % quantile and maximum, the maximum by row class as in
    test_s24d_teacher_forced...: fp32 round-off at the 100-sweep cap grows with the row count)."""
    import ctypes as C
    from mujoco_sim_amd.engine import EP
    L = orc.lib()
    nenv = 4096
    if name == "s24":
        m = ms.scene("s24"); e = ms.Engine(m, nenv); tab = e.load_s24()
    else:
        m, e, tab = _s24d_seeds(list(range(nenv)))
    e.set_cohorts(3)
    e.step(400)
    B = 512
    ds = [orc.OrcData(m.ptr) for _ in range(B)]
    arr = (C.c_void_p * B)(*[d.d for d in ds])
    L.orc_set_threads(min(16, os.cpu_count() or 1))
    from test_gpu_teacher_forced import _same_contacts
    eq_all, ev_all, ag_all, rows_all = [], [], [], []
    rechecked = differ = 0
    for rep in range(3):
        t, q, v, w = e.get_state()
        e.step(1)
        _, q1, v1, _ = e.get_state(); st = e.get_stats()
        for b0 in range(0, nenv, B):
            for k, d in enumerate(ds):
                i = b0 + k
                for key, wh in EP.items():
                    d.set_env_param(wh, tab[key][i])
                d.f("qpos")[:] = q[i]; d.f("qvel")[:] = v[i]; d.f("qacc_warmstart")[:] = w[i]; d.f("qacc")[:] = w[i]; d.f("time")[0] = t[i]
            L.orc_step_many(arr, B, 1, 0)
            for k, d in enumerate(ds):
                i = b0 + k
                same = bool(st[i, 0] == d.i("ncon") and st[i, 1] == d.i("nefc") and (st[i, 3] & 7) == 0)
                eqi = np.abs(q1[i] - d.f("qpos")).max() / max(1.0, np.abs(d.f("qpos")).max())
                evi = np.abs(v1[i] - d.f("qvel")).max() / max(1.0, np.abs(d.f("qvel")).max())
                if same and (evi > 1e-4 or eqi > 1e-6):
                    # equal COUNTS do not make equal contact SETS (fp32 and fp64 geometry can keep different points of one manifold): the device's
                    # contact records at the state both started from, read from a one-env engine, against the oracle's (as teacher_forced does for
                    # every env-step of its smaller samples)
                    if name == "s24":
                        e1 = ms.Engine(m, 1); e1.load_tables({kk: tab[kk][i:i + 1] for kk in tab})
                    else:
                        _, e1, _ = _s24d_seeds([i])
                    e1.set_state(qpos=q[i:i + 1], qvel=v[i:i + 1], time=t[i:i + 1], warmstart=w[i:i + 1])
                    same = _same_contacts(e1.get_contacts(0), d.contacts()); rechecked += 1; differ += int(not same)
                    e1.close()
                ag_all.append(same); eq_all.append(eqi); ev_all.append(evi)
                rows_all.append(d.i("nefc"))
        e.step(29)
    eq, ev, ag, rows = np.array(eq_all), np.array(ev_all), np.array(ag_all, bool), np.array(rows_all)
    line = {"scene": name, "env_steps": int(len(ag)), "agree_fraction": float(ag.mean()), "qpos_max": float(eq[ag].max()), "qvel_q50_q99_max": [float(x) for x in np.quantile(ev[ag], [0.5, 0.99, 1.0])],
            "qvel_max_by_rows": {f"{lo}-{hi}": float(ev[ag & (rows >= lo) & (rows <= hi)].max()) for lo, hi in ((0, 96), (97, 128), (129, 192), (193, 256), (257, 400)) if (ag & (rows >= lo) & (rows <= hi)).any()},
            "qvel_q99_by_rows": {f"{lo}-{hi}": float(np.quantile(ev[ag & (rows >= lo) & (rows <= hi)], 0.99)) for lo, hi in ((0, 96), (97, 128), (129, 192), (193, 256), (257, 400)) if (ag & (rows >= lo) & (rows <= hi)).any()},
            "rows_mean_max": [float(rows.mean()), int(rows.max())], "count_agreeing_env_steps_rechecked_by_contact_records": rechecked, "of_them_with_other_records": differ}
    print("FULL-BATCH", json.dumps(line))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "full_batch_parity.jsonl"), "a") as f:
            f.write(json.dumps(line) + "\n")
    except OSError:
        pass
    assert ag.mean() >= 0.97, line
    # measured (round 6, 12 288 env-steps each): S24 agree 99.7 %, qpos 3.0e-7, qvel 50 % / 99 % / max 2.0e-6 / 9.0e-6 / 7.6e-5; S24D agree 98.9 %, qpos 9.9e-7,
    # qvel 2.0e-6 / 2.8e-5 / 3.6e-4 — the 99 % quantile by row class 8e-6 (<= 96 rows) / 2.5e-5 (<= 128) / 3.2e-5 (<= 192) / 4.2e-5 (<= 256): fp32 round-off of an
    # iteration that the 100-sweep cap ends before it has converged grows with the row count, whichever window form sweeps the env (32-row section on and
    # threshold 208: 2.7e-5; 16-row form only: 2.8e-5).  The 64-env sample of test_s24d_teacher_forced... reads 7.9e-6 at its 99 % quantile: the whole batch is
    # the figure to quote.
    tol_q, tol_v99 = (1e-6, 2e-5) if name == "s24" else (2e-6, 5e-5)
    assert eq[ag].max() <= tol_q and np.quantile(ev[ag], 0.99) <= tol_v99 and ev[ag].max() <= 5e-4, line
    e.close()


def test_every_env_of_the_c4_batch_one_step_against_the_oracle():
    """C4 (PR2 + world + object pool, mj_inverse every step, per-env commands) at BASELINE size, every one of the 2048 envs handed to the oracle for
    one step — round 3's test of the same set-up sampled 32 of them"""
    from test_gpu_round3 import _one_step_on_samples, _robot
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
    eq, ev, ag, ds = _one_step_on_samples(e, make, list(range(nenv)), with_inverse=True)
    line = {"scene": "c4", "env_steps": nenv, "agree_fraction": float(ag.mean()), "qpos_max": float(eq[ag].max()), "qvel_q50_q99_max": [float(x) for x in np.quantile(ev[ag], [0.5, 0.99, 1.0])],
            "rows_min_max": [min(d.i("nefc") for d in ds), max(d.i("nefc") for d in ds)], "not_agreeing_qpos_max": float(eq[~ag].max()) if (~ag).any() else 0.0}
    print("FULL-BATCH", json.dumps(line))
    try:
        with open(os.path.join(ROOT, "gpurun_out", "full_batch_parity.jsonl"), "a") as f:
            f.write(json.dumps(line) + "\n")
    except OSError:
        pass
    assert ag.mean() >= 0.9 and eq[ag].max() <= 2e-6 and ev[ag].max() <= 1e-4, line
    assert eq.max() < 1e-3
    e.close()


@pytest.mark.parametrize("name", ["c3", "c5"])
def test_every_env_of_the_small_model_batches_one_step_against_the_oracle(name):
    """C3 (8192 arms, four per wavefront, in-engine PD law + computed-torque wrapper + mj_inverse) and C5 (4096 pendulum worlds over the bowl) exactly as
    bench.py builds and settles them, then EVERY env one step against the oracle (the existing full-size tests of these configs check invariants)."""
    import ctypes as C
    import types
    sys.path.insert(0, ROOT)
    import bench
    args = types.SimpleNamespace(envs_per_gpu=0, pack=0, maxcon=0, pen_half=0.0)
    w = bench.WORKLOADS[name](ms, args, 0, 0, None)
    if w.cohorts > 0:
        w.eng.set_cohorts(w.cohorts)
    w.step(w.settle_steps + 21, w.inverse)
    nenv = w.nenv
    q, v, ws, st = w.env_state(nenv)
    tt = np.repeat(w.eng.get_state(0, w.rows)[0], w.pack)[:nenv]
    L = orc.lib(); L.orc_set_threads(min(16, os.cpu_count() or 1))
    B = 1024
    ds = [w.oracle_data(orc, i, None) for i in range(B)]
    arr = (C.c_void_p * B)(*[d.d for d in ds])
    w.step(1, w.inverse)
    q1, v1, _, st1 = w.env_state(nenv)
    eq, ev, ag = np.zeros(nenv), np.zeros(nenv), np.zeros(nenv, bool)
    for b0 in range(0, nenv, B):
        for k, d in enumerate(ds):
            i = b0 + k
            if name == "c3":
                d.set_pd(w.target[i], 200.0, 50.0)
            d.f("qpos")[:] = q[i]; d.f("qvel")[:] = v[i]; d.f("qacc_warmstart")[:] = ws[i]; d.f("qacc")[:] = ws[i]; d.f("time")[0] = tt[i]
        L.orc_step_many(arr, B, 1, int(w.inverse))
        for k, d in enumerate(ds):
            i = b0 + k
            eq[i] = np.abs(q1[i] - d.f("qpos")).max() / max(1.0, np.abs(d.f("qpos")).max()); ev[i] = np.abs(v1[i] - d.f("qvel")).max() / max(1.0, np.abs(d.f("qvel")).max())
            ag[i] = w.pack > 1 or (st1[i, 0] == d.i("ncon") and st1[i, 1] == d.i("nefc"))
    line = {"scene": name, "env_steps": int(nenv), "agree_fraction": float(ag.mean()), "qpos_max": float(eq[ag].max()), "qvel_q50_q99_max": [float(x) for x in np.quantile(ev[ag], [0.5, 0.99, 1.0])],
            "envs_in_contact": int((st1[:, 0] > 0).sum())}
    print("FULL-BATCH", json.dumps(line))
    try:
        with open(os.path.join(ROOT, "gpurun_out", "full_batch_parity.jsonl"), "a") as f:
            f.write(json.dumps(line) + "\n")
    except OSError:
        pass
    assert ag.mean() >= 0.99 and eq[ag].max() <= 1e-6 and ev[ag].max() <= 5e-5, line
    w.eng.close()


def test_c2_on_d3s_window_256_envs_one_step_against_the_oracle():
    """C2 where bench.py times it since round 6 — behind 200 settle + 500 steps, 262 contacts / 1338 rows per env instead of the 185 / 907 of the
    200-step state the older tests meet — 256 envs spread over the 4096-env batch, one step against the oracle (16 host threads)."""
    import ctypes as C
    from mujoco_sim_amd.engine import EP
    m = ms.scene("boxpile", 64); m.c.maxcon = 600; m.c.maxefc = 2400
    nenv = 4096
    e = ms.Engine(m, nenv); e.set_cohorts(4)
    tab = e.load_tables(ms.boxes_randomize(m, 0, nenv, jitter=0.01))
    e.step(700)
    sample = list(range(7, nenv, 16))
    t, q, v, w = e.get_state()
    e.step(1)
    _, q1, v1, _ = e.get_state(); st = e.get_stats()
    L = orc.lib(); L.orc_set_threads(min(16, os.cpu_count() or 1))
    B = 32
    ds = [orc.OrcData(m.ptr) for _ in range(B)]
    arr = (C.c_void_p * B)(*[d.d for d in ds])
    from test_gpu_teacher_forced import _same_contacts
    eq, ev, ag, rows = [], [], [], []
    rechecked = differ = 0
    for b0 in range(0, len(sample), B):
        for k, d in enumerate(ds):
            i = sample[b0 + k]
            for key, wh in EP.items():
                d.set_env_param(wh, tab[key][i])
            d.f("qpos")[:] = q[i]; d.f("qvel")[:] = v[i]; d.f("qacc_warmstart")[:] = w[i]; d.f("qacc")[:] = w[i]; d.f("time")[0] = t[i]
        L.orc_step_many(arr, B, 1, 0)
        for k, d in enumerate(ds):
            i = sample[b0 + k]
            same = bool(st[i, 0] == d.i("ncon") and st[i, 1] == d.i("nefc") and (st[i, 3] & 7) == 0); rows.append(d.i("nefc"))
            eqi = np.abs(q1[i] - d.f("qpos")).max() / max(1.0, np.abs(d.f("qpos")).max()); evi = np.abs(v1[i] - d.f("qvel")).max() / max(1.0, np.abs(d.f("qvel")).max())
            if same and (evi > 5e-5 or eqi > 1e-6):       # equal counts, other result: are the ~260 contact RECORDS the same?  (one-env engine at the state both started from)
                e1 = ms.Engine(m, 1); e1.load_tables({kk: tab[kk][i:i + 1] for kk in tab})
                e1.set_state(qpos=q[i:i + 1], qvel=v[i:i + 1], time=t[i:i + 1], warmstart=w[i:i + 1])
                same = _same_contacts(e1.get_contacts(0), d.contacts()); rechecked += 1; differ += int(not same)
                e1.close()
            ag.append(same); eq.append(eqi); ev.append(evi)
    eq, ev, ag, rows = np.array(eq), np.array(ev), np.array(ag), np.array(rows)
    line = {"scene": "c2 (D3 window: 700 steps)", "env_steps": int(len(ag)), "agree_fraction": float(ag.mean()), "qpos_max": float(eq[ag].max()), "qvel_q50_q99_max": [float(x) for x in np.quantile(ev[ag], [0.5, 0.99, 1.0])],
            "rows_mean_max": [float(rows.mean()), int(rows.max())], "not_agreeing_qpos_max": float(eq[~ag].max()) if (~ag).any() else 0.0,
            "count_agreeing_env_steps_rechecked_by_contact_records": rechecked, "of_them_with_other_records": differ}
    print("FULL-BATCH", json.dumps(line))
    try:
        with open(os.path.join(ROOT, "gpurun_out", "full_batch_parity.jsonl"), "a") as f:
            f.write(json.dumps(line) + "\n")
    except OSError:
        pass
    assert rows.mean() > 1100 and ag.mean() >= 0.85 and eq[ag].max() <= 1e-6 and np.quantile(ev[ag], 0.99) <= 2e-5 and ev[ag].max() <= 2e-4, line
    e.close()


def test_dense_sweeps_of_every_row_class_one_step_against_the_oracle():
    """The dense row-space sweeps (dense_pgs.h) in each of their forms, by the rows an env brings: up to 128 rows the lane's part of AR' stays
    in registers (two rows per lane), 129 - 192 and 193 - 256 rows stream it (three / four rows per lane; the pair + single form at three).
    C4's fixture with 0 .. 8 pool objects dropped beside the robot per env (128 envs: below 1024 envs every step takes the dense form), a
    short fall, then every env one step against the oracle from the device's own state — repeated while the objects land and settle, so
    that envs at the sweep cap are met in every class.  Round 6 changed how a row's visit value is kept inside the sweeps (one DPP move
    under a row / bank mask): this is the test that sees all forms of it."""
    from test_gpu_round3 import _one_step_on_samples, _robot
    from test_robot_fixtures import robot_command
    m, z = _robot("c4_pr2_world_objects_mesh")
    lib = m.lib
    nenv = 128
    e = ms.Engine(m, nenv)
    assert e.dense_solver() == 1 or os.environ.get("MJH_DENSE") == "0"
    e.set_controlled_dofs(z["controlled"].astype(np.int32))
    names = [lib.mjh_id2name(m.ptr, 0, b).decode() for b in range(m.c.nbody)]
    slots = [b for b, n in enumerate(names) if n.startswith("object_")]
    sbase = m.c.nbody - 32 if m.c.nbody > 32 else 0
    allmask = 0
    for b in slots:
        e.set_slot_active(b, False); allmask |= 1 << (b - sbase)
    mask = [allmask] * nenv
    rng = np.random.default_rng(606)
    for i in range(nenv):
        for j in range(i % (len(slots) + 1)):
            b = slots[j]
            a, r = rng.uniform(-np.pi, np.pi), rng.uniform(0.9, 1.4)
            quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
            e.set_slot_active(b, True, env0=i, n=1)
            e.set_body_pose(i, b, np.array([r * np.sin(a), r * np.cos(a), rng.uniform(0.15, 0.5)]), quat, np.array([0.1, -0.1, -0.3, *(0.5 * rng.normal(size=3))]))
            mask[i] &= ~(1 << (b - sbase))

    def make(i):
        d = orc.OrcData(m.ptr); d.ifield("controlled")[:] = z["controlled"]
        orc.lib().orc_set_slot_mask(d.d, mask[i])
        return d
    classes = {2: [0, 0, 0.0], 3: [0, 0, 0.0], 4: [0, 0, 0.0]}          # rows per lane -> env-steps compared, of them at >= 50 sweeps, worst qvel
    evs = {2: [], 3: [], 4: []}
    worst_q = 0.0; agreeing = total = 0; step = 0
    for rep in range(8):
        for k in range(6):
            step += 1
            e.set_cmd(ddq=np.tile(robot_command(m, step), (nenv, 1))); e.step(1, True)
        eq, ev, ag, ds = _one_step_on_samples(e, make, list(range(nenv)), with_inverse=True)
        st = e.get_stats()
        assert (st[:, 3] & 7 == 0).all()
        for i, d in enumerate(ds):
            total += 1
            if not ag[i]:
                continue
            agreeing += 1
            rows = d.i("nefc"); kk = 2 if rows <= 128 else (3 if rows <= 192 else 4)
            if rows > 256:
                continue                                       # (beyond the dense capacity: the block solver's env)
            c = classes[kk]; c[0] += 1; c[1] += int(st[i, 2] >= 50); c[2] = max(c[2], float(ev[i])); worst_q = max(worst_q, float(eq[i])); evs[kk].append(float(ev[i]))
    quant = {kk: [float(x) for x in np.quantile(v, [0.5, 0.9, 0.99])] for kk, v in evs.items() if v}
    print("DENSE-QUANT", quant)
    print(f"DENSE-FORMS {total} env-steps, contact sets agree {agreeing / total:.3f}; by rows per lane (env-steps, of them >= 50 sweeps, worst qvel): {classes}, qpos {worst_q:.2e}")
    # measured (MI355X): 1024 env-steps, 716 with two rows per lane (465 of them at >= 50 sweeps), 308 with three (174); qvel 50 / 90 / 99 % quantiles 5e-7 / 1.7e-6 / 4.0e-6
    # (two rows per lane) and 3.6e-7 / 2.1e-6 / 1.7e-3 (three) — the block solver on the same states: 4e-7 / 1.8e-6 / 3.3e-6 and 2.8e-7 / 2.0e-6 / 4.1e-4; the worst single
    # env-step 1.1e-2 / 1.6e-2 (block solver 1.4e-4 / 4.8e-3): an object's first touch-down at the sweep cap, where an fp32 iterate that is still moving is compared
    # with an fp64 one (the carried residual formed anew every 8 sweeps, -DDN_REFRESH=8, takes the two-row worst case to 2.9e-3 and leaves the quantiles where they are)
    assert agreeing / total >= 0.9 and worst_q <= 5e-4
    for kk in (2, 3):
        assert classes[kk][0] >= 20 and classes[kk][1] >= 1, (kk, classes)
    assert quant[2][2] <= 2e-5 and quant[3][1] <= 1e-5 and quant[3][2] <= 1e-2 and all(c[2] <= 5e-2 for c in classes.values()), (quant, classes)
    e.close()
