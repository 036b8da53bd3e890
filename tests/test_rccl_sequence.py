"""The RCCL call sequence of the N-rank publish, asserted WITHOUT a multi-GPU node.

csrc/group.hip resolves its few RCCL entry points with dlopen; MJH_RCCL_LIB points it at tests/nccl_stub (a recording stand-in, test
infrastructure) and mjh_debug_rccl_exchange runs exactly the exchange code of mjh_group_publish (rccl_exchange) with N ranks and no device:
  * one host thread for all ranks: ncclGroupStart, N x ncclAllGather (rank order, every rank's own communicator / stream / buffers), ncclGroupEnd;
  * a host thread per rank: N ungrouped ncclAllGather per publish, every rank always from its own thread;
  * a failing rank: the grouped form still passes ncclGroupEnd; the per-thread form still enqueues every other rank (nobody returns early and
    leaves partners waiting) and then aborts the communicators.
(one all-gather of the published slice per publish, no collective in the step.)  Each case runs in a process of its own:
the library choice is made once per process."""
import json
import os
import subprocess
import sys

from conftest import ROOT

STUB_DIR = os.path.join(ROOT, "tests", "nccl_stub")


def build_stub():
    so = os.path.join(STUB_DIR, "libnccl_stub.so")
    src = os.path.join(STUB_DIR, "nccl_stub.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", so, src, "-lpthread", "-ldl"])
    return so


_SCRIPT = r"""
import ctypes as C, json, os, sys
sys.path.insert(0, {root!r})
from mujoco_sim_amd import capi
lib = capi.load()
stub = C.CDLL(os.environ["MJH_RCCL_LIB"])
class Rec(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("seq", "rank", "nranks", "dtype", "in_group", "group_id", "epoch")] + [("count", C.c_ulong)] + \
               [(n, C.c_ulonglong) for n in ("send", "recv", "stream", "comm", "tid")]
def run(ndev, per_thread, slot, publishes, fail=-1):
    stub.stub_reset()
    if fail >= 0: stub.stub_fail_rank(fail)
    rc = lib.mjh_debug_rccl_exchange(ndev, per_thread, slot, publishes)
    cnt = (C.c_int * 8)(); stub.stub_counters(cnt)
    recs = []
    for i in range(stub.stub_nlog()):
        r = Rec(); assert stub.stub_get(i, C.byref(r)) == 0
        recs.append({{n: getattr(r, n) for n, _ in Rec._fields_}})
    return {{"rc": rc, "err": lib.mjh_last_error().decode() if rc else "", "counters": list(cnt), "devs": [stub.stub_init_dev(k) for k in range(ndev)], "recs": recs}}
print(json.dumps([run(*a) for a in {cases!r}]))
"""


def _run(cases):
    env = dict(os.environ); env["MJH_RCCL_LIB"] = build_stub(); env.pop("NCCL_STUB_COPY", None)
    r = subprocess.run([sys.executable, "-c", _SCRIPT.format(root=ROOT, cases=cases)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_grouped_all_gather_sequence_for_eight_ranks():
    N, SLOT, P = 8, 217088, 3                               # S24's slot: 4096 envs x (1 + 28 + 24) floats per rank
    (res,) = _run([(N, 0, SLOT, P)])
    assert res["rc"] == 0, res["err"]
    n_init, n_destroy, n_abort, n_gs, n_ge, ndev = res["counters"][:6]
    assert (n_init, ndev, res["devs"]) == (1, N, list(range(N))), "one ncclCommInitAll over the N devices of the node"
    assert n_destroy == N and n_abort == 0
    assert n_gs == n_ge == P, "one ncclGroupStart / ncclGroupEnd pair per publish"
    recs = res["recs"]
    assert len(recs) == N * P
    assert len({r["tid"] for r in recs}) == 1, "the grouped form is issued by the caller's one thread"
    for p in range(P):
        chunk = recs[p * N:(p + 1) * N]
        assert [r["rank"] for r in chunk] == list(range(N)), "every rank once per publish, in rank order"
        assert all(r["in_group"] == 1 for r in chunk) and len({r["group_id"] for r in chunk}) == 1, "all N calls inside ONE group"
        assert all(r["count"] == SLOT and r["dtype"] == 7 and r["nranks"] == N for r in chunk), "slot floats of ncclFloat32 per rank"
    assert len({r["group_id"] for r in recs}) == P
    for k in range(N):
        mine = [r for r in recs if r["rank"] == k]
        assert len({(r["comm"], r["stream"], r["send"], r["recv"]) for r in mine}) == 1, "a rank keeps its communicator, stream and buffers"
    for f in ("comm", "stream", "send", "recv"):
        assert len({r[f] for r in recs}) == N, f"every rank has its own {f}"


def test_per_thread_all_gather_sequence_for_eight_ranks():
    N, SLOT, P = 8, 90112, 5                                # C5's slot: 4096 envs x (1 + 12 + 9) floats per rank
    (res,) = _run([(N, 1, SLOT, P)])
    assert res["rc"] == 0, res["err"]
    n_init, n_destroy, n_abort, n_gs, n_ge, ndev = res["counters"][:6]
    assert (n_init, ndev, n_destroy, n_abort) == (1, N, N, 0)
    assert n_gs == n_ge == 0, "a thread per device: no grouped call, every thread enqueues its own rank"
    recs = res["recs"]
    assert len(recs) == N * P and all(r["in_group"] == 0 and r["count"] == SLOT and r["dtype"] == 7 for r in recs)
    for p in range(P):                                      # the pool joins every publish: calls of publish p + 1 come after all of publish p
        assert sorted(r["rank"] for r in recs[p * N:(p + 1) * N]) == list(range(N))
    tids = {}
    for r in recs:
        tids.setdefault(r["rank"], set()).add(r["tid"])
    assert all(len(t) == 1 for t in tids.values()), "a rank is always enqueued by the same host thread"
    assert len({next(iter(t)) for t in tids.values()}) == N, "N ranks, N threads"


def test_a_failing_rank_leaves_no_partner_waiting():
    N, SLOT = 8, 1024
    grouped, threaded = _run([(N, 0, SLOT, 1, 3), (N, 1, SLOT, 1, 3)])
    # grouped: the failure is remembered, ncclGroupEnd is still passed, the error comes back
    assert grouped["rc"] != 0 and "ncclAllGather" in grouped["err"]
    assert grouped["counters"][3] == grouped["counters"][4] == 1
    assert [r["rank"] for r in grouped["recs"]] == [0, 1, 2, 3], "the group is closed at the failing rank"
    # a thread per rank: every OTHER rank's all-gather was enqueued all the same (no thread returned before its collective), then abort
    assert threaded["rc"] != 0 and "ncclAllGather" in threaded["err"]
    assert sorted(r["rank"] for r in threaded["recs"]) == list(range(N))
    assert threaded["counters"][2] == N, "ncclCommAbort on every communicator after a partial failure"
    assert threaded["counters"][1] == 0, "aborted communicators are not destroyed a second time"
