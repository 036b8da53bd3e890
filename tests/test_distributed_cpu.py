"""world_size-2 `gloo` test of the multi-GPU path's host logic (CPU): env sharding and the
all-gather of the published state slice.  The step itself needs no collective."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT
from mujoco_sim_amd import shard

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import mujoco_sim_amd as ms
from mujoco_sim_amd import shard
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ok = True
for total in (10, 7):                       # even and uneven shares
    lo, hi = shard.env_range(total, world, rank)
    n = hi - lo
    # this rank's shard as the engine lays it out in HBM (csrc/engine.hip: one padded record per env, qpos | qvel | ...):
    # the S24 model's sizes, per-env content drawn by the scene's own randomiser with GLOBAL env ids
    m = ms.scene("s24")
    nq, nv = m.nq, m.nv
    nqp = ((nq + 4 * nv + 31) // 32) * 32
    tab = m.s24_randomize(lo, n)            # env_offset = lo: rank r owns the global envs [lo, hi)
    rec = np.zeros((n, nqp), dtype=np.float32)
    rec[:, :nq] = tab["qpos"]
    rec[:, nq:nq + nv] = (np.arange(lo, hi)[:, None] * 100 + np.arange(nv)[None, :]).astype(np.float32)   # qvel: env id + dof id
    time = (0.005 * (np.arange(lo, hi) + 1)).astype(np.float32)
    # mjh_export_kernel's packing (csrc/step_kernel.h): out[i] with e = i // stride, k = i % stride:
    #   k == 0 -> time[e];  k <= nq -> qpos[e * nqp + k - 1];  else qvel[e * nvp + k - 1 - nq]  (qvel = column nq of the record)
    stride = 1 + nq + nv
    flat = np.empty(n * stride, dtype=np.float32)
    for i in range(n * stride):
        e, k = divmod(i, stride)
        flat[i] = time[e] if k == 0 else (rec[e, k - 1] if k <= nq else rec[e, nq + (k - 1 - nq)])
    full = shard.gather_state(torch.from_numpy(flat.reshape(n, stride)), total, world, rank).numpy()
    ref = ms.scene("s24").s24_randomize(0, total)["qpos"].astype(np.float32)      # what ONE engine over all envs would hold
    ok &= full.shape == (total, stride)
    ok &= bool(np.array_equal(full[:, 0], (0.005 * (np.arange(total) + 1)).astype(np.float32)))
    ok &= bool(np.array_equal(full[:, 1:1 + nq], ref))                             # env order across the ranks, row by row
    ok &= bool(np.array_equal(full[:, 1 + nq:], (np.arange(total)[:, None] * 100 + np.arange(nv)[None, :]).astype(np.float32)))
# C5 — the reference's multi-GPU config (launch/multi_mujoco_sim.launch:9-34) — at uneven shares: the fixture's sizes, per-env spin
# drawn per GLOBAL env id (what tests/test_gpu_round4.py's eight-shard run and bench.py --host group hand to the shards)
import os
from mujoco_sim_amd.tables import load_model_tables
m5, z5 = load_model_tables(os.path.join(sys.argv[1], "tests", "golden", "robot_c5_pendulum_bowl_mesh.npz"))
nq5, nv5 = m5.nq, m5.nv
for total in (9, 4096 + 3):
    lo, hi = shard.env_range(total, world, rank)
    spin_all = (z5["qvel0"][None, :] * np.random.default_rng(0xC5).uniform(0.5, 1.5, size=(total, 1))).astype(np.float32)
    q0 = np.tile(m5.array("qpos0"), (total, 1)).astype(np.float32)
    q0[:, 0] += np.arange(total, dtype=np.float32) * 1e-3
    tt = (0.005 * np.arange(total)).astype(np.float32)
    local = np.concatenate([tt[lo:hi, None], q0[lo:hi], spin_all[lo:hi]], axis=1)
    full = shard.gather_state(torch.from_numpy(np.ascontiguousarray(local)), total, world, rank).numpy()
    ok &= full.shape == (total, 1 + nq5 + nv5)
    ok &= bool(np.array_equal(full, np.concatenate([tt[:, None], q0, spin_all], axis=1)))
tmax = shard.max_over_ranks(float(rank + 1))
print("RESULT", rank, int(ok), tmax)
dist.destroy_process_group()
'''


def test_env_range_partition():
    for total, world in ((4096, 8), (10, 3), (7, 8), (32768, 8)):
        spans = [shard.env_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_gather_state_world2_gloo(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
        line = [l for l in o.splitlines() if l.startswith("RESULT")][0].split()
        assert line[2] == "1" and float(line[3]) == 2.0
