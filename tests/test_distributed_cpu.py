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
from mujoco_sim_amd import shard
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
total, stride = 10, 5
lo, hi = shard.env_range(total, world, rank)
# fake "exported state": row e = [time, e, e, e, e]
local = torch.tensor([[0.005 * (e + 1)] + [float(e)] * (stride - 1) for e in range(lo, hi)], dtype=torch.float32)
full = shard.gather_state(local, total, world, rank)
ok = full.shape == (total, stride) and bool((full[:, 1] == torch.arange(total)).all())
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
