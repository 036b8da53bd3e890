"""CPU tests of bench.py's host logic: the plain `--gpus N` invocation becomes an N-rank launch, the usable-core count, and the
cpu_baseline leg (oracle timed on a team of spinning threads: thread-scaling table, >= 256 env-steps behind the 1-thread figure)."""
import os
import re
import subprocess
import sys

import numpy as np

import mujoco_sim_amd as ms
import orc
from conftest import ROOT
from mujoco_sim_amd.engine import EP

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_plain_multi_gpu_invocation_spawns_one_rank_per_gpu():
    """`python bench.py --gpus 2` with no launcher around it re-executes itself under torch.distributed.run (127.0.0.1
    rendezvous): both ranks come up with WORLD_SIZE 2.  --rank-probe makes every rank report and exit before it needs a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for attempt in range(2):        # (the rendezvous port is picked free a moment before the launcher binds it: one retry)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rank-probe"],
                           env=env, capture_output=True, text=True, timeout=300)
        if r.returncode == 0:
            break
    assert r.returncode == 0, r.stderr[-2000:]
    probes = sorted(list(t) for t in re.findall(r"RANKPROBE (\d+) (\d+) (\d+)", r.stdout))
    assert probes == [["0", "2", "0"], ["1", "2", "1"]], r.stdout


def test_launcher_command_is_the_drivers_launch():
    cmd = bench.launcher_command(["--gpus", "4", "--steps", "20"], 4, port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "20"] and cmd[-5].endswith("bench.py")


def test_group_host_is_refused_under_a_launcher():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--host", "group"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "ONE process" in r.stderr


def test_usable_cpus_is_what_the_process_may_run_on():
    n, info = bench.usable_cpus()
    assert 1 <= n <= info["sched_affinity"] <= info["os_cpu_count"]
    assert n == info["usable"]


class _SettledS24:
    """stand-in for bench.Workload: S24 envs settled in the oracle itself (no GPU here)"""
    name = "s24"; nenv = 32

    def __init__(self):
        self.model = ms.scene("s24"); self.base_model = self.model
        self.tab = self.model.s24_randomize(0, self.nenv)
        self.q = np.zeros((self.nenv, self.model.nq)); self.v = np.zeros((self.nenv, self.model.nv)); self.ws = np.zeros_like(self.v)
        self.st = np.zeros((self.nenv, 4))
        for i in range(self.nenv):
            d = self.oracle_data(orc, i, None)
            d.set_qpos(self.tab["qpos"][i]); d.call("reset"); d.step(150)
            self.q[i] = d.f("qpos"); self.v[i] = d.f("qvel"); self.ws[i] = d.f("qacc_warmstart"); self.st[i, :2] = (d.i("ncon"), d.i("nefc"))

    def env_state(self, n):
        return self.q[:n], self.v[:n], self.ws[:n], self.st[:n]

    def oracle_data(self, orc_, i, state):
        d = orc_.OrcData(self.model.ptr)
        for k, wh in EP.items():
            d.set_env_param(wh, self.tab[k][i])
        return d


def test_cpu_baseline_reports_a_thread_scaling_table():
    w = _SettledS24()
    r = bench.cpu_baseline(w, 32, 3.0, False)
    n, _ = bench.usable_cpus()
    assert r["kind"] == "port" and r["cores"] == n == r["scaling"][-1]["threads"] and r["scaling"][0]["threads"] == 1
    assert r["env_steps_1thread"] >= 256, "the 1-thread figure must rest on a real sample"
    assert all(p["value"] > 0 and p["envs"] >= p["threads"] or p["envs"] == 32 for p in r["scaling"])
    # (the leg steps its 32 envs for a time budget, so how far they have moved on depends on the host's speed: measured 1 .. 3.1 contacts of drift — a leg that
    #  started from the reset state instead, boxes in the air, would be ~16 contacts off)
    assert abs(r["mean_ncon"] - r["gpu_mean_ncon_same_envs"]) < 6, "the CPU leg starts from the state it was handed"
    # (no bound on the speed-up here: 32 envs on a shared CI host are too noisy a sample — the table is what the bench line carries)
    assert all("speedup_vs_1thread" in p for p in r["scaling"])


def test_timed_team_steps_every_env_exactly_as_the_plain_loop():
    """orc_step_many_timed (spinning pthread team, (env, 4-step) items) advances every env by warm + n steps, bit for bit"""
    import ctypes as C
    m = ms.scene("s24"); tab = m.s24_randomize(0, 6)
    def make():
        out = []
        for i in range(6):
            d = orc.OrcData(m.ptr)
            for k, wh in EP.items():
                d.set_env_param(wh, tab[k][i])
            d.set_qpos(tab["qpos"][i]); d.call("reset"); out.append(d)
        return out
    a, b = make(), make()
    L = orc.lib(); L.orc_set_threads(3)
    dt = L.orc_step_many_timed((C.c_void_p * 6)(*[d.d for d in a]), 6, 2, 11, 0)
    assert dt > 0
    for d in b:
        d.step(13)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x.f("qpos"), y.f("qpos")); np.testing.assert_array_equal(x.f("qvel"), y.f("qvel"))
    L.orc_set_threads(1)
