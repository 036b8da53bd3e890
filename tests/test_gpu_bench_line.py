"""The driver's contract, end to end on the GPU box: `python bench.py` prints ONE JSON line whose metric is BASELINE.json's, with the
roofline and cpu_baseline objects, the literal loop and the other configs riding along; the one-process group host runs without a
launcher.  (Short windows: this checks the shape and sanity of the line, not the numbers.)"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    assert r.stdout.strip().splitlines()[-1] == lines[0], "the JSON line is the LAST line of stdout"
    return json.loads(lines[0])


def test_driver_line_has_everything_the_contract_names():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    r = _run(["--steps", "12", "--warmup", "3", "--cpu-seconds", "4", "--extra-steps", "6"])
    assert r["metric"] == base["metric"] and r["unit"] == "env-steps/s" and r["n_gpus"] == 1 and r["steps"] == 12 and r["warmup"] == 3
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None and r["dtype"] == "f32" and r["data"] == "synthetic"
    assert r["value"] > 1e6 and abs(r["value"] - 4096 * 12 / (r["ms_per_step"] * 12e-3)) < 1e-3 * r["value"]
    c = r["config"]
    assert c["name"] == "s24" and c["envs_per_gpu"] == 4096 and c["settle_steps"] == 400 and 8 <= c["mean_ncon"] <= 48 and c["overflow_envs"] == 0
    assert "unsettled" not in r
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and rf["algorithmic_bytes_per_env_step"] == 800
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["kernel_ms"] > 0 and rf["launches_timed"] > 0
    assert abs(rf["achieved"] - 800 * rf["envs_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert rf["traffic"] is None or rf["traffic"] > 800 * rf["envs_per_launch"]
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == cb["scaling"][-1]["threads"] == cb["host"]["usable"] and cb["scaling"][0]["threads"] == 1
    assert cb["env_steps_1thread"] >= 256 and cb["value"] > 0 and abs(cb["mean_ncon"] - cb["gpu_mean_ncon_same_envs"]) < 4
    ll = r["literal_loop"]
    # (no ordering between the two is asserted: with the split entry points on cohort streams the literal loop is the faster one in a
    #  process that holds PyTorch's HIP runtime, the fused step in one that does not — HISTORY.md, Round 3)
    assert ll["value"] > 1e5 and ll["fused_step_with_per_step_read_write"]["value"] > 1e5
    assert set(r["configs"]) == {"s24d", "c2", "c3", "c4", "c5", "s24_legacy_patch_order", "s24_row_order_sequential"}
    for name, line in r["configs"].items():
        assert "error" not in line, (name, line)
        assert line["value"] > 0 and line["steps"] >= 6 and line["roofline_frac"] > 0 and line["overflow_envs"] <= 0.02 * line["envs"], (name, line)
    assert r["configs"]["s24d"]["overflow_envs"] == 0, "S24D: no env over the contact / row capacity (VERDICT r04 next #1)"
    # VERDICT r05 #1a / #5: every extra runs as a process of its own (the stand-alone run's streams) and carries its whole roofline block;
    # the 30-contact scene sits at the top level of the line, with its own roofline block and the measured contact / row histograms
    for name, line in r["configs"].items():
        rf2 = line["roofline"]
        assert line["process"].startswith("own") and rf2["kernel_ms"] > 0 and abs(rf2["frac"] - line["roofline_frac"]) < 1e-15, name
        assert rf2["traffic"] is None or rf2["traffic"] > rf2["algorithmic_bytes_per_env_step"] * rf2["envs_per_launch"], name
        assert rf2["traffic"] is None or (rf2["valu_issue_frac"] and 0 < rf2["valu_issue_frac"] < 1 and 0 < rf2["valu_lane_util"] <= 1), name
    assert r["value_30_contact"] == r["configs"]["s24d"]["value"] and r["roofline_30_contact"] == r["configs"]["s24d"]["roofline"]
    c30 = r["config_30_contact"]
    assert c30["overflow_envs"] == 0 and 24 <= c30["mean_ncon"] <= 40 and sum(c30["ncon_histogram"].values()) == 4096 and sum(c30["nefc_histogram"].values()) == 4096
    assert sum(r["configs"]["c2"]["ncon_histogram"].values()) == 4096
    assert r["configs"]["s24d"]["mean_ncon"] >= 20 and r["configs"]["c2"]["mean_ncon"] >= 100 and r["configs"]["c4"]["mean_nefc"] >= 50


def test_group_host_runs_in_one_process_without_a_launcher():
    r = _run(["--gpus", "2", "--host", "group", "--group-devices", "0,0", "--envs-per-gpu", "1024", "--steps", "12", "--warmup", "3"])
    h = r["host"]
    assert r["n_gpus"] == 2 and h["kind"] == "group" and [x["nenv"] for x in h["ranks"]] == [1024, 1024] and h["ranks"][1]["env0"] == 1024
    assert h["rccl"] is False and "peer copies" in h["transport"] and h["all_gather"]["count"] >= 3 and h["all_gather"]["ms_mean"] > 0
    assert r["value"] > 1e6 and r["config"]["envs_total"] == 2048
    r1 = _run(["--gpus", "1", "--host", "group", "--group-devices", "0", "--envs-per-gpu", "1024", "--steps", "12", "--warmup", "3"])
    assert r1["host"]["rccl"] is True and r1["host"]["rccl_ranks"] == 1, "one device: the publish goes through ncclAllGather (the 8-GPU code path)"


def test_group_host_bench_with_eight_ranks_through_the_rccl_code_path():
    """`bench.py --gpus 8 --host group` as an 8-GPU node runs it — eight engines, a host thread per rank, ncclCommInitAll over eight ranks, one
    ncclAllGather per rank and publish — on the ONE device there is: the recording stand-in of tests/nccl_stub performs the collective
    (MJH_RCCL_LIB, NCCL_STUB_COPY=1; MJH_GROUP_TRANSPORT=2 lets the group take the RCCL path although the device repeats)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_rccl_sequence import build_stub
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MJH_RCCL_LIB=build_stub(), NCCL_STUB_COPY="1", MJH_GROUP_TRANSPORT="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--host", "group", "--group-devices", "0,0,0,0,0,0,0,0", "--config", "c5",
                        "--envs-per-gpu", "512", "--steps", "60", "--warmup", "6"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    h = j["host"]
    assert j["n_gpus"] == 8 and h["kind"] == "group" and h["rccl"] is True and h["rccl_ranks"] == 8 and h["host_threads"] == 8
    assert [x["nenv"] for x in h["ranks"]] == [512] * 8 and [x["env0"] for x in h["ranks"]] == [512 * k for k in range(8)]
    assert h["all_gather"]["count"] >= 3 and h["all_gather"]["ms_mean"] > 0 and h["all_gather"]["bytes_per_rank"] == 512 * (1 + 12 + 9) * 4
    assert j["value"] > 1e6 and j["config"]["envs_total"] == 4096


def test_the_drivers_multi_gpu_invocation_runs_the_rccl_publish_path():
    """The driver launches N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N ...`, one rank per GPU.  With the one GPU there is, the same launcher form with N = 1 (process group of one rank
    over RCCL: init, barriers, the max-over-ranks reduction) and `--force-dist` (the 60 Hz publish as an all-gather on the
    communication stream beside the steps): the line carries the ranks host record and the collective's timing."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "30", "--warmup", "6", "--force-dist", "--no-cpu-baseline", "--no-extra-configs", "--no-second-window"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 30 and j["value"] > 1e6 and j["scaling"] == "weak"
    h = j["host"]
    assert h["kind"] == "ranks" and h["rccl_ranks"] == 1 and h["ranks"] == [{"rank": 0, "env0": 0, "nenv": 4096}]
    ag = h["all_gather"]
    assert ag["count"] == 10 and ag["publish_every_steps"] == 3 and ag["bytes_per_rank"] == 4096 * (1 + 28 + 24) * 4 and 0 < ag["ms_mean"] < 50
