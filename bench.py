#!/usr/bin/env python
"""bench.py — env-steps/sec of the S24 scene (BASELINE.json metric) on N MI355X.

A "step" is one pass of the hot path — mj_step1 + controller + mj_step2
(reference loop body, src/mj_main.cpp:82-112) — over one batch of 4096 environments
per GPU, ONE kernel launch per step (the drop-in keeps the reference's per-step host
hand-off to ros_control).  Inputs are resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W

N > 1: launched by torch.distributed.run, one rank per GPU; environments are sharded
(rank r owns envs [r*4096, (r+1)*4096)), no data-path collective; the only collective is the
RCCL all-gather of the published state slice at 60 Hz of simulated time.  The slice is packed
(mjh_export_state_device) at that rate at every N, so that per-GPU work does not depend on N.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The engine steps two env cohorts on their own HIP streams; with RCCL's streams on top, more than the default 4
# hardware queues are in use and streams that share a queue serialise (measured: 4.19 M vs 4.73 M env-steps/s on the
# publish path).  Must be set before the HIP runtime initialises, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ENVS_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (~6.3 TB/s achievable)


def algorithmic_bytes_per_env_step(nq, nv):
    """read qpos,qvel,warmstart,cmd(ddq,dq); write qpos,qvel,warmstart; fp32."""
    return 4 * (2 * nq + 6 * nv)


def cpu_baseline(model, eng, tab, env_offset, sample_envs, sample_steps, with_inverse=0):
    """Oracle (fp64 C restatement, test infrastructure) timed on the host cores on a bounded
    sample of the SAME workload: the first `sample_envs` envs, started from the GPU's settled state."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from mujoco_sim_amd.engine import EP

    L = orc.lib()
    t, q, v, w = eng.get_state(0, sample_envs)
    ds = []
    for i in range(sample_envs):
        d = orc.OrcData(model.ptr)
        for k, wh in EP.items():
            d.set_env_param(wh, tab[k][i])
        d.set_qpos(q[i]); d.f("qvel")[:] = v[i]; d.f("qacc_warmstart")[:] = w[i]
        ds.append(d)
    arr = (C.c_void_p * sample_envs)(*[d.d for d in ds])
    ncores = os.cpu_count() or 1
    out = {}
    for label, threads in (("mt", ncores), ("st", 1)):
        n_envs = sample_envs if threads > 1 else max(1, min(sample_envs, 8))
        L.orc_set_threads(threads)
        t0 = time.perf_counter()
        L.orc_step_many(arr, n_envs, sample_steps, with_inverse)
        dt = time.perf_counter() - t0
        out[label] = n_envs * sample_steps / dt
    rows = float(np.mean([d.i("nefc") for d in ds]))
    return {
        "value": out["mt"], "unit": "env-steps/s", "cores": ncores, "kind": "port",
        "value_1thread": out["st"],
        "sample": f"oracle/ fp64 C restatement (PGS, dense AR), first {sample_envs} S24 envs from the GPU's settled state, "
                  f"{sample_steps} steps, OpenMP over envs; mean nefc {rows:.0f}; reference library (libmujoco 2.3.7) absent",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--with-inverse", type=int, default=0, help="also run mj_inverse every step (MjHWInterface::read)")
    ap.add_argument("--fuse", type=int, default=1, help="steps between host hand-offs (one kernel launch per step either way)")
    ap.add_argument("--cohorts", type=int, default=-1, help="env cohorts stepped on separate HIP streams (-1: engine default)")
    ap.add_argument("--maxcon", type=int, default=0, help="override the scene's contact capacity per env (rows: 6 per contact); 0 = scene default (40)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and run the publish all-gather even with one rank (self-test of the multi-GPU path)")
    ap.add_argument("--cpu-envs", type=int, default=1024)
    ap.add_argument("--cpu-steps", type=int, default=100)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import mujoco_sim_amd as ms

    model = ms.scene("s24")
    if args.maxcon > 0:
        model.c.maxcon = args.maxcon; model.c.maxefc = 6 * args.maxcon
    nenv = args.envs_per_gpu
    stream = torch.cuda.current_stream()
    eng = ms.Engine(model, nenv, device=local_rank, stream=stream.cuda_stream)
    tab = eng.load_s24(env_offset=rank * nenv)
    if args.cohorts > 0:
        eng.set_cohorts(args.cohorts)
    stride = eng.state_stride
    pub = torch.empty(nenv * stride, dtype=torch.float32, device="cuda")
    gathered = torch.empty(world * nenv * stride, dtype=torch.float32, device="cuda") if use_dist else None
    publish_every = max(1, int(round(1.0 / (60.0 * model.opt.timestep))))  # 60 Hz of simulated time

    def run(nsteps):
        s = 0
        while s < nsteps:
            k = min(args.fuse, nsteps - s)
            eng.step(k, args.with_inverse)
            s += k
            if not args.no_gather and (s % publish_every) < k:   # 60 Hz publish: packed state slice (+ RCCL all-gather)
                eng.export_state_device(pub.data_ptr())
                if use_dist:
                    dist.all_gather_into_tensor(gathered, pub)

    run(args.warmup)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # the engine brackets every step-kernel launch with HIP events on the stream it is launched on (cohort streams)
    eng.set_launch_timing(True)
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms, n_launches = eng.get_launch_timing()
    eng.set_launch_timing(False)
    cohorts = eng.cohorts

    st = eng.get_stats()
    total_envs = nenv * world
    value = total_envs * args.steps / elapsed
    bytes_step = algorithmic_bytes_per_env_step(model.nq, model.nv)
    envs_per_launch = nenv * args.steps / max(n_launches, 1)   # one launch = one step of one cohort
    achieved = bytes_step * envs_per_launch / (kernel_ms * 1e-3) / 1e9
    traffic = None
    valu_busy = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))   # rocprofv3 PMC measurement of the committed profile run, per launch
            traffic = tj.get("bytes_per_launch")
            valu_busy = tj.get("valu_issue_busy")
            if traffic is not None and tj.get("envs_per_launch"):
                traffic = traffic * envs_per_launch / tj["envs_per_launch"]
        except Exception:
            traffic = None
    out = {
        "metric": "env-steps/sec (whole node) at 4096 envs, 24-DoF/30-contact scene; 1/2/4/8 GPUs",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "S24: 4 free boxes (24 DoF) in a walled pen on the empty.xml floor, PGS 100 it / tol 1e-8",
                   "envs_per_gpu": nenv, "envs_total": total_envs, "steps_per_launch": 1, "steps_per_call": args.fuse, "cohorts": cohorts,
                   "with_inverse": bool(args.with_inverse), "parallelism": f"env-sharded x{world}",
                   "mean_ncon": float(st[:, 0].mean()), "max_ncon": int(st[:, 0].max()), "mean_nefc": float(st[:, 1].mean()),
                   "max_nefc": int(st[:, 1].max()), "mean_solver_iter": float(st[:, 2].mean()),
                   "overflow_envs": int((st[:, 3] & 3 != 0).sum()), "reset_envs": int((st[:, 3] & 4 != 0).sum()),
                   "lds_bytes_per_env": eng.lds_bytes, "contact_capacity": int(model.maxcon)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel": "mjh_step_kernel", "kernel_ms": kernel_ms,
                     "launches": n_launches, "envs_per_launch": envs_per_launch, "concurrent_launches": cohorts,
                     "algorithmic_bytes_per_env_step": bytes_step,
                     "valu_issue_busy": valu_busy,   # SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x resident waves per SIMD (profiles/): the binding resource
                     "note": "fused per-env pipeline keeps intermediates in LDS: the path is VALU-issue bound, far below the HBM roofline by design (DESIGN.md)"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the CPU baseline is timed at N = 1 only
        out["cpu_baseline"] = cpu_baseline(model, eng, tab, 0, min(args.cpu_envs, nenv), args.cpu_steps, args.with_inverse)
    if use_dist:
        dist.barrier()
    eng.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        try:  # RCCL prints its banner through C stdio: flush it first so that the JSON is the LAST line of stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
