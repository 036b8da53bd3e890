"""sha256 of the state after N steps of a bench config (A/B of two library builds with MJHIP_LIB: same arithmetic -> same bits):
python tools/state_hash.py [s24|s24d] [nenv] [steps] [nostats]     (nostats: state only — a probe build keeps clocks in the statistics)"""
import sys, os, types, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mujoco_sim_amd as ms
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "s24"
nenv = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 450
args = types.SimpleNamespace(envs_per_gpu=nenv, pack=0, maxcon=0, pen_half=0.0)
w = bench.WORKLOADS[name](ms, args, 0, 0, None)
e = w.eng
e.step(steps); e.synchronize()
t, q, v, a = e.get_state(); st = e.get_stats()
print("STATEHASH", name, nenv, steps, hashlib.sha256(q.tobytes() + v.tobytes() + a.tobytes() + (b"" if "nostats" in sys.argv else st[:, :3].tobytes())).hexdigest()[:16], "rows max", int(st[:, 1].max()), "flags", int((st[:, 3] & 7).max()))
