import sys, os, argparse, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
import mujoco_sim_amd as ms
ap = argparse.Namespace(envs_per_gpu=0, pack=0, maxcon=0, extra_steps=20, timing_stride=5, pen_half=0.0, cohorts=-1, steps_per_launch=0, no_gather=False)
stream = torch.cuda.current_stream().cuda_stream
r = bench.short_config_line(ms, ap, "c5", 0, stream); print("c5 alone (extras function):", round(r["value"] / 1e6, 2), "steps", r["steps"])
m = ms.scene("s24"); e = ms.Engine(m, 4096, stream=stream); e.load_s24(); e.set_cohorts(3); e.step(300); e.synchronize()
r = bench.short_config_line(ms, ap, "c5", 0, stream); print("c5 with an idle S24 engine alive:", round(r["value"] / 1e6, 2))
e.close()
r = bench.short_config_line(ms, ap, "c5", 0, stream); print("c5 after closing it:", round(r["value"] / 1e6, 2))
r = bench.short_config_line(ms, ap, "c4", 0, stream); print("c4:", round(r["value"] / 1e6, 3))
r = bench.short_config_line(ms, ap, "c5", 0, stream); print("c5 after c4:", round(r["value"] / 1e6, 2))
