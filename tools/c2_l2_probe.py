"""C2: is the operand stream of the block sweeps a co-limiter?  ("same kernel, operands read from a 1-env-sized
buffer that stays in L2: if the launch shortens by > 20 %, traffic binds")   python tools/c2_l2_probe.py [nenv] [settle]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mujoco_sim_amd as ms
from mujoco_sim_amd import capi
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 200
m = ms.scene("boxpile", 64); m.c.maxcon = 600; m.c.maxefc = 2400
e = ms.Engine(m, nenv)
e.load_tables(ms.boxes_randomize(m, 0, nenv, jitter=0.01))
e.step(settle); e.synchronize()
st = e.get_stats()
print(f"C2 {nenv} envs after {settle} steps: ncon {st[:,0].mean():.1f} nefc {st[:,1].mean():.1f} sweeps {st[:,2].mean():.1f}")
lib = capi.load()
for slices in (1, 8, 64, 512):
    msv = np.zeros(2); it = np.zeros(2)
    rc = lib.mjh_debug_solve_probe(e.h, slices, 3, capi.dptr(msv), capi.dptr(it))
    assert rc == 0, lib.mjh_last_error()
    print(f"solve launch over {nenv} envs: own operands {msv[0]:.3f} ms ({it[0]:.1f} sweeps) | operands of {slices} env slices {msv[1]:.3f} ms ({it[1]:.1f} sweeps) -> {100 * (1 - msv[1] / msv[0]):.1f} % shorter")
    e.step(2); e.synchronize()       # rebuild the pools
