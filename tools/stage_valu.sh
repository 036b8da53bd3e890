#!/bin/bash
# Per-stage instruction counts of the step kernel: launches that stop at successive stage boundaries, under
# rocprofv3 --pmc; a stage's cost is the difference of consecutive rows.   tools/stage_valu.sh  (on the GPU box)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/stage_valu
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/stage_valu_run.py <<PY
import sys; sys.path.insert(0, "$ROOT")
import mujoco_sim_amd as ms
from mujoco_sim_amd import capi
m = ms.scene("s24"); e = ms.Engine(m, 4096); e.set_cohorts(1); e.load_s24(); e.step(400); e.synchronize()
L = capi.load()
for k in range(1, 15):
    for r in range(2):
        assert L.mjh_debug_stop_at(e.h, k, 0) == 0
e.synchronize()
PY
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/p -o p -- python /tmp/stage_valu_run.py > $OUT/run.log 2>&1
python - <<PY
import glob, sqlite3
f = glob.glob("$OUT/p/*.db"); con = sqlite3.connect(f[0])
names = ["", "load state", "FK + geoms", "COM/cdof/CRBA", "factor", "collision", "row headers", "J rows + params", "B, schedule",
         "vel stage (RNE, aref)", "controller/inverse", "smooth acc", "warmstart + A_c + AR", "PGS sweeps", "checkAcc + integrate"]
vals = {}
for cn in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES"):
    rows = [r[0] for r in con.execute("select value from counters_collection where kernel_name like '%mjh_step_kernel%' and counter_name=? order by start", (cn,))]
    vals[cn] = rows[-28:]
print("%-26s %10s %10s %10s %12s   (per env-step; stage = difference to the previous stop)" % ("stage", "VALU", "SALU", "LDS", "wave cyc x4"))
prev = {k: 0.0 for k in vals}
for k in range(1, 15):
    cur = {cn: (vals[cn][2*(k-1)] + vals[cn][2*(k-1)+1]) / 2 / 4096 for cn in vals}
    print("%2d %-23s %10.0f %10.0f %10.0f %12.0f" % (k, names[k], *[cur[cn] - prev[cn] for cn in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES")]))
    prev = cur
PY
