# hardware queues x cohorts for the small-model configs (the cohort streams share GPU_MAX_HW_QUEUES = 4 hardware queues by default)
for c in c3 c5; do for q in 4 8; do for g in 2 3 4 6; do
GPU_MAX_HW_QUEUES=$q python bench.py --config $c --steps 340 --warmup 34 --no-cpu-baseline --no-second-window --cohorts $g > gpurun_out/q_${c}_${q}_$g.json 2>gpurun_out/q_${c}_${q}_$g.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/q_${c}_${q}_$g.json").read().strip().splitlines()[-1])
    print("$c queues $q cohorts $g:", round(d["value"]/1e6,2), "M  ms/step", round(d["ms_per_step"],4), "kernel_ms", round(d["roofline"]["kernel_ms"],4))
except Exception as ex: print("$c queues $q cohorts $g: failed", open("gpurun_out/q_${c}_${q}_$g.err").read()[-300:])
PY
done; done; done
