for c in c5 c3; do for s in 8 12 17 34; do
python bench.py --config $c --steps 340 --warmup 34 --no-cpu-baseline --no-second-window --steps-per-launch $s > gpurun_out/spl_${c}_$s.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/spl_${c}_$s.json").read().strip().splitlines()[-1])
print("$c spl $s:", round(d["value"]/1e6,2), "M  ms/step", round(d["ms_per_step"],4), "kernel_ms", round(d["roofline"]["kernel_ms"],4), "spl", d["config"]["steps_per_launch"])
PY
done; done
