# envs per wavefront (mjh_model_replicate) x cohorts for the small-model configs, with the in-kernel step loop
for c in c3 c5; do for p in 1 2 4 8; do for g in 2 3; do
python bench.py --config $c --steps 340 --warmup 34 --no-cpu-baseline --no-second-window --pack $p --cohorts $g > gpurun_out/pack_${c}_${p}_$g.json 2>gpurun_out/pack_${c}_${p}_$g.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/pack_${c}_${p}_$g.json").read().strip().splitlines()[-1])
    print("$c pack $p cohorts $g:", round(d["value"]/1e6,2), "M  ms/step", round(d["ms_per_step"],4), "kernel_ms", round(d["roofline"]["kernel_ms"],4), "epw", d["config"]["envs_per_wavefront"], "lds", d["config"]["lds_bytes_per_env"])
except Exception as ex: print("$c pack $p cohorts $g: failed", open("gpurun_out/pack_${c}_${p}_$g.err").read()[-300:])
PY
done; done; done
