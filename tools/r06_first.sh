#!/bin/bash
# round 6, first GPU session: the new tests, the driver's bench line with every extra in a process of its own, S24D / C2 stand-alone beside it
set -u
O=gpurun_out/r06a; mkdir -p $O
( time python -m pytest tests/test_gpu_round6.py tests/test_gpu_bench_line.py "tests/test_gpu_round5.py::test_launch_chain_as_a_captured_graph_equals_the_separate_launches" -x -q -m gpu -s ) > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; grep -E "WINDOW-CLAMP|passed|failed|Error|error" $O/pytest_new.log | head -20
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; tail -3 $O/bench_driver.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06a/bench_driver.json").read().strip().splitlines()[-1])
print("value", d["value"], "with_inverse", d.get("value_with_inverse"), "literal", d["literal_loop"]["value"], "v30", d.get("value_30_contact"))
for k,v in d["configs"].items():
    print(k, {x: (round(v[x],4) if isinstance(v[x],float) else v[x]) for x in ("value","steps","ms_per_step","kernel_ms","mean_ncon","mean_nefc","overflow_envs","process_wall_s") if x in v} if "error" not in v else v)
print("c30", d.get("config_30_contact"))
print("c2 hist", d["configs"]["c2"].get("ncon_histogram"))
PY
for c in s24d c2 c4; do
  python bench.py --config $c --steps $( [ $c = c2 ] && echo 500 || echo 200 ) --warmup 20 --no-cpu-baseline --no-second-window --no-extra-configs 2>/dev/null > $O/standalone_$c.json
  python -c "import json; d=json.loads(open('$O/standalone_$c.json').read().strip().splitlines()[-1]); print('standalone $c', round(d['value']/1e6,4), 'M ms/step', round(d['ms_per_step'],4), 'chain', round(d['roofline']['kernel_ms'],4))"
done
