set -u
mkdir -p gpurun_out/r04fin5
( time python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r04fin5/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04fin5/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04fin5/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r04fin5/smoke.log
( time python bench.py ) > gpurun_out/r04fin5/bench_driver.json 2> gpurun_out/r04fin5/bench_driver.err; echo "bench rc=$?"; tail -4 gpurun_out/r04fin5/bench_driver.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04fin5/bench_driver.json").read().strip().splitlines()[-1])
print("value", d["value"], "with_inverse", d.get("value_with_inverse"), "literal", d["literal_loop"]["value"], d["literal_loop"]["fused_step_with_per_step_read_write"]["value"])
print({k: round(v["value"]/1e6,3) for k,v in d["configs"].items()})
print("roofline", {k:d["roofline"].get(k) for k in ["frac","achieved","kernel_ms","valu_issue_frac","valu_lane_util","traffic"]}, "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
