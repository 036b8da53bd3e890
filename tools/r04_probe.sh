#!/bin/bash
# C2 operand-stream probe, cohort sweeps of C2 / C4, the driver's line
set -u
TAG=${1:-r04f}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 600 python tools/c2_l2_probe.py 4096 200 > $OUT/c2_l2_probe.txt 2>&1; cat $OUT/c2_l2_probe.txt | tail -8
run() {
  local label=$1 cfg=$2; shift 2
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 400 python bench.py --config $cfg --no-extra-configs --no-cpu-baseline --no-second-window --steps 60 --warmup 10 "$@" > $OUT/b_$label.json 2> $OUT/b_$label.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_$label.json").read().strip().splitlines()[-1])
    print("$label:", round(r["value"] / 1e6, 4), "M  ms/step", round(r["ms_per_step"], 4), "sweeps", round(r["config"]["mean_solver_iter"], 1), "nefc", round(r["config"]["mean_nefc"], 1), "cohorts", r["config"]["cohorts"], "kernel_ms", round(r["roofline"]["kernel_ms"], 4))
except Exception as ex:
    print("$label: FAILED", ex); print(open("$OUT/b_$label.err").read()[-800:])
PY
}
for c in 2 3 4 6; do run c2_c$c c2 A=1 -- --cohorts $c; done
run c2_w2 c2 MJH_SOLVE_WAVES=2 -- --cohorts 3
for c in 2 3 4 6; do run c4_c$c c4 A=1 -- --cohorts $c; done
timeout 600 python bench.py > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"
python - <<PY
import json
r = json.loads(open("$OUT/bench_driver.json").read().strip().splitlines()[-1])
print("S24:", round(r["value"] / 1e6, 3), "M env-steps/s;", {k: r["roofline"][k] for k in ("frac", "valu_issue_frac", "valu_lane_util", "kernel_ms", "traffic")})
for k, v in (r.get("configs") or {}).items():
    print("  ", k, round(v.get("value", 0) / 1e6, 3), "M", {a: v[a] for a in ("overflow_envs", "steps_per_launch", "mean_solver_iter") if a in v})
print("  literal", (r.get("literal_loop") or {}).get("value"), "cpu", r.get("cpu_baseline", {}).get("value"))
PY
