#!/bin/bash
# round 6: the whole GPU suite, smoke(), the driver's bench line (every extra a process of its own), and the profile of every config over the
# window the driver's line times it on (kernel trace + HBM PMC + SQ passes)      usage: tools/r06_suite.sh <tag> ; CONFIGS="s24:100 s24d:200 ..."
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  ( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -6 $OUT/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
fi
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"; tail -4 $OUT/bench_driver.err
python - <<PY
import json
r = json.loads(open("$OUT/bench_driver.json").read().strip().splitlines()[-1])
print("S24:", round(r["value"] / 1e6, 3), "M env-steps/s; with inverse", round(r.get("value_with_inverse", 0) / 1e6, 3), "; 30-contact", round(r.get("value_30_contact", 0) / 1e6, 3), "; literal", round(r["literal_loop"]["value"] / 1e6, 3))
print("  roofline", {k: r["roofline"].get(k) for k in ("frac", "achieved", "kernel_ms", "traffic", "valu_issue_frac", "valu_lane_util")})
for k, v in (r.get("configs") or {}).items():
    print("  ", k, round(v.get("value", 0) / 1e6, 4), "M", {a: v[a] for a in ("steps", "overflow_envs", "mean_ncon", "mean_nefc", "mean_solver_iter", "kernel_ms") if a in v}, v.get("error", ""))
print("  cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"], r["cpu_baseline"]["value_1thread"])
PY
for cs in ${CONFIGS:-s24:100}; do
  c=${cs%%:*}; s=${cs##*:}
  bash tools/profile_config.sh $c $TAG $s 1 > $OUT/profile_$c.log 2>&1
  tail -3 $OUT/profile_$c.log
done
