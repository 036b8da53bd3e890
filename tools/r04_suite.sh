#!/bin/bash
# round 4: the whole GPU suite, the driver's bench line, and the profile of every config (kernel trace + HBM PMC + SQ passes)
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
timeout 600 python bench.py > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"
python - <<PY
import json
r = json.loads(open("$OUT/bench_driver.json").read().strip().splitlines()[-1])
print("S24:", round(r["value"] / 1e6, 3), "M env-steps/s; roofline", r.get("roofline"))
for k, v in (r.get("configs") or {}).items():
    print("  ", k, round(v.get("value", 0) / 1e6, 3), "M", {a: v[a] for a in ("overflow_envs", "mean_ncon", "mean_nefc", "mean_solver_iter") if a in v})
print("  literal", r.get("literal_loop"))
PY
for c in ${CONFIGS:-s24}; do
  bash tools/profile_config.sh $c $TAG 100 1 > $OUT/profile_$c.log 2>&1
  tail -3 $OUT/profile_$c.log
done
