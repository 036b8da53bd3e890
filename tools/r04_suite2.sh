#!/bin/bash
# whole GPU suite at the step-loop build, group host issue time, profiles of C2 / C3 / C4 / C5
set -u
TAG=${1:-r04e}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 $OUT/pytest_gpu.log | cut -c1-400
for a in "c5_group8 --host group --gpus 8 --group-devices 0,0,0,0,0,0,0,0 --envs-per-gpu 512" "c5_group8_4096 --host group --gpus 8 --group-devices 0,0,0,0,0,0,0,0 --envs-per-gpu 4096" "c5_group1 --host group --gpus 1 --group-devices 0"; do
  set -- $a; label=$1; shift
  timeout 300 python bench.py --config c5 --no-extra-configs --no-cpu-baseline --no-second-window --steps 300 "$@" > $OUT/b_$label.json 2> $OUT/b_$label.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_$label.json").read().strip().splitlines()[-1])
    print("$label:", round(r["value"] / 1e6, 3), "M  ms/step", round(r["ms_per_step"], 4), r["host"]["transport"][:12], r["host"]["host_issue"], r["host"]["all_gather"])
except Exception as ex:
    print("$label: FAILED", ex); print(open("$OUT/b_$label.err").read()[-800:])
PY
done
for c in ${CONFIGS:-c2 c4 c3 c5}; do
  bash tools/profile_config.sh $c $TAG 60 1 > $OUT/profile_$c.log 2>&1
  grep -E "Summed kernel time|launches, mean|per env-step: \*\*|active lanes" $OUT/profile_$c.log | cut -c1-300
done
