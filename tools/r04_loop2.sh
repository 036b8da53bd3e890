#!/bin/bash
# step loop at the final register allocation: bitwise test, group tests, C3 / C5 lines, the host issue time of the group host
set -u
TAG=${1:-r04d}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -s -k "loop or eight_shards" > $OUT/pytest_loop.log 2>&1; echo "pytest loop rc=$?"
grep -E "passed|failed|Error|assert|differ" $OUT/pytest_loop.log | cut -c1-600 | tail -8
run() {
  local label=$1 cfg=$2; shift 2
  timeout 300 python bench.py --config $cfg --no-extra-configs --no-cpu-baseline --no-second-window --steps 300 "$@" > $OUT/b_$label.json 2> $OUT/b_$label.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/b_$label.json").read().strip().splitlines()[-1])
    print("$label:", round(r["value"] / 1e6, 3), "M  ms/step", round(r["ms_per_step"], 4), "spl", r["config"]["steps_per_launch"], "cohorts", r["config"]["cohorts"], "kernel_ms", round(r["roofline"]["kernel_ms"], 4), (r.get("host") or {}).get("host_issue", ""))
except Exception as ex:
    print("$label: FAILED", ex); print(open("$OUT/b_$label.err").read()[-800:])
PY
}
for cfg in c3 c5; do
  for spl in 1 3; do run ${cfg}_spl$spl $cfg --steps-per-launch $spl; done
  run ${cfg}_nogather_spl8 $cfg --steps-per-launch 8 --no-gather
  run ${cfg}_spl3_c2 $cfg --steps-per-launch 3 --cohorts 2
  run ${cfg}_spl3_c4 $cfg --steps-per-launch 3 --cohorts 4
done
run c5_group8_spl3 c5 --host group --gpus 8 --group-devices 0,0,0,0,0,0,0,0 --envs-per-gpu 512
run c5_group8_spl1 c5 --host group --gpus 8 --group-devices 0,0,0,0,0,0,0,0 --envs-per-gpu 512 --steps-per-launch 1
run c5_group1 c5 --host group --gpus 1 --group-devices 0
