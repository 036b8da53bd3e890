# C4: what the dense sweep waits for.  A/B library built with -DDN_PROBE_SAMEGROUP (every column fetch of mjh_dense_solve_kernel aimed at
# the first 16-row group: served from a near cache; the results are garbage and every env sweeps to the cap):
#   MJH_EXTRA_FLAGS=-DDN_PROBE_SAMEGROUP MJH_BUILD_DIR=$PWD/mujoco_sim_amd/build_exp python -c "import mujoco_sim_amd.build as b; b.build(force=True)"
#   gpurun -- bash tools/r04_dense_probe.sh
cd /root/repo
for lib in default exp; do
if [ $lib = default ]; then unset MJHIP_LIB; else export MJHIP_LIB=/root/repo/mujoco_sim_amd/build_exp/libmjhip.so; fi
[ $lib = exp ] && [ ! -f "$MJHIP_LIB" ] && continue
python bench.py --config c4 --steps 100 --warmup 20 --no-cpu-baseline --no-second-window --no-extra-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib c4', round(d['value']/1e6,3), 'M  ms/step', round(d['ms_per_step'],4), 'mean sweeps', round(d['config']['mean_solver_iter'],2))"
done
