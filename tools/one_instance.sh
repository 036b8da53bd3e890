#!/bin/bash
# usage: tools/one_instance.sh [extra hipcc flags]  -> registers / scratch of the one instance
cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-result "$@" -c tools/one_instance.hip -o mujoco_sim_amd/build/one_tmp.o || exit 1
bash tools/kernel_resources.sh mujoco_sim_amd/build/one_tmp.o 2>&1 | awk '{print $1, "vgpr", $3, "sgpr", $7, "scratch", $9, "spill", $11}'
rm -f mujoco_sim_amd/build/one_tmp.o
