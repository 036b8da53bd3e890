#!/bin/bash
# A/B library that differs from the current build in the dense solver's translation unit only: tools/dense_variant.sh <name> <extra hipcc flags...>
# -> build_exp/<name>/libmjhip.so (use with MJHIP_LIB)
set -e
N=$1; shift
R=/root/repo; B=$R/mujoco_sim_amd/build; O=$R/build_exp/$N
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-result "$@" -c $R/mujoco_sim_amd/csrc/dense.hip -o $O/dense.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/engine.o $B/window.o $O/dense.o $B/group.o $B/model_builder.o $B/scenes.o $B/host_sim.o $B/mjcf_loader.o -ldl -o $O/libmjhip.so
rm -f $O/dense.o
bash $R/tools/kernel_resources.sh build_exp/$N/libmjhip.so 2>/dev/null | grep -i "dense"
