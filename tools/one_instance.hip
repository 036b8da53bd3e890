// single-instance compile of the step kernel (register / scratch experiments): tools/one_instance.sh [-DONE_NROW=8 -DONE_DIAGM=true ...]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../include/mjhip.h"
#include "../mujoco_sim_amd/csrc/step_kernel.h"
#ifndef ONE_NROW
#define ONE_NROW 1
#endif
#ifndef ONE_DIAGM
#define ONE_DIAGM false
#endif
#ifndef ONE_EXTRA
#define ONE_EXTRA false
#endif
#ifndef ONE_WPRE
#define ONE_WPRE false
#endif
template __global__ void mjh_step_kernel<ONE_NROW, ONE_DIAGM, ONE_EXTRA, ONE_WPRE>(const DConst*, const DState, int, int, int, int);
#ifdef ONE_NW
template __global__ void mjh_window_kernel<24, ONE_NW>(const DConst*, const DState, int, int, int, int, int);
#endif
#ifdef ONE_SOLVE
template __global__ void mjh_solve_kernel<true, false>(const DConst*, const DState, int);
template __global__ void mjh_solve_kernel<false, false>(const DConst*, const DState, int);
#endif
