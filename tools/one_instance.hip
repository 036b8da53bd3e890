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
#define ONE_WPRE 0
#endif
template __global__ void mjh_step_kernel<ONE_NROW, ONE_DIAGM, ONE_EXTRA, ONE_WPRE>(const DConst*, const DState, int, int, int, int);
#ifdef ONE_NW
template __global__ void mjh_window_kernel<24, ONE_NW>(const DConst*, const DState, int, int, int, int, int);
#endif
#ifdef ONE_SOLVE
template __global__ void mjh_solve_kernel<true, false>(const DConst*, const DState, int);
template __global__ void mjh_solve_kernel<false, false>(const DConst*, const DState, int);
#endif
#ifdef ONE_DENSE_K          // the dense sweep kernel for ONE rows-per-lane count (registers of each instance of dn_solve_env)
#include "../mujoco_sim_amd/csrc/dense_pgs.h"
__global__ __launch_bounds__(64) void one_dense_solve(const DConst* __restrict__ C, const DState S, int env0) {
  const DModel& M = C->M; const Lay& L = C->L;
  extern __shared__ float lds[];
  float* s_df = lds; float* s_x = s_df + DN_CAP_MAX; float* s_qld = s_x + 128; int* s_anc = (int*)(s_qld + M.nM);
  const int lane = threadIdx.x;
  float* const gs = S.gscratch + (size_t)(env0 + blockIdx.x) * (size_t)S.gstride;
  int* meta = (int*)(gs + L.g_meta);
  const int nefc = __builtin_amdgcn_readfirstlane(meta[2]);
  const int nr32 = (nefc + 31) & ~31;
  const DenseOff o = dense_off(M, L);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)gs, 0, 0x7ffffff0, 0x00020000);
  dn_solve_env<ONE_DENSE_K>(M, L, gs, rs, o, nefc, nr32, s_df, s_x, s_qld, s_anc, lane, meta);
}
#endif
