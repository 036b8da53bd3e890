// dense.hip — translation unit of the dense row-space solver's kernels (csrc/dense_pgs.h: mjh_dense_build_kernel, mjh_dense_solve_kernel)
// and their launchers.  A unit of its own so that work on the sweeps rebuilds in seconds instead of with every mjh_step_kernel instance.
// gfx950 only.
#include <hip/hip_runtime.h>

#include "../../include/mjhip.h"
#define MJH_WINDOW_TU 1        // (step_kernel.h: the helper kernels that are not templates live in engine.hip's unit only)
#include "dense_pgs.h"

// the dynamic-LDS ceilings of the two kernels (a launch beyond 64 KB needs the attribute)
hipError_t mjh_dense_attributes(size_t build_lds, size_t solve_lds) {
  hipError_t rc = hipFuncSetAttribute((const void*)mjh_dense_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)build_lds);
  if (rc == hipSuccess) rc = hipFuncSetAttribute((const void*)mjh_dense_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_lds);
  return rc;
}
// build + sweeps for the n envs of a launch range (every env of the range whose assemble launch chose the dense form: meta[7])
hipError_t mjh_launch_dense(hipStream_t st, int n, size_t build_lds, size_t solve_lds, const DConst* dC, const DState& S, int env0) {
  hipLaunchKernelGGL(mjh_dense_build_kernel, dim3(n), dim3(DN_BUILD_THREADS), build_lds, st, dC, S, env0);
  hipLaunchKernelGGL(mjh_dense_solve_kernel, dim3(n), dim3(64), solve_lds, st, dC, S, env0);
  return hipGetLastError();
}
