// window.hip — translation unit of mjh_window_kernel (csrc/window_kernel.h) and its launcher.  gfx950 only.
#include <hip/hip_runtime.h>

#include "../../include/mjhip.h"
#define MJH_WINDOW_TU 1        // (step_kernel.h: the helper kernels that are not templates live in engine.hip's unit only)
#include "window_kernel.h"

// nvt: dof slots of the instance (24 / 32); grid: wavefronts (the 32-row section's first); lds: bytes of the LDS tier
hipError_t mjh_launch_window(hipStream_t st, int nvt, int grid, size_t lds, const DConst* dC, const DState& S, int env0, int n, int nl, int wxf, int n32, int n64) {
  if (nvt == 24) {
    // launches without the LDS tier keep the pairs' cross tiles in LDS (instance XL: 12 KB per wavefront, 48 registers' worth of copies per sweep fewer)
    if (nl == 0 && WN_XLDS_BYTES(24, WN_NW24) > 0) hipLaunchKernelGGL((mjh_window_kernel<24, WN_NW24, true>), dim3(grid), dim3(64), lds + WN_XLDS_BYTES(24, WN_NW24), st, dC, S, env0, n, nl, wxf, n32, n64);
    else hipLaunchKernelGGL((mjh_window_kernel<24, WN_NW24, false>), dim3(grid), dim3(64), lds, st, dC, S, env0, n, nl, wxf, n32, n64);
  }
  else hipLaunchKernelGGL((mjh_window_kernel<32, WN_NW32>), dim3(grid), dim3(64), lds, st, dC, S, env0, n, nl, wxf, 0, 0);
  return hipGetLastError();
}
