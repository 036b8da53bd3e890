// valu_issue_bench.hip — issue-rate micro-benchmark for gfx950 ("does a wave64 fp32 VALU
// instruction issue in 2 or in 4 cycles?").  One 64-lane wave per workgroup; the dynamic LDS size pins the number of
// resident waves per SIMD (W = 1, 2, 3, 4); every wave runs REPS x UNROLL copies of one instruction pattern on
// independent (or, for the *_dep variants, one) register streams and stamps s_memtime around the loop.
//   build: hipcc --offload-arch=gfx950 -O3 tools/valu_issue_bench.hip -o tools/valu_issue_bench
//   run:   tools/valu_issue_bench            (prints one table; cycles are shader clocks from s_memtime)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { K_FMA = 0, K_FMA_DEP, K_PKFMA, K_ADD_DPP_QUAD, K_ADD_DPP_ROWROR, K_ADD_DPP_ROWMIRROR, K_PERMLANE16_SWAP, K_PERMLANE32_SWAP, K_CNDMASK, K_MED3,
       K_READLANE, K_DS_READ_B128, K_DS_READ_B32, K_RCP, K_MIX_SOLVER, K_ROWCHAIN, K_ROWCHAIN_NONOP, K_SNOP, K_SALU, K_COUNT };
static const char* kname[K_COUNT] = {"v_fma_f32 x8 independent", "v_fma_f32 dependent chain", "v_pk_fma_f32 x8 independent", "v_add_f32 dpp quad_perm x8", "v_add_f32 dpp row_ror:4 x8",
                                     "v_add_f32 dpp row_mirror x8", "v_permlane16_swap x4 pairs", "v_permlane32_swap x4 pairs", "v_cndmask_b32 x8", "v_med3_f32 x8",
                                     "v_readlane_b32 + s_add x8", "ds_read_b128 x8 (broadcast addr)", "ds_read_b32 x8 (lane addr)", "v_rcp_f32 x8", "mix: 6 fma + 2 dpp add + 1 ds_read_b128",
                                     "row chain: (v_max, s_nop 1, v_fmac dpp) x4", "row chain without the s_nop x4", "s_nop 1 x8", "s_add_u32 x8"};

template <int KIND>
__global__ __launch_bounds__(64) void bench(float* out, long long* ticks, int reps) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = (float)i * 1e-3f;
  __syncthreads();
  float a0 = lane * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float b = 1.0000001f, c = 1e-7f;
  const float2 pb = {b, b}, pc = {c, c};
  float4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0, q4 = q0, q5 = q0, q6 = q0, q7 = q0;
  int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  const unsigned la = (unsigned)(lane * 4), ba = 64;
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll 4
  for (int r = 0; r < reps; r++) {
    if (KIND == K_FMA)
      asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                   "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    else if (KIND == K_FMA_DEP)
      asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                   "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2"
                   : "+v"(a0) : "v"(b), "v"(c));
    else if (KIND == K_PKFMA)
      asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\t"
                   "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
    else if (KIND == K_ADD_DPP_QUAD)
      asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    else if (KIND == K_ADD_DPP_ROWROR)
      asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %5, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %6, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %7, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xf"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    else if (KIND == K_ADD_DPP_ROWMIRROR)
      asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %4, %4, %4 row_mirror row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %5, %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %6, %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %7, %7, %7 row_mirror row_mask:0xf bank_mask:0xf"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    else if (KIND == K_PERMLANE16_SWAP)
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\ts_nop 1"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    else if (KIND == K_PERMLANE32_SWAP)
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\ts_nop 1"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    else if (KIND == K_CNDMASK)
      asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t"
                   "v_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    else if (KIND == K_MED3)
      asm volatile("v_med3_f32 %0, %0, %8, %9\n\tv_med3_f32 %1, %1, %8, %9\n\tv_med3_f32 %2, %2, %8, %9\n\tv_med3_f32 %3, %3, %8, %9\n\t"
                   "v_med3_f32 %4, %4, %8, %9\n\tv_med3_f32 %5, %5, %8, %9\n\tv_med3_f32 %6, %6, %8, %9\n\tv_med3_f32 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    else if (KIND == K_READLANE)
      asm volatile("v_readlane_b32 %0, %4, 0\n\tv_readlane_b32 %1, %5, 32\n\tv_readlane_b32 %2, %6, 0\n\tv_readlane_b32 %3, %7, 32\n\t"
                   "s_add_i32 %0, %0, %1\n\ts_add_i32 %2, %2, %3\n\t"
                   "v_readlane_b32 %1, %4, 1\n\tv_readlane_b32 %3, %5, 33\n\ts_add_i32 %0, %0, %1\n\ts_add_i32 %2, %2, %3\n\t"
                   "v_readlane_b32 %1, %6, 1\n\tv_readlane_b32 %3, %7, 33\n\ts_add_i32 %0, %0, %1\n\ts_add_i32 %2, %2, %3"
                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    else if (KIND == K_DS_READ_B128)
      asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                   "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                   : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(ba) : "memory");
    else if (KIND == K_DS_READ_B32)
      asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:256\n\tds_read_b32 %2, %8 offset:512\n\tds_read_b32 %3, %8 offset:768\n\t"
                   "ds_read_b32 %4, %8 offset:1024\n\tds_read_b32 %5, %8 offset:1280\n\tds_read_b32 %6, %8 offset:1536\n\tds_read_b32 %7, %8 offset:1792\n\ts_waitcnt lgkmcnt(0)"
                   : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(la) : "memory");
    else if (KIND == K_RCP)
      asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\tv_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    else if (KIND == K_MIX_SOLVER)
      asm volatile("ds_read_b128 %8, %9\n\t"
                   "v_fma_f32 %0, %0, %10, %11\n\tv_fma_f32 %1, %1, %10, %11\n\tv_fma_f32 %2, %2, %10, %11\n\t"
                   "s_nop 1\n\tv_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_fma_f32 %4, %4, %10, %11\n\tv_fma_f32 %5, %5, %10, %11\n\tv_fma_f32 %6, %6, %10, %11\n\t"
                   "s_nop 1\n\tv_add_f32_dpp %7, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_waitcnt lgkmcnt(0)"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=v"(q0) : "v"(ba), "v"(b), "v"(c) : "memory");
    if (KIND == K_ROWCHAIN)     // the Gauss-Seidel row of patch_pgs.h: per-rep 4 rows (the dependent chain t -> d -> t)
      asm volatile("v_max_f32 %1, %0, %2\n\ts_nop 1\n\tv_fmac_f32_dpp %0, %1, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32 %1, %0, %2\n\ts_nop 1\n\tv_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32 %1, %0, %2\n\ts_nop 1\n\tv_fmac_f32_dpp %0, %1, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32 %1, %0, %2\n\ts_nop 1\n\tv_fmac_f32_dpp %0, %1, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                   : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
    if (KIND == K_ROWCHAIN_NONOP)     // (timing only: without the wait states the DPP read is not guaranteed to see the v_max)
      asm volatile("v_max_f32 %1, %0, %2\n\tv_fmac_f32_dpp %0, %1, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32 %1, %0, %2\n\tv_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32 %1, %0, %2\n\tv_fmac_f32_dpp %0, %1, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f32 %1, %0, %2\n\tv_fmac_f32_dpp %0, %1, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                   : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
    if (KIND == K_SNOP)
      asm volatile("s_nop 1\n\ts_nop 1\n\ts_nop 1\n\ts_nop 1\n\ts_nop 1\n\ts_nop 1\n\ts_nop 1\n\ts_nop 1");
    if (KIND == K_SALU)
      asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t"
                   "s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1"
                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  float acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + q0.x + q1.y + q2.z + q3.w + q4.x + q5.y + q6.z + q7.w + (float)(s0 + s1 + s2 + s3);
  if (acc == 123.456f) out[blockIdx.x] = acc;
  if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int KIND> static void run_one(int W, float* dout, long long* dticks, int reps, int ncu, double* cyc_per_instr_wave, double* ms_out) {
  const int nblk = ncu * 4 * W;
  // W waves per SIMD = 4 W workgroups per CU: dynamic LDS of 160 KiB / (4 W) (less a little) caps the residency at exactly that
  size_t lds = (160 * 1024) / (4 * W) - 512;
  lds = lds > 64 * 1024 ? 64 * 1024 : lds;                     // (a workgroup may hold at most 64 KiB; W = 1 is then capped by the grid size instead)
  CHK(hipFuncSetAttribute((const void*)bench<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  hipLaunchKernelGGL(bench<KIND>, dim3(nblk), dim3(64), lds, 0, dout, dticks, 16);    // warm-up
  CHK(hipGetLastError()); CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a));
  hipLaunchKernelGGL(bench<KIND>, dim3(nblk), dim3(64), lds, 0, dout, dticks, reps);
  CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
  float ms = 0; CHK(hipEventElapsedTime(&ms, a, b));
  std::vector<long long> t(nblk);
  CHK(hipMemcpy(t.data(), dticks, nblk * sizeof(long long), hipMemcpyDeviceToHost));
  double mean = 0; for (auto v : t) mean += (double)v; mean /= nblk;
  const int per_rep = (KIND == K_PERMLANE16_SWAP || KIND == K_PERMLANE32_SWAP || KIND == K_ROWCHAIN || KIND == K_ROWCHAIN_NONOP) ? 4 : (KIND == K_READLANE ? 14 : (KIND == K_MIX_SOLVER ? 9 : 8));
  *cyc_per_instr_wave = mean / ((double)reps * per_rep);
  *ms_out = ms;
  CHK(hipEventDestroy(a)); CHK(hipEventDestroy(b));
}

template <int KIND> static void run_kind(float* dout, long long* dticks, int ncu) {
  const int reps = 4000;
  printf("%-42s", kname[KIND]);
  for (int W = 1; W <= 4; W++) {
    double c, ms; run_one<KIND>(W, dout, dticks, reps, ncu, &c, &ms);
    // per-wave cycles per instruction, and the SIMD's aggregate rate: W waves issue W instructions per c cycles
    // wall-clock cross-check of the tick unit: every SIMD holds W waves, each issuing reps*per_rep instructions in `ms`
    printf("  W=%d: %6.2f ticks/instr/wave (SIMD %5.2f) %6.3f ms", W, c, c / W, ms);
  }
  printf("\n");
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz; one wave per workgroup, W resident waves per SIMD; ticks = s_memtime\n", p.gcnArchName, ncu, p.clockRate);
  float* dout; long long* dticks;
  CHK(hipMalloc((void**)&dout, (size_t)ncu * 16 * sizeof(float))); CHK(hipMalloc((void**)&dticks, (size_t)ncu * 16 * sizeof(long long)));
  run_kind<K_FMA>(dout, dticks, ncu); run_kind<K_FMA_DEP>(dout, dticks, ncu); run_kind<K_PKFMA>(dout, dticks, ncu);
  run_kind<K_ADD_DPP_QUAD>(dout, dticks, ncu); run_kind<K_ADD_DPP_ROWROR>(dout, dticks, ncu); run_kind<K_ADD_DPP_ROWMIRROR>(dout, dticks, ncu);
  run_kind<K_PERMLANE16_SWAP>(dout, dticks, ncu); run_kind<K_PERMLANE32_SWAP>(dout, dticks, ncu);
  run_kind<K_CNDMASK>(dout, dticks, ncu); run_kind<K_MED3>(dout, dticks, ncu);
  run_kind<K_DS_READ_B128>(dout, dticks, ncu); run_kind<K_DS_READ_B32>(dout, dticks, ncu); run_kind<K_RCP>(dout, dticks, ncu); run_kind<K_MIX_SOLVER>(dout, dticks, ncu);
  run_kind<K_ROWCHAIN>(dout, dticks, ncu); run_kind<K_ROWCHAIN_NONOP>(dout, dticks, ncu); run_kind<K_SNOP>(dout, dticks, ncu); run_kind<K_SALU>(dout, dticks, ncu);   // (row chain: ticks per ROW)
  // (the v_readlane + s_add variant does not terminate on this box and is left out)
  return 0;
}
