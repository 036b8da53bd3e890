// dense_pgs.h — the dense row-space solver of the many-body layout (articulated robots: BASELINE config C4, PR2 with its object
// pool, nv 97, 100-170 constraint rows).  The projection stage mj_projectConstraint (reached through
// mj_step1, /root/reference/src/mj_main.cpp:83) is the ONE dense contraction of the step:  AR = J M^-1 J^T + diag(R)  over the
// pyramid rows of an environment.  The block solver (step_kernel.h: pgs_many_body) never forms it: it walks the blocks one at a
// time on a whole wave — gather the block's dofs, a full-wave reduction per base row, uniform row math, scatter — ~600 clocks of
// dependent latency per BLOCK, and the environment that sits at the 100-sweep cap sets the length of the launch (C4: 1.75 ms of a
// 2.7 ms step, at 2.7 waves per CU).  Here the contraction is done once per step on the matrix cores and the sweeps become
// column updates of a residual vector that lives in registers:
//
//   (assemble launch)        with M = L^T D L (mj_factorM) the contraction is a SYRK:  J M^-1 J^T = Y D^-1 Y^T,  Y = J L^-1.  An env
//                            that takes the dense solver therefore stores Y where the block solver wants B = J M^-1: the half
//                            solve x <- L^-T x keeps a row's sparsity (only the ancestors of its dofs), ~7x fewer updates on PR2.
//   mjh_dense_build_kernel   one 512-thread workgroup per environment: expands the block rows (compact over <= 2 kinematic trees,
//                            pyramid rows n +- k folded) into the dense Y [rows x nv], multiplies  (Y D^-1) Y^T  with
//                            v_mfma_f32_16x16x4_f32 (16 x 16 tiles, fragments straight from L1 / L2, the A side scaled by 1 / D
//                            in registers; exact fp32), and writes AR' = -AR_pq / AR_qq (column-scaled, diagonal 0) row by row in
//                            Gauss-Seidel VISITING order, plus the per-row start values  t_q = -res_q / AR_qq,  f, lo, hi, AR_qq
//                            (res needs J a0: the base-row products of the assemble launch's warm start).
//   mjh_dense_solve_kernel   one wave per environment, lane q owns rows q, q + 64, ...:   per visited row p
//                               delta = med3(s_p, lo_p, hi_p) - f_p ;  f_p += delta ;  s_q += AR'_pq delta  for every q != p   (s = f + t)
//                            = projected Gauss-Seidel on the dual exactly as mj_solPGS iterates it (same rows, same order as the
//                            block solver and the oracle), ~9 instructions per ROW; then qacc = a0 + L^-1 D^-1 Y^T (f - f0)
//                            (L^-1 level by level: every dof of one tree depth in parallel).
//
// Environments whose row count exceeds the dense capacity keep the block solver (meta[7] says which one ran).  The dense form pays
// ~0.1 ms of build latency per env and wins ~3x per sweep, so it pays when some env of the launch sweeps long — such an env sets
// the length of the solve launch (C4: 1.28 M env-steps/s against 0.82 M) — and loses on robots that converge in a dozen sweeps
// (PR2 on the floor, all envs alike: 1.95 M with it, 2.36 M without).  The HOST therefore decides per cohort (engine.hip: mjh_step):
// whenever mjh_order_kernel rebuilds a cohort's launch order it also leaves "an env of this cohort swept >= M.dense_min_iter (32)
// times" in host-mapped memory; the host adopts the word of two rebuilds ago (after that kernel's event: no stall, and the choice
// depends on the step count only, so runs stay reproducible), queues the build and dense-solve launches only then, and tells the
// assemble launch (XF_DENSE).  Engines without a launch order (< 1024 envs) always take the dense form.
#pragma once
#include "step_kernel.h"

#ifndef DN_CAP_MAX
#define DN_CAP_MAX 256          // rows: four per lane
#endif
#define DN_META_DENSE 7         // meta[7] = 1: this step of this env is solved by the dense kernels

// float offsets of the dense region inside the env's scratch slice (L.g_dense .. ), CAP = M.dense_cap rows, NVS = M.dense_nvs
struct DenseOff { int art, yd, rf, rlo, rhi, rarr, rt, rmap; };
DEV DenseOff dense_off(const DModel& M, const Lay& L) {
  DenseOff o; const int cap = M.dense_cap, nvs = M.dense_nvs;
  o.art = L.g_dense; o.yd = o.art + cap * cap; o.rf = o.yd + cap * nvs;
  o.rlo = o.rf + cap; o.rhi = o.rlo + cap; o.rarr = o.rhi + cap; o.rt = o.rarr + cap; o.rmap = o.rt + cap;
  return o;
}

typedef float mjh_f4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ build
// LDS: s_minv[NVS] | s_inv[CAP] | s_row[CAP] int4 | s_start[maxblk + 1]  (a few KB: several workgroups per CU)
#ifndef DN_BUILD_THREADS
#define DN_BUILD_THREADS 512
#endif
__global__ __launch_bounds__(DN_BUILD_THREADS) void mjh_dense_build_kernel(const DConst* __restrict__ C, const DState S, int env0) {
  const DModel& M = C->M; const Lay& L = C->L;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nv = M.nv;
  const int env = S.env_order ? S.env_order[env0 + blockIdx.x] : env0 + (int)blockIdx.x;
  float* const gs = S.gscratch + (size_t)env * (size_t)S.gstride;
  int* meta = (int*)(gs + L.g_meta);
  const int nblk = __builtin_amdgcn_readfirstlane(meta[0]), nefc = __builtin_amdgcn_readfirstlane(meta[2]);
  const int cap = M.dense_cap, nvs = M.dense_nvs;
  if (nblk == 0 || nefc > cap || nefc <= 0) return;                    // (meta[7] is 0: the block solver takes this env)
  const int nr16 = (nefc + 15) & ~15, nr32 = (nefc + 31) & ~31, K = (nr32 + 63) >> 6, W = 64 * K;
  const DenseOff o = dense_off(M, L);
  float* s_minv = lds; float* s_inv = s_minv + nvs;
  int4* s_row = (int4*)(s_inv + cap); int* s_start = (int*)(s_row + cap);
  const int* g_ord = (const int*)(gs + (-1 - L.order)); const int4* g_hd = (const int4*)(gs + (-1 - L.blki));
  const float* g_bf = gs + (-1 - L.blkf); const float* g_Y = gs + (-1 - L.B);      // (this env's assemble stored Y = J L^-1 there)
  // J a0 per base row: the assemble launch's warm start left J qacc_smooth and J da (da = M^-1 J^T f_warm, zero if rejected) in the pools
  const float* g_ja = gs + (-1 - L.phi); const float* g_jda = gs + (-1 - L.bv);
  for (int d = tid; d < nvs; d += DN_BUILD_THREADS) s_minv[d] = d < nv ? gs[L.g_minv + d] : 0.0f;
  for (int q = tid; q < cap; q += DN_BUILD_THREADS) s_inv[q] = 0.0f;
  // ---- rows in visiting order: position i of the order -> block, its rows start at the running sum of the row counts
  if (wid == 0) {
    int run = 0;
    for (int base = 0; base < nblk; base += 64) {
      const int i = base + lane;
      const int nr = i < nblk ? (g_hd[g_ord[i]].x >> 4) & 15 : 0;
      const int incl = wave_incl_scan_i(nr, lane);
      if (i < nblk) s_start[i] = run + incl - nr;
      run += wave_last_i(incl);
    }
  }
  __syncthreads();
  for (int i = tid; i < nblk; i += DN_BUILD_THREADS) {
    const int b = g_ord[i]; const int4 hd = g_hd[b];
    const int nr = (hd.x >> 4) & 15, p0 = s_start[i], sl4 = BLK_SLOTS(hd.y) == 4 ? 1 : 0;
    for (int r = 0; r < nr; r++) s_row[p0 + r] = make_int4(hd.x, hd.z, hd.w, b | (r << 16) | (sl4 << 24));
  }
  __syncthreads();
  // ---- dense rows, 16 lanes (one DPP row) per matrix row, 16 rows per pass: Y_p expanded to nv columns (compact storage over
  //      <= 2 trees, pyramid rows n +- k folded), written to global for the fragments below; the same pass forms
  //      AR_pp = Y_p D^-1 Y_p^T + R  (sum over the 16 lanes by DPP) and from it the row's start values, stored in the lanes' layout
  //      [lane][K] of the sweep kernel
  float* g_yd = gs + o.yd;
  const int sub = tid & 15;
  for (int p = tid >> 4; p < W; p += DN_BUILD_THREADS / 16) {
    int4 ri = make_int4(0, 0, 0, 0);
    if (p < nefc) ri = s_row[p];
    ROW_TREES(ri.y, ri.z);
    const int kind = ri.x & 15, jo = BLK_JOFF(ri.x), r = (ri.w >> 16) & 255, b = ri.w & 0xffff; const bool quad = (ri.w >> 24) & 1;
    const int kk = 1 + (r >> 1); const float sg = (r & 1) ? -1.0f : 1.0f;
    float jb = 0;
    if (p < nr16)
#pragma unroll 8
      for (int d = sub; d < nvs; d += 16) {
        float yv = 0;
        const int k = (p < nefc && d < nv) ? row_off(d, a1, n1, a2, n2) : -1;
        if (k >= 0) {
          if (quad) {
            yv = g_Y[jo + 4*k];
            if (kind != BK_SINGLE) yv += sg * g_Y[jo + 4*k + kk];
          } else yv = g_Y[jo + k];
        }
        g_yd[p * nvs + d] = yv;
        jb += yv * yv * s_minv[d];
      }
    // sums over the 16 lanes of the DPP row: lane 15 of the row ends up with the totals
    MJH_DPP_ADD(jb, 0x111, 0xf, true); MJH_DPP_ADD(jb, 0x112, 0xf, true); MJH_DPP_ADD(jb, 0x114, 0xf, true); MJH_DPP_ADD(jb, 0x118, 0xf, true);
    if (sub == 15) {
      float f = 0, lo = 0, hi = 0, arr = 1.0f, t = 0; int map = -1;
      if (p < nefc) {
        const float* bf = g_bf + b * BLKF_STRIDE;
        const float R = bf[0];
        const float aref = bf[BF_AREF] + (kind != BK_SINGLE ? sg * bf[BF_AREF + kk] : 0.0f);
        f = bf[BF_F + r]; lo = bf[BF_LO]; hi = bf[BF_LO + 1];
        const float ja = (g_ja[4 * b] + g_jda[4 * b]) + (kind != BK_SINGLE ? sg * (g_ja[4 * b + kk] + g_jda[4 * b + kk]) : 0.0f);
        arr = jb + R;
        const float inv = 1.0f / arr;
        t = -(ja - aref + R * f) * inv;
        map = b | (r << 16);
        s_inv[p] = inv;
      }
      const int ix = (p & 63) * K + (p >> 6);
      gs[o.rf + ix] = f; gs[o.rlo + ix] = lo; gs[o.rhi + ix] = hi; gs[o.rarr + ix] = arr; gs[o.rt + ix] = t;
      ((int*)(gs + o.rmap))[ix] = map;
    }
  }
  __syncthreads();
  // ---- AR = (Y D^-1) Y^T on the matrix cores in 16 x 16 tiles.  A work item = a 16-row panel P and a column class Q0 = Q mod 4: its K
  //      tiles Q0, Q0 + 4, ... hold, for one lane index, exactly the K row owners q, q + 64, ... whose values are adjacent in the
  //      sweep kernel's layout [p][lane][K] — so a lane stores K adjacent floats and a 16-lane DPP row a contiguous run (the stores
  //      are the traffic of this kernel: 4-byte scatters cost it 2x).  Items are dealt round-robin to the eight waves.  Per 16-dof
  //      chunk a lane fetches 16 bytes of its A row and of each B row (both from Y; the A side times 1 / D) — L1 / L2 hits, the rows were just written by
  //      this workgroup — and issues four v_mfma_f32_16x16x4_f32 per tile (the k index inside a chunk is permuted identically on
  //      both sides: lane l carries k = 4 (l >> 4) + j, j = 0..3, so that a lane's four values are contiguous).
  //      D: col = l & 15, row = 4 (l >> 4) + v.  Stored column-scaled: AR'_pq = -AR_pq / AR_qq, diagonal -1.  Panels and tiles
  //      beyond the rows are zero without any arithmetic: the inert border of the sweep.
  const int TD = nr16 >> 4, TP = nr32 >> 4, nch = nvs >> 4;
  const int li = lane & 15, lk = lane >> 4;
  float* g_art = gs + o.art;
  for (int item = wid; item < 4 * TP; item += DN_BUILD_THREADS / 64) {
    const int P = item >> 2, Q0 = item & 3;
    mjh_f4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = (mjh_f4){0, 0, 0, 0};
    if (P < TD) {
      const float* ap = g_yd + (16 * P + li) * nvs + 4 * lk;
      // fragments of chunk c + 1 are requested before the matrix instructions of chunk c (a tile that does not exist reads row 0)
      const float* bp[4];
#pragma unroll
      for (int j = 0; j < 4; j++) bp[j] = g_yd + (16 * (Q0 + 4 * j < TD ? Q0 + 4 * j : 0) + li) * nvs + 4 * lk;
      float4 a = *(const float4*)ap, bq[4];
      { const float4 mv = *(const float4*)(s_minv + 4 * lk); a.x *= mv.x; a.y *= mv.y; a.z *= mv.z; a.w *= mv.w; }
#pragma unroll
      for (int j = 0; j < 4; j++) bq[j] = *(const float4*)bp[j];
      for (int c = 0; c < nch; c++) {
        const int cn = c + 1 < nch ? c + 1 : c;
        float4 an = *(const float4*)(ap + 16 * cn);
        { const float4 mv = *(const float4*)(s_minv + 16 * cn + 4 * lk); an.x *= mv.x; an.y *= mv.y; an.z *= mv.z; an.w *= mv.w; }
        float4 bn[4];
#pragma unroll
        for (int j = 0; j < 4; j++) bn[j] = *(const float4*)(bp[j] + 16 * cn);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (Q0 + 4 * j < TD) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq[j].x, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq[j].y, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq[j].z, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq[j].w, acc[j], 0, 0, 0);
          }
        }
        a = an;
#pragma unroll
        for (int j = 0; j < 4; j++) bq[j] = bn[j];
      }
    }
    const int ql = 16 * Q0 + li;                          // lane index of the row owners q = ql + 64 j
    float iq[4];
#pragma unroll
    for (int j = 0; j < 4; j++) iq[j] = (j < K && ql + 64 * j < cap) ? s_inv[ql + 64 * j] : 0.0f;
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int p = 16 * P + 4 * lk + v;
      float out[4];
#pragma unroll
      for (int j = 0; j < 4; j++) out[j] = (p == ql + 64 * j) ? 0.0f : -acc[j][v] * iq[j];     // (no self-coupling: the sweep carries s = f + t, which a row's own update leaves unchanged)
      float* dst = g_art + p * W + ql * K;
      if (K == 1) dst[0] = out[0];
      else if (K == 2) *(float2*)dst = make_float2(out[0], out[1]);
      else if (K == 3) { dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2]; }
      else *(float4*)dst = make_float4(out[0], out[1], out[2], out[3]);
    }
  }
  __syncthreads();
  if (tid == 0) meta[DN_META_DENSE] = 1;
}

// ------------------------------------------------------------------------------------------------ sweeps
#ifndef DN_GROUP_ROWS
#define DN_GROUP_ROWS(K) ((K) == 3 ? 32 : 16)     // rows per fetch group of the streamed sweeps (dn_sweeps)
#endif
template <int K> struct DnCol { float v[K]; };
// The value a lane's row has at ITS visit is kept by ONE data-parallel move: v_mov_b32_dpp under a row mask and a bank mask writes the
// four consecutive lanes of the visited lane's bank — so a set of 64 rows carries four capture registers, one per position inside a
// bank: the visit of lane l writes register l % 4 in the four lanes of its bank, and a lane only ever reads the register of its own
// position (the other three hold its neighbours' leftovers).  Replaces `s_lshl_b64` on a scalar lane mask + `v_cndmask_b32` (9.5 + 19.3
// clocks of a lone wave's issue, profiles/r02m_valu_issue_bench.txt) by one 9-clock instruction per row; same values, bitwise.
#ifndef DN_RES_PLAIN
#define DN_RES_PLAIN 1
#endif
#ifndef DN_LOAD_BURST
#define DN_LOAD_BURST 0
#endif
#ifndef DN_STRAIGHT
#define DN_STRAIGHT 0
#endif
#ifndef DN_K3_PAIR
#define DN_K3_PAIR 1
#endif
#ifndef DN_DPP_CAPTURE
#define DN_DPP_CAPTURE 1
#endif
// (l is a constant once the row loops are unrolled: the switch folds to its one case — the builtin wants literal masks)
DEV void dn_capture(float (&q)[4], const float v, const int l) {
  float& d = q[l & 3];
#define DN_CC(c) case c: d = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, d), __builtin_bit_cast(int, v), 0xE4, 1 << ((c) >> 2), 1 << ((c) & 3), false)); break;
  switch (l >> 2) { DN_CC(0) DN_CC(1) DN_CC(2) DN_CC(3) DN_CC(4) DN_CC(5) DN_CC(6) DN_CC(7) DN_CC(8) DN_CC(9) DN_CC(10) DN_CC(11) DN_CC(12) DN_CC(13) DN_CC(14) DN_CC(15) default: break; }
#undef DN_CC
}
DEV float dn_captured(const float (&q)[4], const int lane) {
  const int i = lane & 3;
  return i == 0 ? q[0] : (i == 1 ? q[1] : (i == 2 ? q[2] : q[3]));
}
template <int K> DEV DnCol<K> dn_load(const __amdgpu_buffer_rsrc_t rs, const unsigned voff, const int soff) {
  DnCol<K> c;
  if constexpr (K == 1) { c.v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, soff, 0)); }
  else if constexpr (K == 2) { const mjh_v2u u = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, soff, 0); __builtin_memcpy(c.v, &u, 8); }
  else if constexpr (K == 3) { typedef unsigned v3u __attribute__((vector_size(12))); const v3u u = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)voff, soff, 0); __builtin_memcpy(c.v, &u, 12); }
  else { const mjh_v4u u = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0); __builtin_memcpy(c.v, &u, 16); }
  return c;
}

// Gauss-Seidel sweeps over the rows p = 0 .. nr32 - 1 (visiting order; nr32 a multiple of 32, rows beyond nefc inert).  Lane q owns
// rows 64 k + q and carries  s_q = f_q + t_q  (t_q = -res_q / AR_qq):  the update of row p is  f_p <- med3(s_p, lo_p, hi_p)  and
// s_q += AR'_pq delta for q != p — a row's own update leaves its s unchanged (its t moves by -delta, its f by +delta), so the stored
// diagonal is 0 and the row costs: med3, sub, readlane, one select, K multiply-adds in pairs (v_pk_fma_f32) (+ its share of the column
// fetch): 7 instructions at K = 3.  The select keeps s_p as it was at the row's visit (the row's new force is med3 of it, taken at the end of the sweep): with it the decrease of the dual cost, -delta (res + AR_pp delta / 2), is summed per
// row exactly as mj_solPGS (and the block solver, and the oracle) sum it, once per 64-row set.  (Evaluating the sweep's decrease from
// its end points, 1/2 (f1 - f0) . (res1 + res0), saves that select but ends a sweep later on average — C4: 15.9 against 14.8
// sweeps — because the rounding noise of t enters with either sign.)
// The columns AR'_p. (64 K floats per row, a lane's K values contiguous) are fetched a 16-row group ahead into two register buffers;
// the group count is even, so the buffer parity is static over the wrap of a sweep.  Returns the sweep count.
template <int K>
DEV int dn_sweeps(const __amdgpu_buffer_rsrc_t rs, const int art_bytes, const int nr32, const int itmax, const float tol, const float scale,
                  float* f, float* s, const float* lo, const float* hi, const float* arr, const int lane) {
  // K = 3 (129 - 192 rows, the slowest envs of a C4 launch): groups of 32 rows — twice the bytes in flight per wave.  A sweep over an odd
  // number of groups gets one idle slot at its end, in which the first group of the next sweep is requested (buffer parity stays static).
  constexpr int GR = DN_GROUP_ROWS(K);
  static_assert(GR == 16 || GR == 32, "groups must tile the 32-row padding of the sweep (rows beyond it are not initialised)");
  constexpr int GPK = 64 / GR;
  const int G = nr32 / GR, GE = (G + 1) & ~1;
  const unsigned voff = (unsigned)lane * (unsigned)(4 * K);
  constexpr int ROWB = 64 * K * 4;                       // bytes per row of AR'
  DnCol<K> buf[2][GR];
#pragma unroll
  for (int r = 0; r < GR; r++) buf[0][r] = dn_load<K>(rs, voff, art_bytes + r * ROWB);
  int niter = 0;
  // (the sweep carries r = s - f and the interval relative to f: a row's delta is ONE med3 — see dn_sweeps_resident)
#pragma unroll
  for (int k = 0; k < K; k++) s[k] -= f[k];
  for (;;) {
    float sv[K], lor[K], hir[K];
#pragma unroll
    for (int k = 0; k < K; k++) { sv[k] = s[k]; lor[k] = lo[k] - f[k]; hir[k] = hi[k] - f[k]; }      // sv: r of the lane's row at ITS visit (rows never visited are inert: lo = hi = f = 0)
#if DN_DPP_CAPTURE
    float svq[K][4];
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
      for (int i = 0; i < 4; i++) svq[k][i] = s[k];
#endif
    // (opaque per sweep: otherwise the 64 lane masks and the row offsets are hoisted out of the sweep loop as loop invariants and
    //  spilled — v_writelane / v_readlane around every use)
    unsigned long long m1 = 1ull; asm volatile("" : "+s"(m1));
#pragma unroll
    for (int k = 0; k < K; k++) {
#pragma unroll
      for (int gg = 0; gg < GPK; gg++) {
        const int g = GPK * k + gg;
#if DN_STRAIGHT
        // Straight-line fetch schedule: EVERY one of the K * GPK static slots requests its successor's group (an idle slot behind the env's
        // last group, or a group that does not exist: the first group once more, never used) — with the request under a branch (`gn < G`) the
        // compiler's wait-count pass merges the path without it and makes row r of THIS group wait for the r-th fetch of the burst just
        // issued (s_waitcnt vmcnt(31 - r) instead of vmcnt(63 - r)): the group-ahead buffers bought no look-ahead at all, every group
        // started with a full memory round trip.  The row offsets go into the instruction's immediate field (voffset + constant: the backend
        // splits it into six loop-invariant bases + 12 bits) instead of one scalar add per fetch.
        {
          constexpr int NS = GPK * K;
          const int gn = g + 1 < NS ? g + 1 : 0;
          {
            int sb = art_bytes + (gn < G ? gn : 0) * GR * ROWB; asm volatile("" : "+s"(sb));
#pragma unroll
            for (int r = 0; r < GR; r++) buf[(g + 1) & 1][r] = dn_load<K>(rs, voff + (unsigned)(r * ROWB), sb);
          }
          if (false) {
#else
        if (g < GE) {
          const int gn = g + 1 < GE ? g + 1 : 0;
          if (gn < G) {
#endif
#ifdef DN_PROBE_SAMEGROUP     // timing probe (tools/r04_dense_probe.sh): every fetch aimed at the first group — a near cache; results are garbage
            int sb = art_bytes; asm volatile("" : "+s"(sb));
#else
            int sb = art_bytes + gn * GR * ROWB; asm volatile("" : "+s"(sb));
#endif
#pragma unroll
            for (int r = 0; r < GR; r++) buf[(g + 1) & 1][r] = dn_load<K>(rs, voff, sb + r * ROWB);
#if DN_LOAD_BURST
            // (the next group's fetches as one burst in front of this group's rows: interleaved with the rows, every row waits on its own
            //  fetch counter value — an s_waitcnt per row is an issue slot per row of a lone wave)
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
          if (g < G) {
            [[maybe_unused]] unsigned long long mask = m1 << (GR * gg);
#pragma unroll
            for (int r = 0; r < GR; r++) {
              const int l = GR * gg + r;
              const float d = __builtin_amdgcn_fmed3f(s[k], lor[k], hir[k]);
              const float sd = readlane_f(d, l);
#if DN_DPP_CAPTURE
              dn_capture(svq[k], s[k], l);          // (f itself is not touched inside the sweep: a row is visited once, its new force is med3 of the captured value)
#else
              const bool me = __builtin_amdgcn_inverse_ballot_w64(mask);     // lane l: v_cndmask on a scalar mask, no compare
              mask <<= 1;
              sv[k] = me ? s[k] : sv[k];            // (f itself is not touched inside the sweep: a row is visited once, its new force is med3 of sv)
#endif
#ifdef DN_WAIT4
              // (one s_waitcnt per four rows: touching the fourth row's operands here makes the compiler wait for that fetch — the fetches complete in
              //  order —, and the three rows in between need none)
              if constexpr (DN_WAIT4 > 1) if (r % DN_WAIT4 == 0 && r + DN_WAIT4 - 1 < GR) asm volatile("" :: "v"(buf[g & 1][r + DN_WAIT4 - 1].v[0]));
#endif
              if constexpr (K == 3 && GR == 32 && !DN_K3_PAIR) {
                // (three plain multiply-adds: the pair form wants its operands in aligned register pairs, which a 96-bit fetch does not
                //  deliver — with 32-row buffers the copies cost 120 registers)
#pragma unroll
                for (int j = 0; j < K; j++) s[j] = __builtin_fmaf(buf[g & 1][r].v[j], sd, s[j]);
              } else {
                // (pairs of the lane's rows in one v_pk_fma_f32: 2 instead of 3 instructions at K = 3)
                typedef float dn_f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int j = 0; j + 1 < K; j += 2) {
                  const dn_f2 c2 = {buf[g & 1][r].v[j], buf[g & 1][r].v[j + 1]}, d2 = {sd, sd}, s2 = {s[j], s[j + 1]};
                  const dn_f2 o2 = __builtin_elementwise_fma(c2, d2, s2);
                  s[j] = o2.x; s[j + 1] = o2.y;
                }
                if (K & 1) s[K - 1] = __builtin_fmaf(buf[g & 1][r].v[K - 1], sd, s[K - 1]);
              }
            }
          }
        }
      }
    }
    niter++;
    float imp = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {      // -delta (res + AR delta / 2), res = -t AR, t = s - f = r at the row's visit
#if DN_DPP_CAPTURE
      sv[k] = dn_captured(svq[k], lane);
#endif
      const float dc = __builtin_amdgcn_fmed3f(sv[k], lor[k], hir[k]);
      imp += dc * arr[k] * (sv[k] - 0.5f * dc);
      f[k] += dc; s[k] -= dc;          // (r = s - f: the row's own update leaves s where it was)
    }
    const float improvement = wave_sum<4>(imp);
    if (improvement * scale < tol || niter >= itmax) break;
  }
  return niter;
}

// The same sweeps with the lane's part of AR' RESIDENT in registers (K <= 2: up to 128 rows, 64 K floats per lane): fetched once,
// no memory operand inside the sweeps.  What the streamed form above waits for is its column fetch (768 B per row and wave through the
// CU's vector-memory path, every sweep again — tools/r04_dense_probe.sh), so the envs that fit (the mean env: 107 rows) stop competing
// for that path with the ones that do not (129+ rows: the launch's slowest).  Same rows, same order, same arithmetic: bitwise the
// streamed form's iterates.
#ifndef DN_RESIDENT_MAXK
#define DN_RESIDENT_MAXK 2
#endif
template <int K>
DEV int dn_sweeps_resident(const __amdgpu_buffer_rsrc_t rs, const int art_bytes, const int nr32, const int itmax, const float tol, const float scale,
                           float* f, float* s, const float* lo, const float* hi, const float* arr, const int lane) {
  const int G = nr32 >> 4;
  const unsigned voff = (unsigned)lane * (unsigned)(4 * K);
  constexpr int ROWB = 64 * K * 4;                       // bytes per row of AR'
  DnCol<K> A[64 * K];
#pragma unroll
  for (int g = 0; g < 4 * K; g++) {
    if (g < G) {
#pragma unroll
      for (int r = 0; r < 16; r++) A[16 * g + r] = dn_load<K>(rs, voff, art_bytes + (16 * g + r) * ROWB);
    } else {
#pragma unroll
      for (int r = 0; r < 16; r++)
#pragma unroll
        for (int j = 0; j < K; j++) A[16 * g + r].v[j] = 0.0f;
    }
  }
  int niter = 0;
#ifdef DN_REFRESH
  float s0[K], f0[K];
#pragma unroll
  for (int k = 0; k < K; k++) { s0[k] = s[k]; f0[k] = f[k]; }
#endif
  // The sweep carries  r = s - f  and the projection interval relative to the force,  [lo - f, hi - f]  (f does not move inside a sweep: a row
  // is visited once), so that a row's delta  med3(s, lo, hi) - f  is ONE instruction,  med3(r, lo - f, hi - f): six instead of seven
  // VALU instructions per row at K = 3, and one less on the row-to-row chain (round 5: C4's dense solve launch 501 -> ~440 us).  Same
  // iteration in exact arithmetic; in fp32 the iterates differ from the s-form by rounding (the new force is f + delta instead of the
  // clamped s: identical at an active bound lo = 0, where lo - f = -f exactly).
#pragma unroll
  for (int k = 0; k < K; k++) s[k] -= f[k];
  for (;;) {
    float sv[K], lor[K], hir[K];
#pragma unroll
    for (int k = 0; k < K; k++) { sv[k] = s[k]; lor[k] = lo[k] - f[k]; hir[k] = hi[k] - f[k]; }
#if DN_DPP_CAPTURE
    float svq[K][4];
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
      for (int i = 0; i < 4; i++) svq[k][i] = s[k];
#endif
    unsigned long long m1 = 1ull; asm volatile("" : "+s"(m1));
#pragma unroll
    for (int k = 0; k < K; k++) {
#pragma unroll
      for (int gg = 0; gg < 4; gg++) {
        const int g = 4 * k + gg;
        if (g < G) {
          [[maybe_unused]] unsigned long long mask = m1 << (16 * gg);
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int l = 16 * gg + r;
            const float d = __builtin_amdgcn_fmed3f(s[k], lor[k], hir[k]);
            const float sd = readlane_f(d, l);
#if DN_DPP_CAPTURE
            dn_capture(svq[k], s[k], l);
#else
            const bool me = __builtin_amdgcn_inverse_ballot_w64(mask);
            mask <<= 1;
            sv[k] = me ? s[k] : sv[k];
#endif
#if DN_RES_PLAIN
            // (plain multiply-adds: the pair form needs a wait state in front of the med3 that reads its result and two behind the v_readlane
            //  whose scalar it reads — as s_nops they are issue slots of a lone wave like any instruction; the off-chain multiply-add fills one)
#pragma unroll
            for (int j = 0; j < K; j++) s[j] = __builtin_fmaf(A[16 * g + r].v[j], sd, s[j]);
#else
            typedef float dn_f2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int j = 0; j + 1 < K; j += 2) {
              const dn_f2 c2 = {A[16 * g + r].v[j], A[16 * g + r].v[j + 1]}, d2 = {sd, sd}, s2 = {s[j], s[j + 1]};
              const dn_f2 o2 = __builtin_elementwise_fma(c2, d2, s2);
              s[j] = o2.x; s[j + 1] = o2.y;
            }
            if (K & 1) s[K - 1] = __builtin_fmaf(A[16 * g + r].v[K - 1], sd, s[K - 1]);
#endif
          }
        }
      }
    }
    niter++;
    float imp = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
#if DN_DPP_CAPTURE
      sv[k] = dn_captured(svq[k], lane);
#endif
      const float dc = __builtin_amdgcn_fmed3f(sv[k], lor[k], hir[k]);
      imp += dc * arr[k] * (sv[k] - 0.5f * dc);
      f[k] += dc; s[k] -= dc;
    }
    const float improvement = wave_sum<4>(imp);
    if (improvement * scale < tol || niter >= itmax) break;
#ifdef DN_REFRESH
    // (accuracy probe: every DN_REFRESH sweeps the carried residual is formed anew from the start values and the force changes — what it drifts by)
    if (niter % DN_REFRESH == 0) {
      float acc[K], df[K];
#pragma unroll
      for (int k = 0; k < K; k++) { acc[k] = s0[k]; df[k] = f[k] - f0[k]; }
#pragma unroll
      for (int k = 0; k < K; k++)
#pragma unroll
        for (int gg = 0; gg < 4; gg++) {
          const int g = 4 * k + gg;
          if (g < G) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
              const float sd = readlane_f(df[k], 16 * gg + r);
#pragma unroll
              for (int j = 0; j < K; j++) acc[j] = __builtin_fmaf(A[16 * g + r].v[j], sd, acc[j]);
            }
          }
        }
#pragma unroll
      for (int k = 0; k < K; k++) s[k] = acc[k] - f[k];
    }
#endif
  }
  return niter;
}

template <int K>
DEV void dn_solve_env(const DModel& M, const Lay& L, float* gs, const __amdgpu_buffer_rsrc_t rs, const DenseOff& o, const int nefc, const int nr32,
                      float* s_df, float* s_x, const float* s_qld, const int* s_anc, const int lane, int* meta) {
  float f[K], f0[K], t[K], lo[K], hi[K], arr[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int ix = lane * K + k;
    f[k] = gs[o.rf + ix]; f0[k] = f[k]; t[k] = gs[o.rt + ix]; lo[k] = gs[o.rlo + ix]; hi[k] = gs[o.rhi + ix]; arr[k] = gs[o.rarr + ix];
  }
  const float scale = 1.0f / (M.meaninertia * (float)(M.nv > 1 ? M.nv : 1));
#pragma unroll
  for (int k = 0; k < K; k++) t[k] += f[k];                  // s = f + t
  int niter;
  if constexpr (K <= DN_RESIDENT_MAXK) niter = dn_sweeps_resident<K>(rs, 4 * o.art, nr32, M.iterations, M.tolerance, scale, f, t, lo, hi, arr, lane);
  else niter = dn_sweeps<K>(rs, 4 * o.art, nr32, M.iterations, M.tolerance, scale, f, t, lo, hi, arr, lane);
  // forces back into the block records (the integrate launch forms qfrc_constraint from them), force changes to LDS
  const int* rmap = (const int*)(gs + o.rmap);
  float* g_bf = gs + (-1 - L.blkf);
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int q = 64 * k + lane, map = rmap[lane * K + k];
    s_df[q] = f[k] - f0[k];
    if (map >= 0) g_bf[(map & 0xffff) * BLKF_STRIDE + BF_F + ((map >> 16) & 255)] = f[k];
  }
  __syncthreads();
  // qacc = a0 + M^-1 J^T (f - f0) = a0 + L^-1 D^-1 Y^T (f - f0): lanes = dofs
  const int nv = M.nv, nvs = M.dense_nvs;
  const float* g_yd = gs + o.yd;
  float q0 = 0.0f, q1 = 0.0f;
  // (eight rows per round, their fetches issued together: one row per round with a branch on df was a chain of nefc dependent
  //  round trips to L2 / HBM — 100+ us per env, every env; a zero df adds an exact zero, rows are summed in the same order)
#ifdef DN_TAIL_SERIAL
  for (int p = 0; p < nefc; p++) {
    const float df = s_df[p];
    if (df == 0.0f) continue;
    if (lane < nvs) q0 += g_yd[p * nvs + lane] * df;
    if (lane + 64 < nvs) q1 += g_yd[p * nvs + lane + 64] * df;
  }
#else
  const bool c0 = lane < nvs, c1 = lane + 64 < nvs;
  for (int p0 = 0; p0 < nefc; p0 += 8) {
    float y0[8], y1[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int p = p0 + j < nefc ? p0 + j : nefc - 1;
      y0[j] = c0 ? g_yd[p * nvs + lane] : 0.0f;
      y1[j] = c1 ? g_yd[p * nvs + lane + 64] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float df = p0 + j < nefc ? s_df[p0 + j] : 0.0f;
      q0 += y0[j] * df; q1 += y1[j] * df;
    }
  }
#endif
  if (lane < nv) s_x[lane] = q0 * gs[L.g_minv + lane];
  if (lane + 64 < nv) s_x[lane + 64] = q1 * gs[L.g_minv + lane + 64];
  __syncthreads();
  tree_l_levels(s_x, s_qld, s_anc, M.I + M.o_dof_Madr, nv, M.nM, lane);
  if (lane < nv) gs[L.g_qacc + lane] = gs[L.g_a0 + lane] + s_x[lane];
  if (lane + 64 < nv) gs[L.g_qacc + lane + 64] = gs[L.g_a0 + lane + 64] + s_x[lane + 64];
  if (lane == 0) meta[5] = niter;
}

__global__ __launch_bounds__(64) void mjh_dense_solve_kernel(const DConst* __restrict__ C, const DState S, int env0) {
  const DModel& M = C->M; const Lay& L = C->L;
  // LDS: s_df[DN_CAP_MAX] | s_x[128] | factor L [nM] | ancestor lists [nM]
  extern __shared__ float lds[];
  float* s_df = lds; float* s_x = s_df + DN_CAP_MAX; float* s_qld = s_x + 128; int* s_anc = (int*)(s_qld + M.nM);
  const int lane = threadIdx.x;
  const int env = S.env_order ? S.env_order[env0 + blockIdx.x] : env0 + (int)blockIdx.x;
  float* const gs = S.gscratch + (size_t)env * (size_t)S.gstride;
  int* meta = (int*)(gs + L.g_meta);
  if (__builtin_amdgcn_readfirstlane(meta[0]) == 0 || __builtin_amdgcn_readfirstlane(meta[DN_META_DENSE]) == 0) return;
  const int nefc = __builtin_amdgcn_readfirstlane(meta[2]);
  const int nr32 = (nefc + 31) & ~31, K = (nr32 + 63) >> 6;
  const DenseOff o = dense_off(M, L);
  for (int i = lane; i < M.nM; i += 64) { s_qld[i] = gs[L.g_qLD + i]; s_anc[i] = ((const int*)(gs + L.g_anc))[i]; }   // (used after the sweeps)
  __amdgpu_buffer_rsrc_t rs;
  {
    const unsigned long long ga = (unsigned long long)gs;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ga), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ga >> 32));
    const long long nb = S.gstride * 4;
    rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)(nb > 0x7ffffff0ll ? 0x7ffffff0ll : nb), 0x00020000);
  }
  if (K == 1) dn_solve_env<1>(M, L, gs, rs, o, nefc, nr32, s_df, s_x, s_qld, s_anc, lane, meta);
  else if (K == 2) dn_solve_env<2>(M, L, gs, rs, o, nefc, nr32, s_df, s_x, s_qld, s_anc, lane, meta);
  else if (K == 3) dn_solve_env<3>(M, L, gs, rs, o, nefc, nr32, s_df, s_x, s_qld, s_anc, lane, meta);
  else dn_solve_env<4>(M, L, gs, rs, o, nefc, nr32, s_df, s_x, s_qld, s_anc, lane, meta);
}
