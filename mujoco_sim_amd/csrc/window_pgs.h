// window_pgs.h — projected Gauss-Seidel in mj_solPGS's own constraint-row order for small free-body models (every tree a single free
// body, nv <= 32: BASELINE's 24-DoF scene), FOUR environments per wavefront, operands in registers.  Replaces the sweeps of the fused
// step (mj_step2's mj_fwdConstraint -> mj_solPGS, /root/reference/src/mj_main.cpp:108) in mjh_step for these models:
//
//     mjh_step_kernel  (PH_PRE: one env per wave, everything up to the constraint rows; window_emit() below hands the rows over)
//  -> mjh_window_kernel (this file: warm start, sweeps, mj_checkAcc, Euler, state / statistics written back)
//
// Why.  The fused kernel keeps one environment per wavefront because the stages around the solver need ~20 KB of LDS per env: 8 envs
// per CU, 2 waves per SIMD, and the sweeps — a chain of dependent 16-lane operations that leaves three quarters of every wave-step idle
// in the reference's order (the conflict DAG of a 4-box pile is 5.7 patches deep) — run at the latency of that chain.  The sweeps
// themselves need no LDS at all: a WINDOW of 16 consecutive constraint rows is one row per lane of a 16-lane DPP row, so a wavefront
// holds four environments, and with one such wave per SIMD and the row records in the wave's 512 registers all 4096 environments of
// the metric are resident at once (16 per CU).  Gauss-Seidel inside a window is exact and sequential (the strictly lower triangle of
// the window's AR = J^ J^T + R is precomputed: t_q += (-AR_qr / AR_qq) delta_r, two instructions per row for all four envs); between
// windows the running acceleration a^ = M^1/2 a lives in two registers per env (lane q: dofs q and 16 + q) and is updated by a
// transpose-reduce over the 16 lanes — no LDS, no atomics, no schedule: the order IS the constraint order, row by row.
//
//     u = J^ a^ (row_newbcast multiply-adds) -> t = -(u - aref + R f) / AR_qq -> 16 rows -> delta -> a^ += J^T delta
//
// Same scaled coordinates as patch_pgs.h (J^ = J M^-1/2: M is diagonal for these models).  Rows: contacts only (f >= 0).
#pragma once

// per-env slice of DState::wbuf (floats).  Header ints: [0] rows handed over (0: nothing to do for the window kernel — no rows, or the
// env finished the step in the assemble launch), [1] ncon, [2] nefc, [3] flags
#define WN_AS 16        // M^1/2 qacc_smooth  [32]
#define WN_AWS 48       // M^1/2 qacc_warmstart [32]
#define WN_SINV 80      // M^-1/2 per dof [32]
#define WN_QVEL 112     // qvel after the controller [32]
#define WN_QPOS 144     // qpos after mj_kinematics' quaternion normalisation [40]
#define WN_ROWS 192     // window w at WN_ROWS + w * WN_NK * 16: [k][16 rows], k < WN_NK
#define WN_NK 16        // floats of a handed-over row: a contact row touches at most two free bodies — J^ over the first one's six dofs [0..5], over the second's [6..11],
                        // the two body slots (dof address / 6; -1: none) as integers [12], [13], aref [14], R [15]; the window kernel expands it over its dof slots
#define WN32_MIN_ROWS 96   // rows above which an env is swept in 32-row windows (3.5 % of S24's envs: the ones a cohort's step waits for)
#define WN_MAXW 24      // most windows per env any model gets (384 rows); a model's own capacity: DModel::win_maxw = min(WN_MAXW, ceil(maxefc / 16))
// Sweeping the tiers' windows in pairs as well (-DWN_TIER_PAIRS=1: the filled chain on two loaded records and their cross tile) was measured
// and is off: with six register-resident windows the kernel grows to 490 registers (S24 13.3 -> 11.5 M, S24D 5.3 -> 4.7 M), with four
// resident windows and the pairs S24D runs at 4.65 M against 5.03 M — a tier window's sweep waits for its record's loads (a lone
// wavefront hides no latency), and a pair waits for twice as many before its first instruction (HISTORY.md Round 5).
#ifndef WN_TIER_PAIRS
#define WN_TIER_PAIRS 0
#endif
#define WN_XPAIR (WN_TIER_PAIRS ? 16 : 0)     // cross tile of a PAIR of such windows, per row
#define WN_XREC(nvt) (((nvt) + 24) / 4 * 4)   // (floats, a multiple of four: the record is read in 16-byte pieces) record of a window beyond the register-resident ones, per row: J^[nvt], aref, R, 16 tile entries, -1 / AR_qq, AR_qq / 2, force
#ifndef WN_NW24
#define WN_NW24 6       // register-resident windows of the 24-dof instance (96 rows: what the 16-row form meets with the 32-row section on; S24 has 72 on average), the rest is streamed
#endif
#ifndef WN_NW32
#define WN_NW32 6
#endif
#ifndef WN32_NW
#define WN32_NW 4       // 32-row windows of an env (128 rows), all register-resident
#endif
#define WN64_MIN_ROWS 192  // rows above which an env of a model whose rows can exceed 256 is swept in 64-row windows, one env per wavefront (S24D: 9 % of the envs, the ones
                           // a cohort's step waits for).  A multiple of 16: the threshold must not split a window count of the 16-row form (193 - 208 rows = 13 windows —
                           // at 200 half of that class stays the slowest wavefront AND the other half holds SIMDs of its own).  Round 6, with the 32-row section off for
                           // these models (it held 593 two-env wavefronts of 97 - 128-row envs: SIMD time the 64-row form of the slowest envs can use better): 176 / 184 /
                           // 192 / 200 / 208 rows 5.86 / 5.98 / 6.18 / 5.88 / 6.00 M on two cohorts, 192 on three cohorts 6.28 M (round 5, section on: 208 rows, 5.85 M)
#ifndef WN64_NW
#define WN64_NW 3       // 64-row windows of an env that are register-resident (192 rows) ...
#endif
#ifndef WN64_NT
#define WN64_NT 2       // ... and the ones behind them whose tiles live in LDS (320 rows in all; they need the launch's LDS tier: 2 x 16 KB)
#endif

// Assemble launch (mjh_step_kernel with PH_PRE, free-body instance): the constraint blocks of this env -> window rows in global memory.
// blki / blkf / J: the block tables in LDS (step_kernel.h); sinv = M^-1/2 per dof.  Returns the number of rows (0: too many, not written).
// `clamp` (the assemble-only kernel instance, which has no sweep of its own to fall back to): blocks whose rows do not fit the model's window capacity (16 maxw) are dropped whole,
// like contacts beyond maxcon (the caller raises the capacity flag), instead of handing nothing over.
DEV int window_emit(float* __restrict__ wb, const int nvt, const int maxw, const int* blki, const float* blkf, const float* J, const float* sinv, const int nblk, const int lane, const bool clamp = false) {
  const int4* blki4 = (const int4*)blki;
  constexpr int nk = WN_NK; (void)nvt;
  // rows of every block: blocks in chunks of 64 (one per lane), the running row count carried from chunk to chunk
  int total = 0;
  for (int b0 = 0; b0 < nblk; b0 += 64) {
    const int b = b0 + lane;
    const int myn = b < nblk ? (blki4[b].x >> 4) & 15 : 0;
    total += wave_last_i(wave_incl_scan_i(myn, lane));
  }
  int nrow = total;
  if (nrow > 16 * maxw) {
    if (!clamp) return 0;
    // keep whole blocks only: the longest prefix of blocks whose rows fit (a friction pyramid cut in the middle would leave a net
    // tangential force — ADVICE r05); the rows behind it are dropped like contacts beyond maxcon
    int kept = 0, cum0 = 0;
    for (int b0 = 0; b0 < nblk; b0 += 64) {
      const int b = b0 + lane;
      const int myn = b < nblk ? (blki4[b].x >> 4) & 15 : 0;
      const int incl = wave_incl_scan_i(myn, lane);
      int cand = (cum0 + incl <= 16 * maxw) ? cum0 + incl : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) cand = max(cand, __shfl_xor(cand, o));
      kept = max(kept, cand);
      cum0 += wave_last_i(incl);
    }
    nrow = kept;
  }
  const int nwin = (nrow + 15) >> 4;
  // padding rows of the last window: zeros (inert: AR_qq = 0 -> -1 / AR_qq stored as 0)
  for (int t = lane; t < (16 * nwin - nrow) * nk; t += 64) {
    const int r = nrow + t / nk, k = t - (t / nk) * nk;
    wb[WN_ROWS + ((r >> 4) * nk + k) * 16 + (r & 15)] = 0.0f;
  }
  int base = 0;
  for (int b0 = 0; b0 < nblk && base < nrow; b0 += 64) {
    const int b = b0 + lane;
    int4 hd = make_int4(0, 0, 0, 0);
    if (b < nblk) hd = blki4[b];
    const int myn = b < nblk ? (hd.x >> 4) & 15 : 0;
    const int incl = wave_incl_scan_i(myn, lane);
    if (b < nblk) {
      const int a1 = hd.z & 0xffff, a2 = hd.w & 0xffff; const bool two = (hd.w >> 16) != 0, single = myn == 1;
      const float* Jb = J + BLK_JOFF(hd.x);
      float4 jb[12];
#pragma unroll
      for (int k = 0; k < 12; k++) jb[k] = (k < 6 || two) ? *(const float4*)(Jb + 4 * k) : make_float4(0, 0, 0, 0);
      float sc[12];
#pragma unroll
      for (int k = 0; k < 12; k++) sc[k] = (k < 6 || two) ? sinv[k < 6 ? a1 + k : a2 + k - 6] : 0.0f;
      const float* bf = blkf + b * BLKF_STRIDE;
      const int row0 = base + incl - myn;
      for (int r = 0; r < myn; r++) {
        const int g = row0 + r;
        if (g >= nrow) break;
        float* o = wb + WN_ROWS + (g >> 4) * nk * 16 + (g & 15);
        const int kk = 1 + (r >> 1); const float c = (r & 1) ? -1.0f : 1.0f;
#pragma unroll
        for (int k = 0; k < 12; k++) {
          const float v = single ? jb[k].x : jb[k].x + c * (kk == 1 ? jb[k].y : (kk == 2 ? jb[k].z : jb[k].w));
          o[16 * k] = (k < 6 || two) ? v * sc[k] : 0.0f;
        }
        o[16 * 12] = __int_as_float(a1 / 6); o[16 * 13] = __int_as_float(two ? a2 / 6 : -1);
        o[16 * 14] = single ? bf[BF_AREF] : bf[BF_AREF] + c * bf[BF_AREF + kk];
        o[16 * 15] = bf[0];
      }
    }
    base += wave_last_i(incl);
  }
  return nrow;
}

// (the sweep kernel itself: window_kernel.h, compiled in a translation unit of its own — window.hip)
