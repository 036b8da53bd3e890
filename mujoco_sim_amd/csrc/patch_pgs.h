// patch_pgs.h — contact-patch Gauss-Seidel for small free-body models (mj_solPGS and the warm start of mj_fwdConstraint, reached
// from mj_step2, /root/reference/src/mj_main.cpp:108).  Used by the LDS-resident kernels of models whose trees are all single free
// bodies (DModel::patch; BASELINE's 24-DoF scene is one).  HISTORY.md §4c.
//
// A patch = up to 16 constraint rows (whole contacts) between the SAME one or two bodies.  Its rows sit one per lane on a
// 16-lane row of the wavefront, so a wave updates up to four mutually independent patches per step.  The coupling of the
// rows inside a patch is precomputed (AR = J M^-1 J^T + R, 16 x 16, strictly lower triangle).  Each lane carries
// t_q = -res_q / AR_qq, and one Gauss-Seidel row update is two instructions for all lanes,
//     delta = max(t, -f)            t_q <- t_q + (-AR_qr / AR_qq) delta_r        (v_fmac_f32_dpp row_newbcast:r)
// instead of a cross-lane reduction per contact: a lane's t only ever receives the updates of EARLIER rows (the later columns
// of its AR row are stored as zeros), so once row q has been visited lane q keeps recomputing the same delta_q and the value
// left after the last row is the Gauss-Seidel update of every row.  The patches talk to each other through the running
// acceleration only (matrix-free, as before): u = J a before the rows, a += M^-1 J^T delta after them.
//
// Everything is kept in the scaled coordinates  a^ = M^1/2 a,  J^ = J M^-1/2  (M is diagonal here), so that neither product
// needs 1/M_dd:  u = J^ a^,  a^ += J^T delta.
//
// Visiting order.  DEFAULT (row_order 1): mj_solPGS's own constraint-row order.  A patch = a maximal run of CONSECUTIVE contacts (in
// constraint order) of one body pair with at most 16 rows, rows in order inside it; the patches are list-scheduled in that order: a
// patch goes to the first step (with a free 16-lane row) after every EARLIER patch that shares a body with it.  Two patches without
// a common body touch disjoint entries of a^, so their updates commute exactly: every reordering that keeps the relative order of
// conflicting patches produces the iterates of the sequential row-order sweep BIT FOR BIT (row_order 2 runs that sequential sweep —
// one patch per step — for the test that asserts it; the convergence test sums fixed-point integers for the same reason).
// LEGACY (row_order 0, mjh_set_pgs_row_order(0); oracle: orc_set_pgs_row_order(0)): contacts sorted by (couples two bodies first,
// body pair, constraint order); a step = a patch plus up to three later unvisited patches of the sequence that share no body with
// the step (first fit) — fewer steps per sweep (S24: 3.9 against 5.7), but not the reference's iteration.
//
// patch_build (regroup the contact blocks, schedule, fill the pool) -> patch_warmstart -> patch_sweep.
#pragma once

// descriptor of a patch: pool offset / 4 | (rows / 4) << 13 | first dof of body A << 16 | first dof of body B (63: none) << 21
// (in the schedule table every slot of a step also carries the rows / 4 of the step's longest patch << 27)
#define PD_OFF(d) (((d) & 0x1fff) << 2)
#define PD_N4(d) (((d) >> 13) & 7)
#define PD_DA(d) (((d) >> 16) & 31)
#define PD_DB(d) (((d) >> 21) & 63)
#define PD_STEPN4(d) (((d) >> 27) & 7)
#define PD_STEPONE(d) (((d) >> 30) & 1)   // every patch of the step is on one body: 6 dofs instead of 12 in the two products
// layout of a patch of nr4 rows inside the pool (floats): one record per row, then the 4 x 4 tiles (i, c <= i) of the strictly lower
// triangle of -AR_qr / AR_qq, [tile][q & 3][4].  Record of a patch between two bodies, 20 floats:  f, aref, R, 1/AR_qq | J^ [12] |
// AR_qq / 2, pad [3];  of a patch on one body (floor / wall contacts: most rows of a pile), 12 floats:  f, aref, R, 1/AR_qq | J^ [6] |
// AR_qq / 2, pad.  A lane reads its record off one address; strides 20 and 12 keep 16 lanes' ds_read_b128 on distinct banks.
#define PP_REC(two) ((two) ? 20 : 12)
#define PP_HALF(two) ((two) ? 16 : 10)
#define PP_TILES(two, nr4) (PP_REC(two) * (nr4))
#define PP_SIZE(two, n4) (4 * PP_REC(two) * (n4) + 8 * (n4) * ((n4) + 1))
#define PP_ZERO 20     // floats of zeros a lane outside a patch reads instead of a record
#ifndef PP_NSU
#define PP_NSU 6
#endif                 // schedules of up to this many steps run with their per-lane addresses cached in registers
#ifndef PP_NRC
#define PP_NRC 6
#endif                 // ... and the row records (J^, f, row constants) of the first PP_NRC steps stay in registers over the sweeps

// acc += y * (x of lane R of this lane's 16-lane row); x must not have been written by the VALU in the two instructions before
#define PP_FMAC_BC(acc, x, y, R) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #R " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y))
#define PP_BC12(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11)
#define PP_BC16(M) PP_BC12(M) M(12) M(13) M(14) M(15)

struct PatchArgs {
  float* lds;                    // dynamic LDS base
  int pool, pool_floats, pdesc, pslot, zero, ahat;   // float offsets: pool, the two live tables, 4 zero floats, the acceleration [nv]
  const int* blki; const float* blkf; const float* J; const float* qLDinv;   // block tables (inside the pool span: consumed first); qLDinv: M^-1/2 per dof
  int nblk, nv, maxcon;
  int row_order;                 // 0: legacy patch order (first fit), 1: constraint order, list-scheduled, 2: constraint order, one patch per step
};

// Groups the contact blocks into patches, schedules the patches into steps, and converts the blocks' base rows / parameters /
// forces into the pool (which overwrites them: everything is staged through registers first).  Returns the number of steps.
// swork: cost of one sweep in units of about four instructions (launch-order hint): 24 per step + 1 per row of its longest patch.
DEV int patch_build(const PatchArgs& A, const int lane, int& flags, int& swork, int& npatch_out) {
  const int nblk = A.nblk, maxcon = A.maxcon;
  int* s_word = (int*)(A.lds + A.pool);       // [maxcon] block words in patch order
  int* s_pinfo = s_word + maxcon;             // [maxcon] per patch
  int* s_rowmap = s_pinfo + maxcon;           // [min(256, 9 maxcon)] pool row -> block | row << 6 | patch << 10 | row in patch << 16 (bit 30: padding row)
  int* s_pdesc = (int*)(A.lds + A.pdesc);
  int* s_pslot = (int*)(A.lds + A.pslot);
  const int4* blki4 = (const int4*)A.blki;
  // ---- patch order: (single-body contacts last, body pair, constraint order)
  int word = 0x7fffffff;
  if (lane < nblk) {
    const int4 hd = blki4[lane];
    const int a1 = hd.z & 0xffff, a2 = hd.w & 0xffff; const bool two = (hd.w >> 16) != 0;
    const int dA = two ? min(a1, a2) : a1, dB = two ? max(a1, a2) : 63;
    word = ((((two ? 0 : 1) << 11) | (dA << 6) | dB) << 10) | (lane << 4) | ((hd.x >> 4) & 15);
  }
  int rank = lane;               // constraint order: the blocks as they were made
  if (!A.row_order) { rank = 0; for (int i = 0; i < nblk; i++) rank += __builtin_amdgcn_readlane(word, i) < word; }
  if (lane < nblk) s_word[rank] = word;
  WSYNC();
  const int w = lane < nblk ? s_word[lane] : 0;   // lanes = positions of the sequence from here on
  int np = -1, rows = 0, ck = -1, mypatch = -1, myrow0 = 0;
  for (int p = 0; p < nblk; p++) {
    const int wp = __builtin_amdgcn_readlane(w, p), k = wp >> 10, n = wp & 15;
    if (k != ck || rows + n > 16) { np++; rows = 0; ck = k; }
    if (lane == p) { mypatch = np; myrow0 = rows; }
    rows += n;
  }
  int npatch = np + 1;
  const int myn = w & 15;
  {
    const int nxt = __shfl_down(mypatch, 1);
    if (lane < nblk && (lane == nblk - 1 || nxt != mypatch)) s_pinfo[mypatch] = (myrow0 + myn) | ((w >> 10) << 8);
  }
  WSYNC();
  // ---- lanes = patches: rows, pool offsets, descriptors
  const int pinfo = lane < npatch ? s_pinfo[lane] : 0;
  const int pnr = pinfo & 255, pn4 = (pnr + 3) >> 2, pkey = pinfo >> 8;
  const int pdA = (pkey >> 6) & 31, pdB = pkey & 63;
  const int psize = lane < npatch ? PP_SIZE(pdB != 63, pn4) : 0;
  const int pend = wave_incl_scan_i(psize, lane);
  const int prowend = wave_incl_scan_i(lane < npatch ? 4 * pn4 : 0, lane);
  {   // patches that do not fit the pool (or the 256 rows staged below) are dropped with their contacts: capacity flag
    const unsigned long long ok = __ballot(lane < npatch && pend <= A.pool_floats && prowend <= 256);
    const int nfit = __popcll(ok);
    if (nfit < npatch) { flags |= 2; npatch = nfit; }
  }
  const int desc = lane < npatch ? (((pend - psize) >> 2) | (pn4 << 13) | (pdA << 16) | (pdB << 21)) : 0;
  const int tmask = lane < npatch ? ((1 << (pdA / 6)) | (pdB != 63 ? (1 << (pdB / 6)) : 0)) : 0;
  const int prow0 = prowend - 4 * pn4;
  const int tot4 = min(wave_last_i(prowend), 256);
  for (int g = lane; g < tot4; g += 64) s_rowmap[g] = -1;     // (rows of dropped patches stay unmapped)
  WSYNC();
  if (lane < npatch) { s_pdesc[lane] = desc; s_pinfo[lane] = prow0; }
  if (lane < npatch) for (int qq = pnr; qq < 4 * pn4; qq++) s_rowmap[prow0 + qq] = (1 << 30) | (lane << 10) | (qq << 16);   // (kept patches end within 256 rows)
  WSYNC();
  if (lane < nblk && mypatch < npatch) {
    const int base = s_pinfo[mypatch] + myrow0, b = (w >> 4) & 63;
    for (int r = 0; r < myn; r++) s_rowmap[base + r] = b | (r << 6) | (mypatch << 10) | ((myrow0 + r) << 16);
  }
  int nstep = 0;
  if (A.row_order) {
    // ---- list schedule of the constraint order: patch i goes to the first step >= (1 + the step of the last earlier patch on one of
    //      its bodies) that still has a free 16-lane row.  lastv: lane b = first step body b is free again; cntv / n4v / onev: lane s =
    //      patches, longest patch and "all on one body" of step s (at most 64 patches, so at most 64 steps).
    const int cap = A.row_order == 2 ? 1 : 4;
    int lastv = 0, cntv = 0, n4v = 0, onev = 1;
    for (int i = 0; i < npatch; i++) {
      const int di = __builtin_amdgcn_readlane(desc, i);
      const int bA = PD_DA(di) / 6, bB = PD_DB(di) == 63 ? bA : PD_DB(di) / 6;
      const int e = max(__builtin_amdgcn_readlane(lastv, bA), __builtin_amdgcn_readlane(lastv, bB));
      const unsigned long long bal = __ballot(cntv < cap && lane >= e);
      const int st = __ffsll((long long)bal) - 1;              // (steps <= patches <= 64: a free step always exists)
      const int c = __builtin_amdgcn_readlane(cntv, st);
      if (lane == 0) s_pslot[4 * st + c] = di;
      if (lane == st) { cntv++; n4v = max(n4v, PD_N4(di)); onev &= PD_DB(di) == 63; }
      if (lane == bA || lane == bB) lastv = st + 1;
      nstep = max(nstep, st + 1);
    }
    WSYNC();
    if (lane < nstep) {
      for (int c = 0; c < 4; c++) { const int d = c < cntv ? s_pslot[4 * lane + c] : 0; s_pslot[4 * lane + c] = d | (n4v << 27) | (onev << 30); }
    }
    swork += wave_sum_dpp_i(lane < nstep ? 24 + 4 * n4v : 0);
  } else
  // ---- legacy steps: a patch plus up to three later unvisited patches that share no body with the step
  {
    unsigned long long used = 0;
    for (int i = 0; i < npatch; i++) {
      if ((used >> i) & 1ull) continue;
      used |= 1ull << i;
      int gmask = __builtin_amdgcn_readlane(tmask, i), cnt = 1;
      const int di = __builtin_amdgcn_readlane(desc, i);
      int n4max = PD_N4(di), allone = PD_DB(di) == 63;
      if (lane == 0) s_pslot[4 * nstep] = di;
      while (cnt < 4) {
        const bool cand = lane < npatch && lane > i && !((used >> lane) & 1ull) && !(tmask & gmask);
        const unsigned long long bal = __ballot(cand);
        if (!bal) break;
        const int q = __ffsll((long long)bal) - 1;
        used |= 1ull << q; gmask |= __builtin_amdgcn_readlane(tmask, q);
        const int dq = __builtin_amdgcn_readlane(desc, q);
        if (lane == 0) s_pslot[4 * nstep + cnt] = dq;
        n4max = max(n4max, PD_N4(dq)); allone &= PD_DB(dq) == 63;
        cnt++;
      }
      swork += 24 + 4 * n4max;
      if (lane == 0) {
        for (int c = cnt; c < 4; c++) s_pslot[4 * nstep + c] = 0;
        for (int c = 0; c < 4; c++) s_pslot[4 * nstep + c] |= (n4max << 27) | (allone << 30);
      }
      nstep++;
    }
  }
  WSYNC();
  // ---- stage every pool row in registers: the pool overwrites the tables it is built from
  float jr[4][12], pr[4][3]; int mm[4];
#pragma unroll
  for (int ps = 0; ps < 4; ps++) {
    mm[ps] = -1;
#pragma unroll
    for (int k = 0; k < 12; k++) jr[ps][k] = 0;
    pr[ps][0] = 0; pr[ps][1] = 0; pr[ps][2] = 0;
    const int g = ps * 64 + lane;
    if (ps * 64 < tot4 && g < tot4) {
      const int m = s_rowmap[g];
      mm[ps] = m;
      if (!(m & (1 << 30)) && ((m >> 10) & 63) < npatch) {
        const int b = m & 63, r = (m >> 6) & 15;
        const int4 hd = blki4[b];
        const int a1 = hd.z & 0xffff, a2 = hd.w & 0xffff; const bool two = (hd.w >> 16) != 0, sw = two && a1 > a2;
        const bool single = ((hd.x >> 4) & 15) == 1;
        const int kk = 1 + (r >> 1); const float c = (r & 1) ? -1.0f : 1.0f;
        const float* Jb = A.J + BLK_JOFF(hd.x);
        const int dA = sw ? a2 : a1, dB = sw ? a1 : a2;
#pragma unroll
        for (int kp = 0; kp < 12; kp++) {
          if (kp >= 6 && !two) continue;
          const int k = sw ? (kp < 6 ? kp + 6 : kp - 6) : kp;
          const float4 jb = *(const float4*)(Jb + 4 * k);
          const float v = single ? jb.x : jb.x + c * (kk == 1 ? jb.y : (kk == 2 ? jb.z : jb.w));
          jr[ps][kp] = v * A.qLDinv[kp < 6 ? dA + kp : dB + kp - 6];
        }
        const float* bf = A.blkf + b * BLKF_STRIDE;
        pr[ps][0] = 0.0f; pr[ps][1] = single ? bf[BF_AREF] : bf[BF_AREF] + c * bf[BF_AREF + kk]; pr[ps][2] = bf[0];   // (forces: patch_warmstart)
      }
    }
  }
  WSYNC();
  float* const pool = A.lds + A.pool;
#pragma unroll
  for (int ps = 0; ps < 4; ps++) {
    const int m = mm[ps];
    if (m < 0 || ((m >> 10) & 63) >= npatch) continue;
    const int d = s_pdesc[(m >> 10) & 63], q = (m >> 16) & 15;
    const bool two = PD_DB(d) != 63;
    float* Rq = pool + PD_OFF(d) + PP_REC(two) * q;
    *(float4*)(Rq) = make_float4(pr[ps][0], pr[ps][1], pr[ps][2], 0.0f);
    *(float4*)(Rq + 4) = make_float4(jr[ps][0], jr[ps][1], jr[ps][2], jr[ps][3]);
    if (two) {
      *(float4*)(Rq + 8) = make_float4(jr[ps][4], jr[ps][5], jr[ps][6], jr[ps][7]);
      *(float4*)(Rq + 12) = make_float4(jr[ps][8], jr[ps][9], jr[ps][10], jr[ps][11]);
      *(float4*)(Rq + 16) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    } else *(float4*)(Rq + 8) = make_float4(jr[ps][4], jr[ps][5], 0.0f, 0.0f);
  }
  WSYNC();
  // ---- AR = J^ J^T + R of every patch, four patches at a time (lanes = rows): strictly lower triangle in 4 x 4 tiles
  const int rho = lane >> 4, q = lane & 15, ti = q >> 2;
  const float* zero = A.lds + A.zero;
  for (int p0 = 0; p0 < npatch; p0 += 4) {
    const int pt = p0 + rho;
    const int d = pt < npatch ? s_pdesc[pt] : 0, nr4 = PD_N4(d) << 2;
    const bool on = q < nr4;
    const bool two = PD_DB(d) != 63;
    float* P = pool + PD_OFF(d);
    float* Rq = P + PP_REC(two) * q;
    const float* Rz = on ? Rq : zero;
    const float4 j0 = *(const float4*)(Rz + 4);
    float4 j1 = *(const float4*)(Rz + 8);
    const float4 j2 = *(const float4*)((on & two) ? Rq + 12 : zero);
    if (!two) { j1.z = 0; j1.w = 0; }
    const float R = Rz[2];
    float Jv[12] = {j0.x, j0.y, j0.z, j0.w, j1.x, j1.y, j1.z, j1.w, j2.x, j2.y, j2.z, j2.w};
    asm volatile("" : "+v"(Jv[0]), "+v"(Jv[1]), "+v"(Jv[2]), "+v"(Jv[3]), "+v"(Jv[4]), "+v"(Jv[5]));
    asm volatile("" : "+v"(Jv[6]), "+v"(Jv[7]), "+v"(Jv[8]), "+v"(Jv[9]), "+v"(Jv[10]), "+v"(Jv[11]));
    int nmax = max(max(__builtin_amdgcn_readlane(nr4, 0), __builtin_amdgcn_readlane(nr4, 16)), max(__builtin_amdgcn_readlane(nr4, 32), __builtin_amdgcn_readlane(nr4, 48)));
    float acc[16];
#pragma unroll
    for (int s = 0; s < 16; s++) acc[s] = 0;
    asm volatile("s_nop 1");
#define PP_ACC(s) if (s < nmax) { PP_FMAC_BC(acc[s], Jv[0], Jv[0], s); PP_FMAC_BC(acc[s], Jv[1], Jv[1], s); PP_FMAC_BC(acc[s], Jv[2], Jv[2], s); PP_FMAC_BC(acc[s], Jv[3], Jv[3], s); \
                                PP_FMAC_BC(acc[s], Jv[4], Jv[4], s); PP_FMAC_BC(acc[s], Jv[5], Jv[5], s); PP_FMAC_BC(acc[s], Jv[6], Jv[6], s); PP_FMAC_BC(acc[s], Jv[7], Jv[7], s); \
                                PP_FMAC_BC(acc[s], Jv[8], Jv[8], s); PP_FMAC_BC(acc[s], Jv[9], Jv[9], s); PP_FMAC_BC(acc[s], Jv[10], Jv[10], s); PP_FMAC_BC(acc[s], Jv[11], Jv[11], s); }
    PP_BC16(PP_ACC)
#undef PP_ACC
    float diag = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) diag += Jv[k] * Jv[k];
    const float ARqq = diag + R;
    if (on) {
      const float inv = ARqq < MJ_MINVAL ? 0.0f : 1.0f / ARqq, ninv = -inv;
      Rq[3] = inv;
      Rq[PP_HALF(two)] = 0.5f * ARqq;
      // the tiles hold -AR_qr / AR_qq: the sweep carries t_q = -res_q / AR_qq instead of the residual itself
      float* T = P + PP_TILES(two, nr4) + ((ti * (ti + 1)) >> 1) * 16 + (q & 3) * 4;
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (c <= ti) *(float4*)(T + 16 * c) = make_float4(4*c < q ? ninv * acc[4*c] : 0.0f, 4*c + 1 < q ? ninv * acc[4*c + 1] : 0.0f,
                                                          4*c + 2 < q ? ninv * acc[4*c + 2] : 0.0f, 4*c + 3 < q ? ninv * acc[4*c + 3] : 0.0f);
    }
  }
  WSYNC();
  npatch_out = npatch;
  return nstep;
}

// four rows r0 .. r0+3 of the sweep, on t_q = -res_q / AR_qq:  delta = max(t, -f),  t_q += (-AR_qr / AR_qq) delta_r
#ifdef MJH_NO_ROW_NOP      // (experiment: the row chain without the two wait states of the DPP read — tools/r04_nop.sh)
#define PP_ROW_NOP ""
#else
#define PP_ROW_NOP "s_nop 1\n\t"
#endif
#define PP_ROW(r, ar) "v_max_f32 %[d], %[t], %[nf]\n\t" PP_ROW_NOP "v_fmac_f32_dpp %[t], %[d], " ar " row_newbcast:" #r " row_mask:0xf bank_mask:0xf\n\t"
#define PP_ROWS4(r0, r1, r2, r3, T) asm volatile(PP_ROW(r0, "%[a0]") PP_ROW(r1, "%[a1]") PP_ROW(r2, "%[a2]") PP_ROW(r3, "%[a3]") \
    : [t] "+v"(tt), [d] "=&v"(dl) : [nf] "v"(nf), [a0] "v"(T.x), [a1] "v"(T.y), [a2] "v"(T.z), [a3] "v"(T.w))

// dst = src + (src of the lane `ror` places down the row), written in the quads of bank mask bm only
#define PP_ADD_ROR(dst, src, ror, bm) "v_add_f32_dpp " dst ", " src ", " src " row_ror:" #ror " row_mask:0xf bank_mask:" #bm "\n\t"
#define PP_ADD_QP(dst, a, b, c, d) "v_add_f32_dpp " dst ", " dst ", " dst " quad_perm:[" #a "," #b "," #c "," #d "] row_mask:0xf bank_mask:0xf\n\t"

// sum_k J_k * (al of lane k of this lane's 16-lane row), k = 0..11: a row of J^ times the dof vector held by lanes 0..11
DEV float pp_dot12(const float al, const float4& J0, const float4& J1, const float4& J2) {
  float u;
  asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf"
               : "=&v"(u) : "v"(al), "v"(J0.x), "v"(J0.y), "v"(J0.z), "v"(J0.w), "v"(J1.x), "v"(J1.y), "v"(J1.z), "v"(J1.w),
                 "v"(J2.x), "v"(J2.y), "v"(J2.z), "v"(J2.w));
  return u;
}
// J^T x over the 16 lanes of a row: 12 sums, folded while they are reduced — across the quads first (the DPP bank mask picks which
// quads keep which half), then inside the quads; every lane of quad j ends up with the sums of dofs 3j .. 3j+2 in b[0..2].
DEV void pp_jt(const float4& J0, const float4& J1, const float4& J2, const float x, float* b) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f d2 = {x, x};
  const v2f p01 = v2f{J0.x, J0.y} * d2, p23 = v2f{J0.z, J0.w} * d2, p45 = v2f{J1.x, J1.y} * d2;
  const v2f p67 = v2f{J1.z, J1.w} * d2, p89 = v2f{J2.x, J2.y} * d2, pab = v2f{J2.z, J2.w} * d2;
  float r0, r1, r2, r3, r4, r5;
  asm volatile("s_nop 1\n\t"
               PP_ADD_ROR("%0", "%6", 8, 0x3) PP_ADD_ROR("%1", "%7", 8, 0x3) PP_ADD_ROR("%2", "%8", 8, 0x3)
               PP_ADD_ROR("%3", "%9", 8, 0x3) PP_ADD_ROR("%4", "%10", 8, 0x3) PP_ADD_ROR("%5", "%11", 8, 0x3)
               PP_ADD_ROR("%0", "%12", 8, 0xc) PP_ADD_ROR("%1", "%13", 8, 0xc) PP_ADD_ROR("%2", "%14", 8, 0xc)
               PP_ADD_ROR("%3", "%15", 8, 0xc) PP_ADD_ROR("%4", "%16", 8, 0xc) PP_ADD_ROR("%5", "%17", 8, 0xc)
               : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5)
               : "v"(p01.x), "v"(p01.y), "v"(p23.x), "v"(p23.y), "v"(p45.x), "v"(p45.y), "v"(p67.x), "v"(p67.y), "v"(p89.x), "v"(p89.y), "v"(pab.x), "v"(pab.y));
  asm volatile("s_nop 1\n\t"
               PP_ADD_ROR("%0", "%3", 12, 0x5) PP_ADD_ROR("%1", "%4", 12, 0x5) PP_ADD_ROR("%2", "%5", 12, 0x5)
               PP_ADD_ROR("%0", "%6", 4, 0xa) PP_ADD_ROR("%1", "%7", 4, 0xa) PP_ADD_ROR("%2", "%8", 4, 0xa)
               PP_ADD_QP("%0", 1, 0, 3, 2) PP_ADD_QP("%1", 1, 0, 3, 2) PP_ADD_QP("%2", 1, 0, 3, 2)
               PP_ADD_QP("%0", 2, 3, 0, 1) PP_ADD_QP("%1", 2, 3, 0, 1) PP_ADD_QP("%2", 2, 3, 0, 1)
               : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5));
}

// the same two products for a step whose patches all sit on ONE body: 6 dofs
DEV float pp_dot6(const float al, const float4& J0, const float4& J1) {
  float u;
  asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f32_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf"
               : "=&v"(u) : "v"(al), "v"(J0.x), "v"(J0.y), "v"(J0.z), "v"(J0.w), "v"(J1.x), "v"(J1.y));
  return u;
}
// 6 sums over the 16 lanes: the even quads end up with dofs 0..2, the odd quads with dofs 3..5
DEV void pp_jt6(const float4& J0, const float4& J1, const float x, float* b) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f d2 = {x, x};
  const v2f p01 = v2f{J0.x, J0.y} * d2, p23 = v2f{J0.z, J0.w} * d2, p45 = v2f{J1.x, J1.y} * d2;
  asm volatile("s_nop 1\n\t"
               PP_ADD_ROR("%0", "%3", 12, 0x5) PP_ADD_ROR("%1", "%4", 12, 0x5) PP_ADD_ROR("%2", "%5", 12, 0x5)
               PP_ADD_ROR("%0", "%6", 4, 0xa) PP_ADD_ROR("%1", "%7", 4, 0xa) PP_ADD_ROR("%2", "%8", 4, 0xa)
               "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               PP_ADD_QP("%0", 1, 0, 3, 2) PP_ADD_QP("%1", 1, 0, 3, 2) PP_ADD_QP("%2", 1, 0, 3, 2)
               PP_ADD_QP("%0", 2, 3, 0, 1) PP_ADD_QP("%1", 2, 3, 0, 1) PP_ADD_QP("%2", 2, 3, 0, 1)
               : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]) : "v"(p01.x), "v"(p01.y), "v"(p23.x), "v"(p23.y), "v"(p45.x), "v"(p45.y));
}

// Warm start in patch form (mj_fwdConstraint): the forces implied by qacc_warmstart, f = max(0, -(J a_ws - aref) / R), kept if
// their dual cost  sum f (1/2 (J da + R f) + J a_smooth - aref),  da = M^-1 J^T f,  is not positive.  what / ashat: M^1/2 qacc_warmstart
// and M^1/2 qacc_smooth (LDS vectors, float offsets); dahat: zeroed on entry, M^1/2 da on exit (zero again if the forces were dropped).
DEV void patch_warmstart(const PatchArgs& A, const int lane, const int nstep, const int npatch, const int what, const int ashat, const int dahat) {
  const int rho = lane >> 4, q = lane & 15;
  const int* s_pslot = (const int*)(A.lds + A.pslot) + rho;
  const int* s_pdesc = (const int*)(A.lds + A.pdesc);
  float* const pool = A.lds + A.pool;
  float* const zero = A.lds + A.zero;
  const int recoff2 = PP_REC(true) * q, recoff1 = PP_REC(false) * q;
  const int gq = q < 6 ? q : q - 6; const bool gA = q < 6, gB = q >= 6 && q < 12;
  const int addoff = (q & 4) ? 3 : 0; const bool addB = q >= 8, adder = (q & 3) == 0;
  struct Row { float4 P, J0, J1, J2; float* rec; int gidx, aidx; };     // gidx: dof this lane carries (-1: none), aidx: first dof it adds to
  auto rowd = [&](const int d) __attribute__((always_inline)) {
    Row o;
    const int nr4 = PD_N4(d) << 2, dA = PD_DA(d), dB = PD_DB(d);
    const bool on = q < nr4, hasB = dB != 63;
    o.rec = on ? pool + PD_OFF(d) + (hasB ? recoff2 : recoff1) : zero;
    const float* rec2 = (on & hasB) ? o.rec : zero;
    o.P = *(const float4*)(o.rec); o.J0 = *(const float4*)(o.rec + 4); o.J1 = *(const float4*)(o.rec + 8); o.J2 = *(const float4*)(rec2 + 12);
    if (!hasB) { o.J1.z = 0; o.J1.w = 0; }
    o.gidx = (gA | (gB & hasB)) ? (gA ? dA : dB) + gq : -1;
    o.aidx = ((addB & hasB) ? dB : dA) + addoff;
    return o;
  };
  auto row = [&](const int t) __attribute__((always_inline)) { return rowd(s_pslot[4 * t]); };
  auto gather = [&](const int vec, const int gidx) __attribute__((always_inline)) { return *(gidx >= 0 ? A.lds + vec + gidx : zero); };
  for (int t = 0; t < nstep; t++) {
    Row o = row(t);
    const float jar = pp_dot12(gather(what, o.gidx), o.J0, o.J1, o.J2) - o.P.y;
    const float f = (jar < 0.0f && o.P.z > 0.0f) ? -jar / o.P.z : 0.0f;
    o.rec[0] = f;
    float b[3];
    pp_jt(o.J0, o.J1, o.J2, f, b);
    if (adder) { float* pd = A.lds + dahat + o.aidx; atomicAdd(pd, b[0]); atomicAdd(pd + 1, b[1]); atomicAdd(pd + 2, b[2]); }
  }
  WSYNC();
  // the dual cost: summed over the patches in THEIR order, four at a time (not over the steps: the sum, and with it the decision,
  // must not depend on the schedule)
  float cost = 0;
  for (int p0 = 0; p0 < npatch; p0 += 4) {
    Row o = rowd(p0 + rho < npatch ? s_pdesc[p0 + rho] : 0);
    const float jda = pp_dot12(gather(dahat, o.gidx), o.J0, o.J1, o.J2);
    const float bb = pp_dot12(gather(ashat, o.gidx), o.J0, o.J1, o.J2) - o.P.y;
    const float f = o.P.x;
    cost += f * (0.5f * (jda + o.P.z * f) + bb);
  }
  cost = wave_sum<4>(cost);
  if (cost > 0.0f) {
    for (int t = 0; t < nstep; t++) { Row o = row(t); o.rec[0] = 0.0f; }
    for (int d = lane; d < A.nv; d += 64) A.lds[dahat + d] = 0.0f;
  }
  WSYNC();
}

struct PatchOps { float4 J0, J1, J2, P, A0, A1, A2, A3; float half; float* rec; const float* pa; float* padd; int nr4, stepone; };   // nr4: rows of the STEP's longest patch

// The sweeps.  A.ahat holds a^ = M^1/2 a on entry and on exit.  Returns the number of sweeps.
DEV int patch_sweep(const PatchArgs& A, const int lane, const int nstep, const int itmax, const float tol, const float scale) {
  const int rho = lane >> 4, q = lane & 15, ti = q >> 2;
  const int* s_pslot = (const int*)(A.lds + A.pslot) + rho;
  float* const pool = A.lds + A.pool;
  float* const zero = A.lds + A.zero;
  float* const ahat = A.lds + A.ahat;
  // per-lane constants of the operand addresses
  const int recoff2 = PP_REC(true) * q, recoff1 = PP_REC(false) * q, tileoff = ((ti * (ti + 1)) >> 1) * 16 + (q & 3) * 4;
  const int gq = q < 6 ? q : q - 6;                        // u = J^ a^: lanes 0..5 carry the dofs of body A, 6..11 those of body B
  const bool gA = q < 6, gB = q >= 6 && q < 12;
  const int addoff = (q & 4) ? 3 : 0;                      // a^ += J^T delta: the first lane of quad j adds dofs 3j .. 3j+2
  const bool addB = q >= 8, adder = (q & 3) == 0;
  auto load = [&](const int d) __attribute__((always_inline)) {
    PatchOps o;
    const int nr4 = PD_N4(d) << 2, dA = PD_DA(d), dB = PD_DB(d);
    const bool on = q < nr4, hasB = dB != 63;
    float* P = pool + PD_OFF(d);
    o.rec = on ? P + (hasB ? recoff2 : recoff1) : zero;
    const float* rec2 = (on & hasB) ? o.rec : zero;          // the second half of a two-body record
    o.P = *(const float4*)(o.rec); o.J0 = *(const float4*)(o.rec + 4); o.J1 = *(const float4*)(o.rec + 8);
    o.J2 = *(const float4*)(rec2 + 12);
    const float h2 = rec2[16];
    o.half = hasB ? h2 : o.J1.z;
    if (!hasB) { o.J1.z = 0; o.J1.w = 0; }
    const float* T = P + PP_TILES(hasB, nr4) + tileoff;
    o.A0 = *(const float4*)(on ? T : zero);
    o.A1 = *(const float4*)((on & (ti >= 1)) ? T + 16 : zero);
    o.A2 = *(const float4*)((on & (ti >= 2)) ? T + 32 : zero);
    o.A3 = *(const float4*)((on & (ti >= 3)) ? T + 48 : zero);
    o.pa = (gA | (gB & hasB)) ? ahat + (gA ? dA : dB) + gq : zero;
    o.padd = ahat + ((addB & hasB) ? dB : dA) + addoff;      // (a patch on one body has zeros in the B half: they go to body A)
    o.nr4 = PD_STEPN4(d) << 2; o.stepone = PD_STEPONE(d);
    return o;
  };
  // one step: the four patches of the wave's rows.  al = this lane's entry of a^ (read before the next step's operands were
  // requested, so that waiting for it does not wait for them)
  // the rows of a step once every lane has its t = -res / AR_qq (tt); returns the lane's force change
  const ImpQ iq = imp_quantum(scale, tol);
  auto rows = [&](const PatchOps& o, const float4& J0, const float4& J1, const float4& J2, const float half, const bool allone,
                  const float f, float tt, int& impl) __attribute__((always_inline)) {
    const int nmax = __builtin_amdgcn_readfirstlane(o.nr4);      // rows of the step's longest patch
    const float nf = -f;
    float dl;
    PP_ROWS4(0, 1, 2, 3, o.A0);
    if (nmax > 4) {
      PP_ROWS4(4, 5, 6, 7, o.A1);
      if (nmax > 8) {
        PP_ROWS4(8, 9, 10, 11, o.A2);
        if (nmax > 12) PP_ROWS4(12, 13, 14, 15, o.A3);
      }
    }
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(dl) : "v"(tt), "v"(nf));     // every lane's own update (its t is final: header comment)
    impl += imp_fixed((half * dl) * (2.0f * tt - dl), iq.qs);     // cost decrease  -(delta res + AR_qq delta^2 / 2),  res = -t AR_qq  (fixed point: dev_math.h)
    // a^ += J^T delta: the first lane of quad j adds dofs 3j .. 3j+2
    float b[3];
    if (allone) pp_jt6(J0, J1, dl, b); else pp_jt(J0, J1, J2, dl, b);
    if (adder && (!allone || q < 8)) { atomicAdd(o.padd, b[0]); atomicAdd(o.padd + 1, b[1]); atomicAdd(o.padd + 2, b[2]); }
    return dl;
  };
  auto solve = [&](PatchOps& o, const float al, int& impl) __attribute__((always_inline)) {
    const bool allone = __builtin_amdgcn_readfirstlane(o.stepone) != 0;   // every patch of the step on one body: 6 dofs in the two products
    const float u = allone ? pp_dot6(al, o.J0, o.J1) : pp_dot12(al, o.J0, o.J1, o.J2);
    const float f = o.P.x;
    const float tt = ((u - o.P.y) + o.P.z * f) * -o.P.w;
    o.rec[0] = f + rows(o, o.J0, o.J1, o.J2, o.half, allone, f, tt, impl);
  };
  int niter = 0;
  if (nstep == 1) {
    for (int it = 0; it < itmax; it++) {
      int impl = 0;
      PatchOps o = load(s_pslot[0]);
      solve(o, *o.pa, impl);
      niter = it + 1;
      if (wave_sum_dpp_i(impl) < iq.thr) break;
    }
    return niter;
  }
  // two to PP_NSU steps (nearly every environment): the schedule does not change over the sweeps, so every lane works out the
  // eight LDS addresses a step needs ONCE, keeps them in registers (four packed words per step; records and tiles are 16-byte
  // aligned, their low bits carry the step's flags), and the step loop is unrolled over them: 10 instead of 40 address
  // instructions per step and sweep.  Same pipeline as below: operands of step t+1 requested before step t is solved.
  if (nstep <= PP_NSU) {
    typedef __attribute__((address_space(3))) float lds_float;
    auto laddr = [&](const float* pp) __attribute__((always_inline)) { return (unsigned)(unsigned long)(lds_float*)pp; };
    auto lptr = [&](const unsigned a) __attribute__((always_inline)) { return (float*)(lds_float*)(unsigned long)a; };
    unsigned c0[PP_NSU + 1], c1[PP_NSU + 1], c2[PP_NSU + 1], c3[PP_NSU + 1];   // entry nstep (and every later one) = step 0: the step after the sweep's last
#pragma unroll
    for (int t = 0; t <= PP_NSU; t++) {
      const int d = s_pslot[4 * (t < nstep ? t : 0)];
      const int nr4 = PD_N4(d) << 2, dA = PD_DA(d), dB = PD_DB(d);
      const bool on = q < nr4, hasB = dB != 63;
      float* P = pool + PD_OFF(d);
      float* rec = on ? P + (hasB ? recoff2 : recoff1) : zero;
      const float* rec2 = (on & hasB) ? rec : zero;
      const float* T = P + PP_TILES(hasB, nr4) + tileoff;
      const float* pa = (gA | (gB & hasB)) ? ahat + (gA ? dA : dB) + gq : zero;
      const float* padd = ahat + ((addB & hasB) ? dB : dA) + addoff;
      c0[t] = laddr(rec) | (laddr(rec2) << 16) | (hasB ? 1u : 0u);
      c1[t] = laddr(on ? T : zero) | (laddr((on & (ti >= 1)) ? T + 16 : zero) << 16) | (unsigned)(PD_STEPN4(d) - 1);
      c2[t] = laddr((on & (ti >= 2)) ? T + 32 : zero) | (laddr((on & (ti >= 3)) ? T + 48 : zero) << 16) | (unsigned)PD_STEPONE(d);
      c3[t] = laddr(pa) | (laddr(padd) << 16);
    }
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4f lds_v4f;
    auto ld4 = [&](const unsigned a) __attribute__((always_inline)) { const v4f v = *(const lds_v4f*)(unsigned long)a; return make_float4(v.x, v.y, v.z, v.w); };   // ds_read_b128
    auto loadc = [&](const unsigned a0, const unsigned a1, const unsigned a2, const unsigned a3) __attribute__((always_inline)) {
      PatchOps o;
      const bool hasB = (a0 & 1u) != 0;
      const unsigned ra = a0 & 0xfff0u, rb = a0 >> 16;
      o.rec = lptr(ra);
      o.P = ld4(ra); o.J0 = ld4(ra + 16); o.J1 = ld4(ra + 32); o.J2 = ld4(rb + 48);
      const float h2 = *(const lds_float*)(unsigned long)(rb + 64);
      o.half = hasB ? h2 : o.J1.z;
      if (!hasB) { o.J1.z = 0; o.J1.w = 0; }
      o.A0 = ld4(a1 & 0xfff0u); o.A1 = ld4(a1 >> 16); o.A2 = ld4(a2 & 0xfff0u); o.A3 = ld4(a2 >> 16);
      o.pa = lptr(a3 & 0xffffu); o.padd = lptr(a3 >> 16);
      o.nr4 = (int)((a1 & 3u) + 1u) << 2; o.stepone = (int)(a2 & 1u);
      return o;
    };
    // The row records of the first PP_NRC steps do not change over the sweeps either (only the force does, and only its own lane
    // reads it): they stay in registers, 17 per step; the forces go
    // back to the records after the last sweep.  Per sweep such a step then fetches its tiles and its entry of a^ only.
    struct RowC { float4 J0, J1, J2; float f, aref, R, nw, half; };
    RowC rc[PP_NRC];
#pragma unroll
    for (int t = 0; t < PP_NRC; t++) {
      if (t < nstep) {
        const unsigned ra = c0[t] & 0xfff0u, rb = c0[t] >> 16; const bool hasB = (c0[t] & 1u) != 0;
        const float4 P = ld4(ra);
        rc[t].J0 = ld4(ra + 16); rc[t].J1 = ld4(ra + 32); rc[t].J2 = ld4(rb + 48);
        const float h2 = *(const lds_float*)(unsigned long)(rb + 64);
        rc[t].half = hasB ? h2 : rc[t].J1.z;
        if (!hasB) { rc[t].J1.z = 0; rc[t].J1.w = 0; }
        rc[t].f = P.x; rc[t].aref = P.y; rc[t].R = P.z; rc[t].nw = -P.w;
      }
    }
    auto loadt = [&](const unsigned a0, const unsigned a1, const unsigned a2, const unsigned a3) __attribute__((always_inline)) {   // tiles + addresses only
      PatchOps o;
      o.rec = lptr(a0 & 0xfff0u);
      o.A0 = ld4(a1 & 0xfff0u); o.A1 = ld4(a1 >> 16); o.A2 = ld4(a2 & 0xfff0u); o.A3 = ld4(a2 >> 16);
      o.pa = lptr(a3 & 0xffffu); o.padd = lptr(a3 >> 16);
      o.nr4 = (int)((a1 & 3u) + 1u) << 2; o.stepone = (int)(a2 & 1u);
      return o;
    };
    // operands of step t (t = nstep: step 0 of the next sweep)
#define PP_LOADSTEP(t) ((((t) < PP_NRC && (t) < nstep) || ((t) >= nstep)) ? loadt(c0[t], c1[t], c2[t], c3[t]) : loadc(c0[t], c1[t], c2[t], c3[t]))
    PatchOps op[2];
    op[0] = loadt(c0[0], c1[0], c2[0], c3[0]);
    for (int it = 0; it < itmax; it++) {
      int impl = 0;
#pragma unroll
      for (int t = 0; t < PP_NSU; t++) {
        if (t < nstep) {
          const float al = *op[t & 1].pa;
          op[(t + 1) & 1] = PP_LOADSTEP(t + 1);
          asm volatile("" ::: "memory");     // the requests above stay above: they are consumed one step later
          if (t < PP_NRC) {
            PatchOps& o = op[t & 1];
            const bool allone = __builtin_amdgcn_readfirstlane(o.stepone) != 0;
            const float u = allone ? pp_dot6(al, rc[t].J0, rc[t].J1) : pp_dot12(al, rc[t].J0, rc[t].J1, rc[t].J2);
            const float f = rc[t].f;
            const float tt = ((u - rc[t].aref) + rc[t].R * f) * rc[t].nw;     // (same arithmetic as the uncached step)
            rc[t].f = f + rows(o, rc[t].J0, rc[t].J1, rc[t].J2, rc[t].half, allone, f, tt, impl);
          } else solve(op[t & 1], al, impl);
        }
      }
      if (nstep & 1) op[0] = op[1];          // odd step count: step 0 of the next sweep was requested into the other set
      niter = it + 1;
      if (wave_sum_dpp_i(impl) < iq.thr) break;
    }
#undef PP_LOADSTEP
#pragma unroll
    for (int t = 0; t < PP_NRC; t++) if (t < nstep) *lptr(c0[t] & 0xfff0u) = rc[t].f;
    return niter;
  }
  // more steps: software pipeline over the cyclic schedule, descriptor of step t+2 -> operands of step t+1 -> solve
  // step t (a patch is only written in its own step, so what is in flight is never stale); two operand sets in ping-pong
  auto cyc = [&](const int x) __attribute__((always_inline)) { return x >= nstep ? x - nstep : x; };
  PatchOps opA = load(s_pslot[0]), opB;
  int dn = s_pslot[4];
  for (int it = 0; it < itmax; it++) {
    int impl = 0;
    for (int t = 0; t < nstep; t += 2) {
      {
        const float al = *opA.pa;
        opB = load(dn);
        dn = s_pslot[4 * cyc(t + 2)];
        asm volatile("" ::: "memory");     // the requests above stay above: they are consumed one step later
        solve(opA, al, impl);
      }
      if (t + 1 < nstep) {
        const float al = *opB.pa;
        opA = load(dn);
        dn = s_pslot[4 * cyc(t + 3)];
        asm volatile("" ::: "memory");
        solve(opB, al, impl);
      } else opA = opB;                    // odd step count: step 0 of the next sweep was requested into B
    }
    niter = it + 1;
    if (wave_sum_dpp_i(impl) < iq.thr) break;
  }
  return niter;
}
