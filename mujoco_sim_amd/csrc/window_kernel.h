// window_kernel.h — mjh_window_kernel: the sweeps, mj_checkAcc and mj_Euler of the window chain (see window_pgs.h for the design and the
// hand-over layout).  Included by window.hip only: the kernel is a translation unit of its own (seconds to compile, against minutes
// for the step kernel's instances), launched from engine.hip through mjh_launch_window.
#pragma once
#include "step_kernel.h"

#ifdef WN_PROF_CLK
#define WN_STAT0(x) min((int)(((long long)__builtin_amdgcn_s_memtime() - wn_t0) >> 5) + 1, (1 << 22) - 1)       // (probe build: the wave's clocks where the contact count goes; the launch order keeps its own hint)
#else
#define WN_STAT0(x) (x)
#endif

#ifdef WN_PROF_CLK
__shared__ long long wn_t0;      // (probe build: the wavefront's start clock)
#endif

#define WN_BC8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
template <int NV> struct WnWin { float J[NV]; float4 A0, A1, A2, A3; float aref, R, nw, half; };   // one window row: J^, tile row (-AR_qr / AR_qq, r < q), constants

// transpose-reduce: x[k](lane q) -> lane q of the 16-lane row receives sum over the row's lanes of x[q]   (16 values, 33 instructions)
#define WN_ROR2(d, s0, s1, r0, m0, r1, m1) "v_add_f32_dpp " d ", " s0 ", " s0 " row_ror:" #r0 " row_mask:0xf bank_mask:" #m0 "\n\tv_add_f32_dpp " d ", " s1 ", " s1 " row_ror:" #r1 " row_mask:0xf bank_mask:" #m1 "\n\t"
DEV float wn_fold_tail(const float* z) {     // four values over the quads (lane bit 1 and bit 0 pick the value)
  const int q = threadIdx.x & 15;
  const bool b1 = (q & 2) != 0, b0 = (q & 1) != 0;
  float w[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const float s = b1 ? z[k + 2] : z[k], o = b1 ? z[k] : z[k + 2];
    w[k] = s + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, o), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  }
  const float s = b0 ? w[1] : w[0], o = b0 ? w[0] : w[1];
  return s + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, o), 0xB1, 0xf, 0xf, true));       // quad_perm [1,0,3,2]
}
DEV void wn_fold_8to4(const float* y, float* z) {   // lanes with bit 2 clear keep values 0..3, the others 4..7 (partner: 4 lanes away, same half of the row)
  asm volatile(WN_ROR2("%0", "%4", "%8", 12, 0x5, 4, 0xa) WN_ROR2("%1", "%5", "%9", 12, 0x5, 4, 0xa) WN_ROR2("%2", "%6", "%10", 12, 0x5, 4, 0xa) WN_ROR2("%3", "%7", "%11", 12, 0x5, 4, 0xa)
               : "=&v"(z[0]), "=&v"(z[1]), "=&v"(z[2]), "=&v"(z[3]) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]));
}
DEV float wn_fold16(const float* x) {
  float y[8], z[4];
  // lanes 0..7 keep values 0..7, lanes 8..15 values 8..15 (partner: 8 lanes away)
  asm volatile("s_nop 1\n\t"
               WN_ROR2("%0", "%8", "%16", 8, 0x3, 8, 0xc) WN_ROR2("%1", "%9", "%17", 8, 0x3, 8, 0xc) WN_ROR2("%2", "%10", "%18", 8, 0x3, 8, 0xc) WN_ROR2("%3", "%11", "%19", 8, 0x3, 8, 0xc)
               WN_ROR2("%4", "%12", "%20", 8, 0x3, 8, 0xc) WN_ROR2("%5", "%13", "%21", 8, 0x3, 8, 0xc) WN_ROR2("%6", "%14", "%22", 8, 0x3, 8, 0xc) WN_ROR2("%7", "%15", "%23", 8, 0x3, 8, 0xc)
               : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]), "=&v"(y[4]), "=&v"(y[5]), "=&v"(y[6]), "=&v"(y[7])
               : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
                 "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(x[14]), "v"(x[15]));
  wn_fold_8to4(y, z);
  return wn_fold_tail(z);
}
// 8 values: lanes 2j and 2j + 1 both receive the sum of x[j]   (16 instructions: every stage halves the values a lane keeps)
DEV float wn_fold8(const float* x) {
  float y[4], z[2];
  asm volatile("s_nop 1\n\t" WN_ROR2("%0", "%4", "%8", 8, 0x3, 8, 0xc) WN_ROR2("%1", "%5", "%9", 8, 0x3, 8, 0xc) WN_ROR2("%2", "%6", "%10", 8, 0x3, 8, 0xc) WN_ROR2("%3", "%7", "%11", 8, 0x3, 8, 0xc)
               : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));   // lanes 0..7: values 0..3, lanes 8..15: values 4..7
  asm volatile(WN_ROR2("%0", "%2", "%4", 12, 0x5, 4, 0xa) WN_ROR2("%1", "%3", "%5", 12, 0x5, 4, 0xa)
               : "=&v"(z[0]), "=&v"(z[1]) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]));                                  // lane bit 2 clear: values 0, 1 of the half; set: 2, 3
  const bool b1 = (threadIdx.x & 2) != 0;
  const float s = b1 ? z[1] : z[0], o = b1 ? z[0] : z[1];
  float r = s + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, o), 0x4E, 0xf, 0xf, true));   // value index = 4 bit3 + 2 bit2 + bit1 of the lane = lane >> 1
  MJH_DPP_ADD(r, 0xB1, 0xf, true);                                                                                             // + the neighbour lane (bit 0): both keep the value
  return r;
}
// sum over the 16 lanes of a row, result in every lane of the row (integers: order-independent)
DEV int wn_rowsum_i(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true); v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  return __builtin_amdgcn_update_dpp(0, v, 0x15F, 0xf, 0xf, false);     // row_newbcast:15
}
DEV float wn_rowsum_f(float v) {
  MJH_DPP_ADD(v, 0x111, 0xf, true); MJH_DPP_ADD(v, 0x112, 0xf, true); MJH_DPP_ADD(v, 0x114, 0xf, true); MJH_DPP_ADD(v, 0x118, 0xf, true);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x15F, 0xf, 0xf, false));
}

// u = J^ . a^ : a_lo / a_hi hold the dof vector (lane q of the row: dofs q and 16 + q); two independent chains
#define WN_FM(acc, x, y, R) "v_fmac_f32_dpp " acc ", " x ", " y " row_newbcast:" #R " row_mask:0xf bank_mask:0xf\n\t"
template <int NV> DEV float wn_dot(const float* J, const float a_lo, const float a_hi) {
  static_assert(NV == 24 || NV == 32, "window kernel instances: 24 or 32 dof slots");
  float u0, u1;
  asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
               WN_FM("%0", "%1", "%3", 1) WN_FM("%0", "%1", "%4", 2) WN_FM("%0", "%1", "%5", 3) WN_FM("%0", "%1", "%6", 4) WN_FM("%0", "%1", "%7", 5)
               WN_FM("%0", "%1", "%8", 6) WN_FM("%0", "%1", "%9", 7) WN_FM("%0", "%1", "%10", 8) WN_FM("%0", "%1", "%11", 9) WN_FM("%0", "%1", "%12", 10)
               WN_FM("%0", "%1", "%13", 11) WN_FM("%0", "%1", "%14", 12) WN_FM("%0", "%1", "%15", 13) WN_FM("%0", "%1", "%16", 14) WN_FM("%0", "%1", "%17", 15)
               : "=&v"(u0) : "v"(a_lo), "v"(J[0]), "v"(J[1]), "v"(J[2]), "v"(J[3]), "v"(J[4]), "v"(J[5]), "v"(J[6]), "v"(J[7]),
                 "v"(J[8]), "v"(J[9]), "v"(J[10]), "v"(J[11]), "v"(J[12]), "v"(J[13]), "v"(J[14]), "v"(J[15]));
  if constexpr (NV == 24)      // (a_hi: lane 2j carries dof 16 + j — wn_fold8)
    asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 WN_FM("%0", "%1", "%3", 2) WN_FM("%0", "%1", "%4", 4) WN_FM("%0", "%1", "%5", 6) WN_FM("%0", "%1", "%6", 8) WN_FM("%0", "%1", "%7", 10)
                 WN_FM("%0", "%1", "%8", 12) WN_FM("%0", "%1", "%9", 14)
                 : "=&v"(u1) : "v"(a_hi), "v"(J[16]), "v"(J[17]), "v"(J[18]), "v"(J[19]), "v"(J[20]), "v"(J[21]), "v"(J[22]), "v"(J[23]));
  else
    asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 WN_FM("%0", "%1", "%3", 1) WN_FM("%0", "%1", "%4", 2) WN_FM("%0", "%1", "%5", 3) WN_FM("%0", "%1", "%6", 4) WN_FM("%0", "%1", "%7", 5)
                 WN_FM("%0", "%1", "%8", 6) WN_FM("%0", "%1", "%9", 7)
                 : "=&v"(u1) : "v"(a_hi), "v"(J[16]), "v"(J[17]), "v"(J[18]), "v"(J[19]), "v"(J[20]), "v"(J[21]), "v"(J[22]), "v"(J[23]));
  if constexpr (NV == 32)
    asm volatile("s_nop 1\n\t" WN_FM("%0", "%1", "%2", 8) WN_FM("%0", "%1", "%3", 9) WN_FM("%0", "%1", "%4", 10) WN_FM("%0", "%1", "%5", 11)
                 WN_FM("%0", "%1", "%6", 12) WN_FM("%0", "%1", "%7", 13) WN_FM("%0", "%1", "%8", 14) WN_FM("%0", "%1", "%9", 15)
                 : "+v"(u1) : "v"(a_hi), "v"(J[NV == 32 ? 24 : 0]), "v"(J[NV == 32 ? 25 : 0]), "v"(J[NV == 32 ? 26 : 0]), "v"(J[NV == 32 ? 27 : 0]),
                   "v"(J[NV == 32 ? 28 : 0]), "v"(J[NV == 32 ? 29 : 0]), "v"(J[NV == 32 ? 30 : 0]), "v"(J[NV == 32 ? 31 : 0]));
  return u0 + u1;
}
// a^ += J^T x over the row's 16 lanes
template <int NV> DEV void wn_jt(const float* J, const float x, float& a_lo, float& a_hi) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f x2 = {x, x};
  float p[NV];
#pragma unroll
  for (int k = 0; k < NV; k += 2) { const v2f pr = v2f{J[k], J[k + 1]} * x2; p[k] = pr.x; p[k + 1] = pr.y; }     // v_pk_mul_f32
  a_lo += wn_fold16(p);
  if constexpr (NV == 32) a_hi += wn_fold16(p + 16); else a_hi += wn_fold8(p + 16);
}


// ---- 32-row windows: TWO environments per wavefront (a 32-lane half each), for the environments with many rows.  A cohort's step waits
// for its slowest wavefront, and that one carries an env at the sweep cap with 7 or 8 windows of 16 (S24: 3.5 % of the envs have more
// than 96 rows — tools/s24_critical_path.py, tools/s24_hybrid_model.py).  Per 32 rows one dot and one transpose-reduce instead of two:
// ~186 instead of 260 instructions on that env's chain.  Lanes: env slot es = lane >> 5, half hq = (lane >> 4) & 1 (rows 0..15 / 16..31
// of the window), q = lane & 15.  Gauss-Seidel stays row by row: the lower half's 16 rows (DPP row mask 0x5), then the upper half
// receives the lower deltas through the cross tile (C: -AR_{16+q, r} / AR_qq, r < 16 — the lower deltas come over by ds_swizzle),
// then its own 16 rows (row mask 0xa).  Both halves carry the same a^ (dof q / 16 + q / 2), their partial J^T sums are exchanged by
// ds_swizzle.  The MODE of an env is a function of its own row count alone (96 < rows <= 128, set by the assemble launch): results do
// not depend on which envs share a wavefront.  Same row math, same order, same stopping rule as the 16-row form; the grouping of
// the arithmetic differs (fp32 rounding).
#define WN_SWZ16(x) __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, (x)), 0x401F))   // lane ^ 16 within 32 lanes
#define WN_ROWM(r, ar, m) "v_max_f32 %[d], %[t], %[nf]\n\ts_nop 1\n\tv_fmac_f32_dpp %[t], %[d], " ar " row_newbcast:" #r " row_mask:" #m " bank_mask:0xf\n\t"
#define WN_ROWS4M(r0, r1, r2, r3, T, m) asm volatile(WN_ROWM(r0, "%[a0]", m) WN_ROWM(r1, "%[a1]", m) WN_ROWM(r2, "%[a2]", m) WN_ROWM(r3, "%[a3]", m) \
    : [t] "+v"(tt), [d] "=&v"(dl) : [nf] "v"(nf), [a0] "v"(T.x), [a1] "v"(T.y), [a2] "v"(T.z), [a3] "v"(T.w))
#define WN_XFM(r, ar) "v_fmac_f32_dpp %[t], %[x], " ar " row_newbcast:" #r " row_mask:0xa bank_mask:0xf\n\t"
#define WN_CROSS4(r0, r1, r2, r3, T) asm volatile(WN_XFM(r0, "%[a0]") WN_XFM(r1, "%[a1]") WN_XFM(r2, "%[a2]") WN_XFM(r3, "%[a3]") \
    : [t] "+v"(tt) : [x] "v"(dx), [a0] "v"(T.x), [a1] "v"(T.y), [a2] "v"(T.z), [a3] "v"(T.w))
struct WnWin32 { float J[24]; float4 A0, A1, A2, A3, C0, C1, C2, C3; float aref, R, nw, half; };

DEV void wn_jt32(const float* J, const float x, float& a_lo, float& a_hi) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f x2 = {x, x};
  float p[24];
#pragma unroll
  for (int k = 0; k < 24; k += 2) { const v2f pr = v2f{J[k], J[k + 1]} * x2; p[k] = pr.x; p[k + 1] = pr.y; }
  float s_lo = wn_fold16(p), s_hi = wn_fold8(p + 16);
  // + the other half's 16 rows: v_permlane16_swap exchanges the odd 16-lane rows of the first register with the even rows of the second,
  // so (x, y) = (lower half's sum, upper half's sum) in EVERY lane afterwards: the same sum, in the same order, in both halves
  // (a VALU instruction: the LDS crossbar's round trip — ds_swizzle — sat three times in every window-sweep's chain)
  float y_lo = s_lo, y_hi = s_hi;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 1" : "+v"(s_lo), "+v"(s_hi), "+v"(y_lo), "+v"(y_hi));
  a_lo += s_lo + y_lo; a_hi += s_hi + y_hi;
}

// one hand-over row (window_pgs.h: WN_NK floats, sparse — a contact row touches at most two free bodies) expanded into the dense J^ over the
// NV dof slots the sweeps use: slot 6 b + j = entry j of the first body's six if b is it, of the second's if b is that one, else 0
template <int NV> DEV void wn_load_row(const float* p, const bool ok, float* J, float& aref, float& R) {
  float v[12];
#pragma unroll
  for (int k = 0; k < 12; k++) v[k] = ok ? p[16 * k] : 0.0f;
  const int b1 = ok ? __float_as_int(p[16 * 12]) : -1, b2 = ok ? __float_as_int(p[16 * 13]) : -1;
  aref = ok ? p[16 * 14] : 0.0f; R = ok ? p[16 * 15] : 0.0f;
#pragma unroll
  for (int b = 0; 6 * b < NV; b++) {
    const bool i1 = b == b1, i2 = b == b2;
#pragma unroll
    for (int j = 0; j < 6; j++) if (6 * b + j < NV) J[6 * b + j] = i1 ? v[j] : (i2 ? v[6 + j] : 0.0f);
  }
}

// launch-order hint of an env (mjh_order_kernel sorts by hint >> 6 into 256 buckets, longest job first): the number of its 16-row windows
// first, its sweeps second — a wavefront of the 16-row form sweeps max(windows) x max(sweeps) over its four envs, so envs of EQUAL window
// count belong together (sorted by the product alone, a wave of {3 windows x 100 sweeps, 6 x 50, ...} ran 6 x 100: S24's mean wave ran
// 12 % more window-sweeps than its mean env; S24 11.24 -> 11.76 M env-steps/s).  Scaled so that the model's largest env fills the buckets.
DEV int wn_cost_hint(const DModel& M, const int nwin16, const int niter) {
#ifdef WN_HINT_PRODUCT
  return min(niter * nwin16 * 20 + 1, 1 << 22);
#else
  const int scale = max(1, 16320 / ((M.win_maxw + 1) * 104));
  return (nwin16 * 104 + min(niter, 103)) * scale + 1;
#endif
}

// MjSim::set_odom_vels (/root/reference/src/mujoco_sim/mj_sim.cpp:1079-1153) behind mj_Euler, as in the fused kernel (step_kernel.h): the commanded
// twist, rotated by the odom angles of the INTEGRATED qpos, overwrites the odom dofs' velocities for the next step.  One lane per env;
// qp: the env's integrated qpos (LDS), qv: its row of S.qvel.
DEV void wn_odom(const DModel& M, const DState& S, const int env, const float* qp, float* qv) {
  const Tab<int> odom{M.I, M.o_odom};
  const float* v = S.odom_vel + (size_t)env * 6;
  const float ax = odom[6] >= 0 ? qp[odom[6]] : 0.0f, ay = odom[7] >= 0 ? qp[odom[7]] : 0.0f, az = odom[8] >= 0 ? qp[odom[8]] : 0.0f;
  const float sx = sinf(ax), cx = cosf(ax), sy = sinf(ay), cy = cosf(ay), sz = sinf(az), cz = cosf(az);
  if (odom[0] >= 0) qv[odom[0]] = v[0]*cy*cz + v[1]*(sx*sy*cz - cx*sz) + v[2]*(cx*sy*cz + sx*sz);
  if (odom[1] >= 0) qv[odom[1]] = v[0]*cy*sz + v[1]*(sx*sy*sz + cx*cz) + v[2]*(cx*sy*sz - sx*cz);
  if (odom[2] >= 0) qv[odom[2]] = -v[0]*sy + v[1]*sx*cy + v[2]*cx*cy;
  for (int k = 0; k < 3; k++) if (odom[3+k] >= 0) qv[odom[3+k]] = v[3+k];
}

// ---- qacc, mj_checkAcc, semi-implicit Euler, state and statistics of the wide forms (32-row: two envs per wavefront, es = 0 / 1; 64-row: one,
// es = 0).  dl_lane: the 16 lanes of the env that carry its dofs (lane q: dofs q and 16 + q / 2); envmask: the env's lanes (mj_checkAcc's vote)
DEV void wn_finish_wide(const DModel& M, const DState& S, float* const wb, const int* const wh, const int env, const int env0, const int xflags,
                        const bool mine, const bool dl_lane, const int es, const unsigned long long envmask, const int q,
                        const float a_lo, const float a_hi, const float as_lo, const float as_hi, const int niter, const int nwin16) {
  const int nv = M.nv;
  const int dhi = 16 + (q >> 1);
  const bool lo_on = q < nv, hi_on = dhi < nv;
  const float sv_lo = (dl_lane && lo_on) ? wb[WN_SINV + q] : 0.0f, sv_hi = (dl_lane && hi_on) ? wb[WN_SINV + dhi] : 0.0f;
  float qa_lo = a_lo * sv_lo, qa_hi = a_hi * sv_hi;
  float qv_lo = (dl_lane && lo_on) ? wb[WN_QVEL + q] : 0.0f, qv_hi = (dl_lane && hi_on) ? wb[WN_QVEL + dhi] : 0.0f;
  int flags = mine ? wh[3] : 0;
  if (dl_lane && (xflags & XF_FORCE)) {
    const size_t xe = (size_t)(env - env0) * M.nvp;
    if (lo_on) { if (S.x_smooth) S.x_smooth[xe + q] = as_lo * sv_lo; if (S.x_constraint) S.x_constraint[xe + q] = (a_lo - as_lo) / sv_lo; }
    if (hi_on && !(q & 1)) { if (S.x_smooth) S.x_smooth[xe + dhi] = as_hi * sv_hi; if (S.x_constraint) S.x_constraint[xe + dhi] = (a_hi - as_hi) / sv_hi; }
  }
  const bool badl = !(qa_lo == qa_lo) || fabsf(qa_lo) > MJ_MAXVAL || !(qa_hi == qa_hi) || fabsf(qa_hi) > MJ_MAXVAL;
  const bool bad = (__ballot(badl) & envmask) != 0ull;
  const size_t qrow = (size_t)env * M.nqp, vrow = (size_t)env * M.nvp;
  if (bad) { qa_lo = qa_hi = 0.0f; qv_lo = qv_hi = 0.0f; flags |= 4; }
  const float h = M.timestep;
  const Tab<int> dof_bodyid{M.I, M.o_dof_bodyid}, jnt_qposadr{M.I, M.o_jnt_qposadr}, jnt_dofadr{M.I, M.o_jnt_dofadr};
  const Tab<float> dof_damping{M.F, M.o_dof_damping};
  const unsigned slotmask = S.slot_mask ? S.slot_mask[env] : 0u;
  const int sbase = M.nbody > 32 ? M.nbody - 32 : 0;
  __shared__ float s_v32[2][32];
  __shared__ float s_qp32[2][40];
  const bool has_odom = M.I[M.o_odom + 9] != 0;
  auto advance = [&](const int d, float& qa, float& qv) __attribute__((always_inline)) {
    float qint = qa;
    if (M.has_damping && !(M.disableflags & MJH_DSBL_EULERDAMP)) {
      const float sv = wb[WN_SINV + d], Mdd = 1.0f / (sv * sv), D = dof_damping[d];
      qint = qa - h * (D * qa) / (Mdd + h * D);
    }
    const unsigned rb = (unsigned)(dof_bodyid[d] - sbase);
    const bool parked = rb < 32u && ((slotmask >> rb) & 1u);
    qv = parked ? 0.0f : qv + h * qint;
    if (parked) qa = 0.0f;
    S.qvel[vrow + d] = qv; S.qacc_ws[vrow + d] = qa;
    if (bad && (xflags & XF_SPLIT2)) S.qfrc_applied[vrow + d] = 0.0f;
    s_v32[es][d] = qv;
  };
  if (dl_lane && lo_on) advance(q, qa_lo, qv_lo);
  if (dl_lane && hi_on && !(q & 1)) advance(dhi, qa_hi, qv_hi);
  __syncthreads();
  if (dl_lane && q < M.njnt) {
    const int qadr = jnt_qposadr[q], da = jnt_dofadr[q];
    float p[7];
#pragma unroll
    for (int k = 0; k < 7; k++) p[k] = bad ? S.initial_qpos[qrow + qadr + k] : wb[WN_QPOS + qadr + k];
    const float* v = s_v32[es] + da;
    p[0] += h * v[0]; p[1] += h * v[1]; p[2] += h * v[2];
    float w3[3] = {v[3], v[4], v[5]};
    quat_integrate(p + 3, w3, h);
#pragma unroll
    for (int k = 0; k < 7; k++) { S.qpos[qrow + qadr + k] = p[k]; s_qp32[es][qadr + k] = p[k]; }
  }
  if (has_odom) {       // (the qvel rows above are this wave's own stores: the overwrite follows them in program order)
    __syncthreads();
    if (dl_lane && q == 0) wn_odom(M, S, env, s_qp32[es], S.qvel + vrow);
  }
  if (dl_lane && q == 0) {
    S.time[env] += M.timestep_d;
    const int cost_hint = wn_cost_hint(M, nwin16, niter);
    S.stats[4 * env] = WN_STAT0(wh[1]); S.stats[4 * env + 1] = wh[2]; S.stats[4 * env + 2] = niter;
    S.stats[4 * env + 3] = ((S.stats[4 * env + 3] | flags) & 0xff) | (cost_hint << 8);
  }
}

DEV void wn_run32(const DConst* __restrict__ C, const DState& S, const int env0, const int nenv, const int xflags, const int blk) {
  const DModel& M = C->M;
  constexpr int NV = 24;
  const int lane = threadIdx.x, es = lane >> 5, hq = (lane >> 4) & 1, q = lane & 15;
  const int slot = blk * 2 + es;
  const bool have = slot < nenv;
  const int env = have ? (S.env_order ? S.env_order[env0 + slot] : env0 + slot) : 0;
  float* const wb = S.wbuf + (size_t)env * (size_t)S.wstride;
  const int* const wh = (const int*)wb;
  const int nrow = (have && wh[4] == 1) ? wh[0] : 0;          // (envs of the 16-row form are not this section's)
  if (__ballot(nrow > 0) == 0ull) return;
  const bool mine = nrow > 0;
  const int nwin16 = (nrow + 15) >> 4, nwin = (nrow + 31) >> 5;
  const int nwmax = max(__builtin_amdgcn_readlane(nwin, 0), __builtin_amdgcn_readlane(nwin, 32));
  const int nv = M.nv;
  const int dhi = 16 + (q >> 1);
  const bool lo_on = q < nv, hi_on = dhi < nv;
  const float as_lo = lo_on ? wb[WN_AS + q] : 0.0f, as_hi = hi_on ? wb[WN_AS + dhi] : 0.0f;
  const float ws_lo = lo_on ? wb[WN_AWS + q] : 0.0f, ws_hi = hi_on ? wb[WN_AWS + dhi] : 0.0f;
  WnWin32 win[WN32_NW];
  float f[WN32_NW];
  const float* rows = wb + WN_ROWS + q;
#pragma unroll
  for (int w = 0; w < WN32_NW; w++) if (w < nwmax) {
    WnWin32& W = win[w];
    const bool ok = (2 * w + hq) < nwin16;                  // (the assemble launch pads the last 16-row block with zero rows; a missing upper block is all zeros)
    wn_load_row<NV>(rows + (2 * w + hq) * WN_NK * 16, ok, W.J, W.aref, W.R);
    float acc[16], acx[16], Jx[NV];
#pragma unroll
    for (int sidx = 0; sidx < 16; sidx++) { acc[sidx] = 0.0f; acx[sidx] = 0.0f; }
#pragma unroll
    for (int k = 0; k < NV; k++) Jx[k] = WN_SWZ16(W.J[k]);   // the other half's row q: upper lanes see the lower rows
#pragma unroll
    for (int k = 0; k < NV; k++) asm volatile("" : "+v"(W.J[k]), "+v"(Jx[k]));
    asm volatile("s_nop 1");
    // (dof k outermost, the 16 tile entries inside: consecutive multiply-adds go to DIFFERENT accumulators.  Entry after entry the compiler put an
    //  `s_nop` between every two of them — its hazard model counts the accumulator of a DPP multiply-add as a DPP source of the next one — and a lone
    //  wave pays an issue slot for each: half of the tile build's slots.  Every accumulator still sums its dofs in the same order: bitwise.)
#define WN_ACC(sidx) PP_FMAC_BC(acc[sidx], W.J[k], W.J[k], sidx); PP_FMAC_BC(acx[sidx], Jx[k], W.J[k], sidx);
#pragma unroll
    for (int k = 0; k < NV; k++) { PP_BC16(WN_ACC) }
#undef WN_ACC
    float diag = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; k++) diag += W.J[k] * W.J[k];
    const float ARqq = diag + W.R;
    const float inv = ARqq < MJ_MINVAL ? 0.0f : 1.0f / ARqq, ninv = -inv, cinv = hq ? ninv : 0.0f;
    W.nw = ninv; W.half = 0.5f * ARqq;
    W.A0 = make_float4(0 < q ? ninv * acc[0] : 0.0f, 1 < q ? ninv * acc[1] : 0.0f, 2 < q ? ninv * acc[2] : 0.0f, 3 < q ? ninv * acc[3] : 0.0f);
    W.A1 = make_float4(4 < q ? ninv * acc[4] : 0.0f, 5 < q ? ninv * acc[5] : 0.0f, 6 < q ? ninv * acc[6] : 0.0f, 7 < q ? ninv * acc[7] : 0.0f);
    W.A2 = make_float4(8 < q ? ninv * acc[8] : 0.0f, 9 < q ? ninv * acc[9] : 0.0f, 10 < q ? ninv * acc[10] : 0.0f, 11 < q ? ninv * acc[11] : 0.0f);
    W.A3 = make_float4(12 < q ? ninv * acc[12] : 0.0f, 13 < q ? ninv * acc[13] : 0.0f, 14 < q ? ninv * acc[14] : 0.0f, 0.0f);
    W.C0 = make_float4(cinv * acx[0], cinv * acx[1], cinv * acx[2], cinv * acx[3]);
    W.C1 = make_float4(cinv * acx[4], cinv * acx[5], cinv * acx[6], cinv * acx[7]);
    W.C2 = make_float4(cinv * acx[8], cinv * acx[9], cinv * acx[10], cinv * acx[11]);
    W.C3 = make_float4(cinv * acx[12], cinv * acx[13], cinv * acx[14], cinv * acx[15]);
  }
  auto rowsum_i32 = [&](int v) __attribute__((always_inline)) { v = wn_rowsum_i(v); return v + __builtin_amdgcn_ds_swizzle(v, 0x401F); };
  auto rowsum_f32 = [&](float v) __attribute__((always_inline)) { v = wn_rowsum_f(v); return v + WN_SWZ16(v); };
#define WN32_FOR_WINDOWS(...) do { _Pragma("unroll") for (int w = 0; w < WN32_NW; w++) if (w < nwmax) { WnWin32& W = win[w]; float& fw = f[w]; __VA_ARGS__ } } while (0)
  // ---- warm start
  float a_lo = as_lo, a_hi = as_hi;
#pragma unroll
  for (int w = 0; w < WN32_NW; w++) f[w] = 0.0f;
  if (!(M.disableflags & MJH_DSBL_WARMSTART)) {
    float da_lo = 0.0f, da_hi = 0.0f;
    WN32_FOR_WINDOWS({
      const float jar = wn_dot<NV>(W.J, ws_lo, ws_hi) - W.aref;
      fw = (jar < 0.0f && W.R > 0.0f) ? -jar / W.R : 0.0f;
      wn_jt32(W.J, fw, da_lo, da_hi);
    });
    float cost = 0.0f;
    WN32_FOR_WINDOWS({
      const float jda = wn_dot<NV>(W.J, da_lo, da_hi), bb = wn_dot<NV>(W.J, as_lo, as_hi) - W.aref;
      cost += fw * (0.5f * (jda + W.R * fw) + bb);
    });
    cost = rowsum_f32(cost);
    if (cost > 0.0f) {
#pragma unroll
      for (int w = 0; w < WN32_NW; w++) f[w] = 0.0f;
    } else { a_lo += da_lo; a_hi += da_hi; }
  }
  // ---- sweeps
  const ImpQ iq = imp_quantum(1.0f / (M.meaninertia * (float)(nv > 1 ? nv : 1)), M.tolerance);
  const int itmax = M.iterations;
  int niter = 0;
  bool act = nrow > 0;
  while (__ballot(act) != 0ull) {
    if (act) {
      int impl = 0;
      WN32_FOR_WINDOWS({
        const float u = wn_dot<NV>(W.J, a_lo, a_hi);
        const float fo = fw;
        float tt = ((u - W.aref) + W.R * fo) * W.nw;
        const float nf = -fo;
        float dl;
        // rows 0..15 (lower half) ...
        WN_ROWS4M(0, 1, 2, 3, W.A0, 0x5); WN_ROWS4M(4, 5, 6, 7, W.A1, 0x5); WN_ROWS4M(8, 9, 10, 11, W.A2, 0x5); WN_ROWS4M(12, 13, 14, 15, W.A3, 0x5);
        asm volatile("v_max_f32 %0, %1, %2" : "=v"(dl) : "v"(tt), "v"(nf));
        // ... their deltas reach the upper half through the cross tile ...
        float dx = dl, dy = dl;                                         // (dx: the lower half's deltas in the upper half's lanes)
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(dx), "+v"(dy));
        WN_CROSS4(0, 1, 2, 3, W.C0); WN_CROSS4(4, 5, 6, 7, W.C1); WN_CROSS4(8, 9, 10, 11, W.C2); WN_CROSS4(12, 13, 14, 15, W.C3);
        // ... rows 16..31 (upper half)
        WN_ROWS4M(0, 1, 2, 3, W.A0, 0xa); WN_ROWS4M(4, 5, 6, 7, W.A1, 0xa); WN_ROWS4M(8, 9, 10, 11, W.A2, 0xa); WN_ROWS4M(12, 13, 14, 15, W.A3, 0xa);
        asm volatile("v_max_f32 %0, %1, %2" : "=v"(dl) : "v"(tt), "v"(nf));
        impl += imp_fixed((W.half * dl) * (2.0f * tt - dl), iq.qs);
        fw = fo + dl;
        wn_jt32(W.J, dl, a_lo, a_hi);
      });
      niter++;
      if (rowsum_i32(impl) < iq.thr || niter >= itmax) act = false;
    }
  }
  wn_finish_wide(M, S, wb, wh, env, env0, xflags, mine, mine && hq == 0, es, 0xffffffffull << (32 * es), q, a_lo, a_hi, as_lo, as_hi, niter, nwin16);
#undef WN32_FOR_WINDOWS
}

// ---- 64-row windows: ONE environment per wavefront, for the environments a cohort's step waits for (S24D: more than WN64_MIN_ROWS rows —
// 11 % of the envs, every one of them at the 100-sweep cap; in the 16-row form 13 .. 19 windows a sweep, most of them streamed).  A window
// = 64 consecutive rows = one row per lane; inside it Gauss-Seidel stays row by row on the strictly lower triangle of the window's AR
// (64 tile entries per lane; the wavefront's four 16-lane rows one after the other, each by the 16-row form's chain, a finished row's
// deltas carried to the rows behind it through the cross entries), between windows the acceleration a^ moves
// through ONE dot and ONE transpose-reduce per 64 rows: ~4.8 issue slots per row against 8 in the 16-row form (which serves four envs
// with them: the wide form buys the shorter chain of the slowest envs with the SIMDs the window kernel leaves idle).  Every 16-lane row
// of the wavefront carries the same a^ (lane q: dofs q and 16 + q / 2): the rows' partial J^T sums are exchanged by v_permlane16_swap /
// v_permlane32_swap, the same sum in the same order in every lane.  WN64_NW windows (192 rows) register-resident, WN64_NT more (320 rows) with their tiles in LDS.  The form is a
// function of the env's own row count (set by the assemble launch); same rows, same order, same row math and stopping rule as the
// other forms, another grouping of the arithmetic (fp32 rounding).
extern __shared__ __attribute__((aligned(16))) float wn_lds[];
DEV float* wn_lds_base() { return wn_lds; }
struct WnWin64 { float J[24]; float A[64]; float aref, R, nw, half; };

// sum over the four 16-lane rows of a wavefront, the same value (same order of additions) in every lane
DEV void wn_rows4_sum2(float& x, float& y) {
  float x2 = x, y2 = y;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 1" : "+v"(x), "+v"(y), "+v"(x2), "+v"(y2));
  x += x2; y += y2;                       // rows 0, 1: r0 + r1; rows 2, 3: r2 + r3
  x2 = x; y2 = y;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1" : "+v"(x), "+v"(y), "+v"(x2), "+v"(y2));
  x += x2; y += y2;                       // (r0 + r1) + (r2 + r3) everywhere
}
DEV void wn_jt64(const float* J, const float x, float& a_lo, float& a_hi) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f x2 = {x, x};
  float p[24];
#pragma unroll
  for (int k = 0; k < 24; k += 2) { const v2f pr = v2f{J[k], J[k + 1]} * x2; p[k] = pr.x; p[k + 1] = pr.y; }
  float s_lo = wn_fold16(p), s_hi = wn_fold8(p + 16);
  wn_rows4_sum2(s_lo, s_hi);
  a_lo += s_lo; a_hi += s_hi;
}
// one 16-lane row's 16 constraint rows (lanes of DPP row mask m): per row  v_max, two wait states, v_fmac ... row_newbcast — the 16-row form's chain
#define WN64_R4(r0, r1, r2, r3, a0_, a1_, a2_, a3_, m) asm volatile(WN_ROWM(r0, "%[a0]", m) WN_ROWM(r1, "%[a1]", m) WN_ROWM(r2, "%[a2]", m) WN_ROWM(r3, "%[a3]", m) \
    : [t] "+v"(tt), [d] "=&v"(dl) : [nf] "v"(nf), [a0] "v"(a0_), [a1] "v"(a1_), [a2] "v"(a2_), [a3] "v"(a3_))
#define WN64_CHAIN16(A, b, m) do { WN64_R4(0, 1, 2, 3, A[b], A[b + 1], A[b + 2], A[b + 3], m); WN64_R4(4, 5, 6, 7, A[b + 4], A[b + 5], A[b + 6], A[b + 7], m); \
    WN64_R4(8, 9, 10, 11, A[b + 8], A[b + 9], A[b + 10], A[b + 11], m); WN64_R4(12, 13, 14, 15, A[b + 12], A[b + 13], A[b + 14], A[b + 15], m); \
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(dl) : "v"(tt), "v"(nf)); } while (0)
// the deltas dx of a finished 16-lane row (replicated into every row) reach the rows behind it (mask m) through their cross entries
#define WN64_XF(r, ar, m) "v_fmac_f32_dpp %[t], %[x], " ar " row_newbcast:" #r " row_mask:" #m " bank_mask:0xf\n\t"
#define WN64_X4(r0, r1, r2, r3, a0_, a1_, a2_, a3_, m) asm volatile(WN64_XF(r0, "%[a0]", m) WN64_XF(r1, "%[a1]", m) WN64_XF(r2, "%[a2]", m) WN64_XF(r3, "%[a3]", m) \
    : [t] "+v"(tt) : [x] "v"(dx), [a0] "v"(a0_), [a1] "v"(a1_), [a2] "v"(a2_), [a3] "v"(a3_))
#define WN64_CROSS16(A, b, m) do { WN64_X4(0, 1, 2, 3, A[b], A[b + 1], A[b + 2], A[b + 3], m); WN64_X4(4, 5, 6, 7, A[b + 4], A[b + 5], A[b + 6], A[b + 7], m); \
    WN64_X4(8, 9, 10, 11, A[b + 8], A[b + 9], A[b + 10], A[b + 11], m); WN64_X4(12, 13, 14, 15, A[b + 12], A[b + 13], A[b + 14], A[b + 15], m); } while (0)
// one sweep over a 64-row window: u = J^ a^, then the four 16-lane rows one after the other (Gauss-Seidel row by row inside each: DPP row
// masks 0x1 .. 0x8), the deltas of a finished row carried to every later row by two lane swaps and 16 broadcast multiply-adds
// (A: the tile array, Bg: index of the 16 entries of lane row g in it, Lg: statement in front of row g — the register-resident windows read
//  W.A[16 g ..], the LDS window loads the 16 entries of row g first)
#define WN64_SWEEP_X(W, fw, A, B0, B1, B2, B3, L0, L1, L2, L3) do { \
    const float u = wn_dot<NV>(W.J, a_lo, a_hi); \
    const float fo = fw; \
    float tt = ((u - W.aref) + W.R * fo) * W.nw; \
    const float nf = -fo; \
    float dl, dx, dy; \
    L0; \
    WN64_CHAIN16(A, B0, 0x1); \
    dx = dl; dy = dl; \
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(dx), "+v"(dy));        /* dx = (d0, d0, ., .) */ \
    dy = dx; \
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(dx), "+v"(dy));        /* dx = (d0, d0, d0, d0) */ \
    WN64_CROSS16(A, B0, 0xe); \
    L1; \
    WN64_CHAIN16(A, B1, 0x2); \
    dx = dl; dy = dl; \
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(dx), "+v"(dy));        /* dy = (d1, d1, ., .) */ \
    dx = dy; \
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(dy), "+v"(dx));        /* dy = (d1, d1, d1, d1) */ \
    dx = dy; \
    WN64_CROSS16(A, B1, 0xc); \
    L2; \
    WN64_CHAIN16(A, B2, 0x4); \
    dx = dl; dy = dl; \
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(dx), "+v"(dy));        /* dx = (., ., d2, d2) */ \
    WN64_CROSS16(A, B2, 0x8); \
    L3; \
    WN64_CHAIN16(A, B3, 0x8); \
    impl += imp_fixed((W.half * dl) * (2.0f * tt - dl), iq.qs); \
    fw = fo + dl; \
    wn_jt64(W.J, dl, a_lo, a_hi); } while (0)
// The same sweep with the chains' wait states put to work (register-resident windows; see WN_FILL below: one independent VALU instruction between
// v_max and the DPP read is enough for a wavefront that is alone on its SIMD).  A finished row's deltas are applied to the NEXT row at once (it
// needs them before its chain starts) and to the rows behind that one inside the next row's chain, one multiply-add per wait slot: rows 1 and 2
// of the wavefront run without nops; every lane still receives the rows' contributions in the same order (bitwise the sweep above).
#define WN64_RF(r, ar, cr, m, mc) "v_max_f32 %[d], %[t], %[nf]\n\tv_fmac_f32_dpp %[t], %[xp], " cr " row_newbcast:" #r " row_mask:" #mc " bank_mask:0xf\n\t" \
                                  "v_fmac_f32_dpp %[t], %[d], " ar " row_newbcast:" #r " row_mask:" #m " bank_mask:0xf\n\t"
#define WN64_R4F(r0, r1, r2, r3, a0_, a1_, a2_, a3_, c0_, c1_, c2_, c3_, m, mc, xp_) asm volatile( \
    WN64_RF(r0, "%[a0]", "%[c0]", m, mc) WN64_RF(r1, "%[a1]", "%[c1]", m, mc) WN64_RF(r2, "%[a2]", "%[c2]", m, mc) WN64_RF(r3, "%[a3]", "%[c3]", m, mc) \
    : [t] "+v"(tt), [d] "=&v"(dl) : [nf] "v"(nf), [xp] "v"(xp_), [a0] "v"(a0_), [a1] "v"(a1_), [a2] "v"(a2_), [a3] "v"(a3_), [c0] "v"(c0_), [c1] "v"(c1_), [c2] "v"(c2_), [c3] "v"(c3_))
#define WN64_CHAIN16F(A, b, m, bc, mc, xp_) do { \
    WN64_R4F(0, 1, 2, 3, A[b], A[b + 1], A[b + 2], A[b + 3], A[bc], A[bc + 1], A[bc + 2], A[bc + 3], m, mc, xp_); \
    WN64_R4F(4, 5, 6, 7, A[b + 4], A[b + 5], A[b + 6], A[b + 7], A[bc + 4], A[bc + 5], A[bc + 6], A[bc + 7], m, mc, xp_); \
    WN64_R4F(8, 9, 10, 11, A[b + 8], A[b + 9], A[b + 10], A[b + 11], A[bc + 8], A[bc + 9], A[bc + 10], A[bc + 11], m, mc, xp_); \
    WN64_R4F(12, 13, 14, 15, A[b + 12], A[b + 13], A[b + 14], A[b + 15], A[bc + 12], A[bc + 13], A[bc + 14], A[bc + 15], m, mc, xp_); \
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(dl) : "v"(tt), "v"(nf)); } while (0)
#define WN64_SWEEP_F(W, fw) do { \
    const float u = wn_dot<NV>(W.J, a_lo, a_hi); \
    const float fo = fw; \
    float tt = ((u - W.aref) + W.R * fo) * W.nw; \
    const float nf = -fo; \
    float dl, dx, dy, dx0, dx1; \
    WN64_CHAIN16(W.A, 0, 0x1); \
    dx = dl; dy = dl; \
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(dx), "+v"(dy));        /* dx = (d0, d0, ., .) */ \
    dy = dx; \
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(dx), "+v"(dy));        /* dx = (d0, d0, d0, d0) */ \
    dx0 = dx; \
    WN64_CROSS16(W.A, 0, 0x2);                                  /* row 0's deltas -> row 1 (now) */ \
    WN64_CHAIN16F(W.A, 16, 0x2, 0, 0xc, dx0);                   /* row 1's chain; in its wait slots: row 0's deltas -> rows 2, 3 */ \
    dx = dl; dy = dl; \
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(dx), "+v"(dy));        /* dy = (d1, d1, ., .) */ \
    dx = dy; \
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(dy), "+v"(dx));        /* dy = (d1, d1, d1, d1) */ \
    dx = dy; dx1 = dy; \
    WN64_CROSS16(W.A, 16, 0x4);                                 /* row 1's deltas -> row 2 (now) */ \
    WN64_CHAIN16F(W.A, 32, 0x4, 16, 0x8, dx1);                  /* row 2's chain; in its wait slots: row 1's deltas -> row 3 */ \
    dx = dl; dy = dl; \
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(dx), "+v"(dy));        /* dx = (., ., d2, d2) */ \
    WN64_CROSS16(W.A, 32, 0x8); \
    WN64_CHAIN16(W.A, 48, 0x8); \
    impl += imp_fixed((W.half * dl) * (2.0f * tt - dl), iq.qs); \
    fw = fo + dl; \
    wn_jt64(W.J, dl, a_lo, a_hi); } while (0)
#ifndef WN_FILL64
#define WN_FILL64 1
#endif
#if WN_FILL64
#define WN64_SWEEP(W, fw) WN64_SWEEP_F(W, fw)
#else
#define WN64_SWEEP(W, fw) WN64_SWEEP_X(W, fw, W.A, 0, 16, 32, 48, (void)0, (void)0, (void)0, (void)0)
#endif

DEV void wn_run64(const DConst* __restrict__ C, const DState& S, const int env0, const int nenv, const int xflags, const int blk, float* const lds64) {
  const DModel& M = C->M;
  constexpr int NV = 24;
  const int lane = threadIdx.x, g = lane >> 4, q = lane & 15;
  const bool have = blk < nenv;
  const int env = have ? (S.env_order ? S.env_order[env0 + blk] : env0 + blk) : 0;
  float* const wb = S.wbuf + (size_t)env * (size_t)S.wstride;
  const int* const wh = (const int*)wb;
  const int nrow = (have && wh[4] == 2) ? wh[0] : 0;            // (uniform: one env per wavefront)
  if (nrow <= 0) return;
  const int nwin16 = (nrow + 15) >> 4, nwin = (nrow + 63) >> 6;
  const int nv = M.nv;
  const int dhi = 16 + (q >> 1);
  const bool lo_on = q < nv, hi_on = dhi < nv;
  const float as_lo = lo_on ? wb[WN_AS + q] : 0.0f, as_hi = hi_on ? wb[WN_AS + dhi] : 0.0f;
  const float ws_lo = lo_on ? wb[WN_AWS + q] : 0.0f, ws_hi = hi_on ? wb[WN_AWS + dhi] : 0.0f;
  WnWin64 win[WN64_NW];
  float f[WN64_NW];
  const float* rows = wb + WN_ROWS + q;
#pragma unroll
  for (int w = 0; w < WN64_NW; w++) if (w < nwin) {
    WnWin64& W = win[w];
    const bool ok = (4 * w + g) < nwin16;                      // (the assemble launch pads the last 16-row block with zero rows; a missing block is all zeros)
    wn_load_row<NV>(rows + (4 * w + g) * WN_NK * 16, ok, W.J, W.aref, W.R);
    float diag = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; k++) diag += W.J[k] * W.J[k];
    const float ARqq = diag + W.R;
    const float inv = ARqq < MJ_MINVAL ? 0.0f : 1.0f / ARqq, ninv = -inv;
    W.nw = ninv; W.half = 0.5f * ARqq;
    // tile row: acc_r = J^_lane . J^_r for the 64 rows r of the window.  The rows of 16-lane group G come over through the LDS crossbar
    // (ds_bpermute: lane (G, q) to every group's lane q), then sixteen broadcast multiply-add chains as in the 16-row form
#pragma unroll
    for (int G = 0; G < 4; G++) {
      float Jg[NV], acc[16];
#pragma unroll
      for (int k = 0; k < NV; k++) Jg[k] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((16 * G + q) << 2, __builtin_bit_cast(int, W.J[k])));
#pragma unroll
      for (int sidx = 0; sidx < 16; sidx++) acc[sidx] = 0.0f;
#pragma unroll
      for (int k = 0; k < NV; k++) asm volatile("" : "+v"(W.J[k]), "+v"(Jg[k]));
      asm volatile("s_nop 1");
#define WN_ACC(sidx) PP_FMAC_BC(acc[sidx], Jg[k], W.J[k], sidx);
#pragma unroll
      for (int k = 0; k < NV; k++) { PP_BC16(WN_ACC) }
#undef WN_ACC
#pragma unroll
      for (int sidx = 0; sidx < 16; sidx++) W.A[16 * G + sidx] = (16 * G + sidx) < lane ? ninv * acc[sidx] : 0.0f;
    }
  }
  // the WN64_NT windows behind the register-resident ones (rows 193 .. 320): J^ and the constants in registers, their 64 x 64 tiles in LDS
  // (16 KB each of the launch's LDS tier; [r / 4][lane][4]: the 16 entries of a lane row are four 16-byte reads), read again every sweep.
  // (All five windows in registers would take 500 of them — and a window kernel beyond 448 registers costs the 16-row form 10 % on S24,
  // measured: the forms share one kernel.)
  struct WnTail { float J[NV]; float aref, R, nw, half; };
  WnTail TLS[WN64_NT];
  float ftl[WN64_NT];
  float* const tile = lds64 + lane * 4;
#pragma unroll
  for (int t = 0; t < WN64_NT; t++) {
    ftl[t] = 0.0f;
    if (WN64_NW + t < nwin) {
    WnTail& TL = TLS[t];
    const bool ok = (4 * (WN64_NW + t) + g) < nwin16;
    wn_load_row<NV>(rows + (4 * (WN64_NW + t) + g) * WN_NK * 16, ok, TL.J, TL.aref, TL.R);
    float diag = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; k++) diag += TL.J[k] * TL.J[k];
    const float ARqq = diag + TL.R;
    const float inv = ARqq < MJ_MINVAL ? 0.0f : 1.0f / ARqq, ninv = -inv;
    TL.nw = ninv; TL.half = 0.5f * ARqq;
#pragma unroll
    for (int G = 0; G < 4; G++) {
      float Jg[NV], acc[16];
#pragma unroll
      for (int k = 0; k < NV; k++) Jg[k] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((16 * G + q) << 2, __builtin_bit_cast(int, TL.J[k])));
#pragma unroll
      for (int sidx = 0; sidx < 16; sidx++) acc[sidx] = 0.0f;
#pragma unroll
      for (int k = 0; k < NV; k++) asm volatile("" : "+v"(TL.J[k]), "+v"(Jg[k]));
      asm volatile("s_nop 1");
#define WN_ACC(sidx) PP_FMAC_BC(acc[sidx], Jg[k], TL.J[k], sidx);
#pragma unroll
      for (int k = 0; k < NV; k++) { PP_BC16(WN_ACC) }
#undef WN_ACC
#pragma unroll
      for (int c4 = 0; c4 < 4; c4++) {
        const int r0 = 16 * G + 4 * c4;
        *(float4*)(tile + t * 4096 + (r0 >> 2) * 256) = make_float4(r0 < lane ? ninv * acc[4 * c4] : 0.0f, r0 + 1 < lane ? ninv * acc[4 * c4 + 1] : 0.0f,
                                                                    r0 + 2 < lane ? ninv * acc[4 * c4 + 2] : 0.0f, r0 + 3 < lane ? ninv * acc[4 * c4 + 3] : 0.0f);
      }
    }
    }
  }
  float T[16];
#define WN64_TLOAD(t, gb) do { const float* tp_ = tile + (t) * 4096 + (4 * (gb)) * 256; \
    const float4 t0_ = *(const float4*)(tp_), t1_ = *(const float4*)(tp_ + 256), t2_ = *(const float4*)(tp_ + 512), t3_ = *(const float4*)(tp_ + 768); \
    T[0] = t0_.x; T[1] = t0_.y; T[2] = t0_.z; T[3] = t0_.w; T[4] = t1_.x; T[5] = t1_.y; T[6] = t1_.z; T[7] = t1_.w; \
    T[8] = t2_.x; T[9] = t2_.y; T[10] = t2_.z; T[11] = t2_.w; T[12] = t3_.x; T[13] = t3_.y; T[14] = t3_.z; T[15] = t3_.w; } while (0)
#define WN64_FOR_TAILS(...) do { _Pragma("unroll") for (int t = 0; t < WN64_NT; t++) if (WN64_NW + t < nwin) { WnTail& TL = TLS[t]; float& ft = ftl[t]; __VA_ARGS__ } } while (0)
  // ---- warm start
  float a_lo = as_lo, a_hi = as_hi;
#pragma unroll
  for (int w = 0; w < WN64_NW; w++) f[w] = 0.0f;
#define WN64_FOR_WINDOWS(...) do { _Pragma("unroll") for (int w = 0; w < WN64_NW; w++) if (w < nwin) { WnWin64& W = win[w]; float& fw = f[w]; __VA_ARGS__ } } while (0)
  if (!(M.disableflags & MJH_DSBL_WARMSTART)) {
    float da_lo = 0.0f, da_hi = 0.0f;
    WN64_FOR_WINDOWS({
      const float jar = wn_dot<NV>(W.J, ws_lo, ws_hi) - W.aref;
      fw = (jar < 0.0f && W.R > 0.0f) ? -jar / W.R : 0.0f;
      wn_jt64(W.J, fw, da_lo, da_hi);
    });
    WN64_FOR_TAILS({
      const float jar = wn_dot<NV>(TL.J, ws_lo, ws_hi) - TL.aref;
      ft = (jar < 0.0f && TL.R > 0.0f) ? -jar / TL.R : 0.0f;
      wn_jt64(TL.J, ft, da_lo, da_hi);
    });
    float cost = 0.0f;
    WN64_FOR_WINDOWS({
      const float jda = wn_dot<NV>(W.J, da_lo, da_hi), bb = wn_dot<NV>(W.J, as_lo, as_hi) - W.aref;
      cost += fw * (0.5f * (jda + W.R * fw) + bb);
    });
    WN64_FOR_TAILS({
      const float jda = wn_dot<NV>(TL.J, da_lo, da_hi), bb = wn_dot<NV>(TL.J, as_lo, as_hi) - TL.aref;
      cost += ft * (0.5f * (jda + TL.R * ft) + bb);
    });
    float dummy = 0.0f;
    cost = wn_rowsum_f(cost); wn_rows4_sum2(cost, dummy);
    if (cost > 0.0f) {
#pragma unroll
      for (int w = 0; w < WN64_NW; w++) f[w] = 0.0f;
#pragma unroll
      for (int t = 0; t < WN64_NT; t++) ftl[t] = 0.0f;
    } else { a_lo += da_lo; a_hi += da_hi; }
  }
  // ---- sweeps
  const ImpQ iq = imp_quantum(1.0f / (M.meaninertia * (float)(nv > 1 ? nv : 1)), M.tolerance);
  const int itmax = M.iterations;
  int niter = 0;
  while (true) {
    int impl = 0;
    WN64_FOR_WINDOWS({ WN64_SWEEP(W, fw); });
    WN64_FOR_TAILS({ WN64_SWEEP_X(TL, ft, T, 0, 0, 0, 0, WN64_TLOAD(t, 0), WN64_TLOAD(t, 1), WN64_TLOAD(t, 2), WN64_TLOAD(t, 3)); });
    niter++;
    if (wave_sum_dpp_i(impl) < iq.thr || niter >= itmax) break;
  }
#undef WN64_FOR_WINDOWS
#undef WN64_TLOAD
#undef WN64_FOR_TAILS
  wn_finish_wide(M, S, wb, wh, env, env0, xflags, true, g == 0, 0, ~0ull, q, a_lo, a_hi, as_lo, as_hi, niter, nwin16);
}

// a^ += J_a^T x_a + J_b^T x_b : two windows of the same 16 lanes, one transpose-reduce
template <int NV> DEV void wn_jt2(const float* JA, const float xa, const float* JB, const float xb, float& a_lo, float& a_hi) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f xa2 = {xa, xa}, xb2 = {xb, xb};
  float p[NV];
#pragma unroll
  for (int k = 0; k < NV; k += 2) {
    const v2f pr = __builtin_elementwise_fma(v2f{JB[k], JB[k + 1]}, xb2, v2f{JA[k], JA[k + 1]} * xa2);     // v_pk_mul_f32, v_pk_fma_f32
    p[k] = pr.x; p[k + 1] = pr.y;
  }
  a_lo += wn_fold16(p);
  if constexpr (NV == 32) a_hi += wn_fold16(p + 16); else a_hi += wn_fold8(p + 16);
}
#define WN_XFMA(r, ar) "v_fmac_f32_dpp %[t], %[x], " ar " row_newbcast:" #r " row_mask:0xf bank_mask:0xf\n\t"
#define WN_CROSS4A(r0, r1, r2, r3, T) asm volatile(WN_XFMA(r0, "%[a0]") WN_XFMA(r1, "%[a1]") WN_XFMA(r2, "%[a2]") WN_XFMA(r3, "%[a3]") \
    : [t] "+v"(tt) : [x] "v"(dx), [a0] "v"(T.x), [a1] "v"(T.y), [a2] "v"(T.z), [a3] "v"(T.w))

// ---- the row chain with its wait states put to work (16-row form, register-resident window pairs).  `v_max -> v_fmac ... row_newbcast` needs
// two wait states between the VALU write and the DPP read; as `s_nop 1` that is a third issue slot per row (11.5 of 27.5 clocks) in which
// nothing happens — 32 per window pair.  This kernel's wavefront is ALONE on its SIMD (425 registers), and a lone wave issues one
// instruction per 8 clocks (profiles/r02m_valu_issue_bench.txt: every instruction kind, never less): ONE independent VALU instruction
// between the two puts 16 clocks between their issues — more than the 12 the two wait states stand for (no instruction between them: 8 clocks,
// measurably too few — HISTORY.md Round 4 "the row chain without its two wait states").  Independent work exists in a PAIR of windows:
// window 2j's chain carries the first 16 multiply-adds of window 2j+1's dot u = J^ a^ (both dots start from the same a^), window 2j+1's
// chain carries window 2j's products J^ delta (12 packed multiplies) and the float part of its cost decrease (4).  Same instructions, same
// operands, same order of every sum: bitwise the chain with the nops (state hash against a -DWN_FILL=0 build).  Kept to this kernel.
#ifndef WN_FILL
#define WN_FILL 1
#endif
#define WN_ROWF(r, ar, fill) "v_max_f32 %[d], %[t], %[nf]\n\t" fill "\n\tv_fmac_f32_dpp %[t], %[d], " ar " row_newbcast:" #r " row_mask:0xf bank_mask:0xf\n\t"
#define WN_FDOT(r, jb) "v_fmac_f32_dpp %[ub], %[al], " jb " row_newbcast:" #r " row_mask:0xf bank_mask:0xf"
// window A's rows r0 .. r0 + 3 (tile entries a0 .. a3), filler: window B's dot over dofs r0 .. r0 + 3 (b0 .. b3 = B.J[r0 ..])
#define WN_ROWS4_FA(r0, r1, r2, r3, T, b0_, b1_, b2_, b3_) asm volatile( \
    WN_ROWF(r0, "%[a0]", WN_FDOT(r0, "%[b0]")) WN_ROWF(r1, "%[a1]", WN_FDOT(r1, "%[b1]")) WN_ROWF(r2, "%[a2]", WN_FDOT(r2, "%[b2]")) WN_ROWF(r3, "%[a3]", WN_FDOT(r3, "%[b3]")) \
    : [t] "+v"(tt), [d] "=&v"(dl), [ub] "+v"(ub0) : [nf] "v"(nf), [a0] "v"(T.x), [a1] "v"(T.y), [a2] "v"(T.z), [a3] "v"(T.w), [al] "v"(a_lo), [b0] "v"(b0_), [b1] "v"(b1_), [b2] "v"(b2_), [b3] "v"(b3_))
// window B's rows, filler: window A's products p_k = J^_A[2k .. 2k + 1] * delta_A (packed)
#define WN_FPK(pk, ja) "v_pk_mul_f32 " pk ", " ja ", %[x2]"
#define WN_ROWS4_FB(r0, r1, r2, r3, T, p0_, p1_, p2_, p3_, j0_, j1_, j2_, j3_) asm volatile( \
    WN_ROWF(r0, "%[a0]", WN_FPK("%[p0]", "%[j0]")) WN_ROWF(r1, "%[a1]", WN_FPK("%[p1]", "%[j1]")) WN_ROWF(r2, "%[a2]", WN_FPK("%[p2]", "%[j2]")) WN_ROWF(r3, "%[a3]", WN_FPK("%[p3]", "%[j3]")) \
    : [t] "+v"(tt), [d] "=&v"(dl), [p0] "=&v"(p0_), [p1] "=&v"(p1_), [p2] "=&v"(p2_), [p3] "=&v"(p3_) \
    : [nf] "v"(nf), [a0] "v"(T.x), [a1] "v"(T.y), [a2] "v"(T.z), [a3] "v"(T.w), [x2] "v"(xa2), [j0] "v"(j0_), [j1] "v"(j1_), [j2] "v"(j2_), [j3] "v"(j3_))
// ... and the float part of A's cost decrease: e = ((half_A delta_A) (2 t_A - delta_A)) qs
#define WN_ROWS4_FC(r0, r1, r2, r3, T) asm volatile( \
    WN_ROWF(r0, "%[a0]", "v_mul_f32 %[e1], %[hf], %[da]") WN_ROWF(r1, "%[a1]", "v_fma_f32 %[e2], 2.0, %[ta], -%[da]") WN_ROWF(r2, "%[a2]", "v_mul_f32 %[e1], %[e1], %[e2]") WN_ROWF(r3, "%[a3]", "v_mul_f32 %[e1], %[e1], %[qs]") \
    : [t] "+v"(tt), [d] "=&v"(dl), [e1] "=&v"(e1), [e2] "=&v"(e2) \
    : [nf] "v"(nf), [a0] "v"(T.x), [a1] "v"(T.y), [a2] "v"(T.z), [a3] "v"(T.w), [hf] "v"(A.half), [da] "v"(dla), [ta] "v"(tta), [qs] "v"(iq.qs))

// Cross tiles of the register-resident pairs in LDS instead of registers (the 24-slot instance): 16 values per pair and lane that the sweep uses once,
// in four `ds_read_b128` instead of sixteen copies out of the accumulation half of the register file (a VALU instruction takes its operands from
// the architectural half only; the kernel holds 422 registers) — 48 registers fewer, 12 KB of LDS per wavefront behind the tier.  Same values,
// same arithmetic: bitwise.
#ifndef WN_X_LDS
#define WN_X_LDS 1
#endif
// Only launches WITHOUT the LDS tier take it (instance XL; S24's default): beside the tier's 36 KB the 12 KB would cost the fourth window wavefront
// of a CU its place (S24D: 5.83 -> 4.67 M, measured).
#define WN_XLDS_BYTES(nvt, nw) (((nvt) == 24 && WN_FILL && WN_X_LDS) ? ((nw) / 2) * 4 * 64 * 16 : 0)
template <int NV, int NW, bool XL = false>
__global__ __launch_bounds__(64, 1) void mjh_window_kernel(const DConst* __restrict__ C, const DState S, const int env0, const int nenv, const int nl, const int xflags, const int n32waves, const int n64waves) {
  // the first n64waves wavefronts: the section of the envs with the most rows (one per wavefront, 64-row windows); the next n32waves: the
  // section of the envs with many rows (two per wavefront, 32-row windows); dispatched first — they are the launch's longest jobs
#ifdef WN_PROF_CLK
  if (threadIdx.x == 0) wn_t0 = (long long)__builtin_amdgcn_s_memtime();
  __syncthreads();
#endif
  if constexpr (NV == 24) {
#ifndef WN_NO_F64
    if ((int)blockIdx.x < n64waves) { wn_run64(C, S, env0, nenv, xflags, (int)blockIdx.x, wn_lds_base()); return; }
#endif
    if ((int)blockIdx.x < n64waves + n32waves) { wn_run32(C, S, env0, nenv, xflags, (int)blockIdx.x - n64waves); return; }
  }
  const DModel& M = C->M;
  const int lane = threadIdx.x, rho = lane >> 4, q = lane & 15;
  const int slot = ((int)blockIdx.x - n32waves - n64waves) * 4 + rho;
  const bool have = slot < nenv;
  const int env = have ? (S.env_order ? S.env_order[env0 + slot] : env0 + slot) : 0;
  float* const wb = S.wbuf + (size_t)env * (size_t)S.wstride;
  const int* const wh = (const int*)wb;
  const int nrow = (have && !(n32waves > 0 && wh[4] == 1) && !(n64waves > 0 && wh[4] == 2)) ? wh[0] : 0;
  const bool mine = nrow > 0 || (have && wh[5] == 1);      // (else: a row without an environment, or one that finished in the assemble launch — it must not write anything; [5]: an env without rows handed over by the split API: integrated here)
  if (__ballot(mine) == 0ull) return;
  const int nwin = (nrow + 15) >> 4;
  const int nwmax = max(max(__builtin_amdgcn_readlane(nwin, 0), __builtin_amdgcn_readlane(nwin, 16)), max(__builtin_amdgcn_readlane(nwin, 32), __builtin_amdgcn_readlane(nwin, 48)));
  const int nv = M.nv;
  // dof vectors: lane q of the row carries dof q (lo) and dof 16 + q (hi; the 24-slot instance: dof 16 + (q >> 1), wn_fold8)
  const int dhi = NV == 24 ? 16 + (q >> 1) : 16 + q;
  const bool lo_on = q < nv, hi_on = dhi < nv;
  const float as_lo = lo_on ? wb[WN_AS + q] : 0.0f, as_hi = hi_on ? wb[WN_AS + dhi] : 0.0f;
  const float ws_lo = lo_on ? wb[WN_AWS + q] : 0.0f, ws_hi = hi_on ? wb[WN_AWS + dhi] : 0.0f;

  WnWin<NV> win[NW];
  float f[NW];                                                 // forces of the register-resident windows
  const float* rows = wb + WN_ROWS + q;
  auto load_rows = [&](WnWin<NV>& W, const int w) __attribute__((always_inline)) {
    const bool ok = w < nwin;
    wn_load_row<NV>(rows + w * WN_NK * 16, ok, W.J, W.aref, W.R);
  };
  // tile row of a window: acc_r = J^_q . J^_r for the 16 rows r of the window (every lane of the row at once), then -AR_qr / AR_qq, r < q
  auto make_tile = [&](WnWin<NV>& W) __attribute__((always_inline)) {
    float acc[16];
#pragma unroll
    for (int s = 0; s < 16; s++) acc[s] = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; k++) asm volatile("" : "+v"(W.J[k]));     // (materialised before the DPP reads below: no VALU write within two instructions of them)
    asm volatile("s_nop 1");
    // (dof k outermost: consecutive multiply-adds go to different accumulators, no `s_nop` between them — see wn_run32's tile build)
#define WN_ACC(s) PP_FMAC_BC(acc[s], W.J[k], W.J[k], s);
#pragma unroll
    for (int k = 0; k < NV; k++) { PP_BC16(WN_ACC) }
#undef WN_ACC
    float diag = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; k++) diag += W.J[k] * W.J[k];
    const float ARqq = diag + W.R;
    const float inv = ARqq < MJ_MINVAL ? 0.0f : 1.0f / ARqq, ninv = -inv;
    W.nw = ninv; W.half = 0.5f * ARqq;
    W.A0 = make_float4(0 < q ? ninv * acc[0] : 0.0f, 1 < q ? ninv * acc[1] : 0.0f, 2 < q ? ninv * acc[2] : 0.0f, 3 < q ? ninv * acc[3] : 0.0f);
    W.A1 = make_float4(4 < q ? ninv * acc[4] : 0.0f, 5 < q ? ninv * acc[5] : 0.0f, 6 < q ? ninv * acc[6] : 0.0f, 7 < q ? ninv * acc[7] : 0.0f);
    W.A2 = make_float4(8 < q ? ninv * acc[8] : 0.0f, 9 < q ? ninv * acc[9] : 0.0f, 10 < q ? ninv * acc[10] : 0.0f, 11 < q ? ninv * acc[11] : 0.0f);
    W.A3 = make_float4(12 < q ? ninv * acc[12] : 0.0f, 13 < q ? ninv * acc[13] : 0.0f, 14 < q ? ninv * acc[14] : 0.0f, 0.0f);
  };
  // windows beyond the register-resident ones: the whole record — the
  // next NL windows of every env in LDS (the kernel has no other use for it: 40 KB per wave at four waves per CU), the rest in the env's
  // slice of global memory (read every sweep: slow, and rare — S24D's 140-row piles reach the LDS tier only)
  // record of such a window, float4 [NX4][16 lanes]: J^ (NV / 4), the tile row (4), {aref, R, -1 / AR_qq, AR_qq / 2}, {force, -, -, -} — read
  // with NX4 - 1 sixteen-byte loads and one four-byte load per sweep (45 four-byte loads of the [k][16] layout cost the lone wave 33 more
  // issue slots per window and sweep)
  constexpr int NX4 = WN_XREC(NV) / 4, NJ4 = NV / 4;
  static_assert(NV % 4 == 0 && NX4 == NJ4 + 6, "window record layout");
  auto store_ext = [&](float4* t, const WnWin<NV>& W, const float fw) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NJ4; k++) t[16 * k] = make_float4(W.J[4 * k], W.J[4 * k + 1], W.J[4 * k + 2], W.J[4 * k + 3]);
    t[16 * NJ4] = W.A0; t[16 * (NJ4 + 1)] = W.A1; t[16 * (NJ4 + 2)] = W.A2; t[16 * (NJ4 + 3)] = W.A3;
    t[16 * (NJ4 + 4)] = make_float4(W.aref, W.R, W.nw, W.half);
    *(float*)(t + 16 * (NJ4 + 5)) = fw;
  };
  auto load_ext = [&](const float4* t, WnWin<NV>& W) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NJ4; k++) { const float4 j = t[16 * k]; W.J[4 * k] = j.x; W.J[4 * k + 1] = j.y; W.J[4 * k + 2] = j.z; W.J[4 * k + 3] = j.w; }
    W.A0 = t[16 * NJ4]; W.A1 = t[16 * (NJ4 + 1)]; W.A2 = t[16 * (NJ4 + 2)]; W.A3 = t[16 * (NJ4 + 3)];
    const float4 c = t[16 * (NJ4 + 4)];
    W.aref = c.x; W.R = c.y; W.nw = c.z; W.half = c.w;
  };
#define WN_XF(t) (*(float*)((t) + 16 * (NJ4 + 5)))       /* the record's force */
  float4* const xl = (float4*)wn_lds + (rho * nl * NX4) * 16 + q;                                   // LDS tier: window NW + j at xl + j * NX4 * 16
  float4* const xg = (float4*)(wb + WN_ROWS + M.win_maxw * WN_NK * 16) + q;                         // global tier: window w at xg + w * NX4 * 16
  const int nwl = min(nwmax, NW + nl);
#pragma unroll
  for (int w = 0; w < NW; w++) if (w < nwmax) { load_rows(win[w], w); make_tile(win[w]); }
  // The register-resident windows are swept two at a time: window 2j + 1 takes its residual from the acceleration BEFORE window 2j's
  // deltas plus the cross tile X_j[r] = (-1 / AR_qq) J^_{2j+1,q} . J^_{2j,r} times those deltas (16 broadcast multiply-adds), so that both
  // windows' J^T delta go through ONE transpose-reduce: ~255 instead of 300 issue slots per 32 rows on every env's chain.  Same rows, same
  // order, same row math; the grouping of the arithmetic differs (fp32 rounding).
  static_assert(NW % 2 == 0, "register-resident windows are swept in pairs");
  float4 X[NW / 2][4];
  auto make_cross = [&](WnWin<NV>& A, WnWin<NV>& B, float4 (&Xo)[4]) __attribute__((always_inline)) {
    float acx[16];
#pragma unroll
    for (int sidx = 0; sidx < 16; sidx++) acx[sidx] = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; k++) asm volatile("" : "+v"(A.J[k]), "+v"(B.J[k]));
    asm volatile("s_nop 1");
#define WN_ACX(sidx) PP_FMAC_BC(acx[sidx], A.J[k], B.J[k], sidx);
#pragma unroll
    for (int k = 0; k < NV; k++) { PP_BC16(WN_ACX) }
#undef WN_ACX
    Xo[0] = make_float4(B.nw * acx[0], B.nw * acx[1], B.nw * acx[2], B.nw * acx[3]);
    Xo[1] = make_float4(B.nw * acx[4], B.nw * acx[5], B.nw * acx[6], B.nw * acx[7]);
    Xo[2] = make_float4(B.nw * acx[8], B.nw * acx[9], B.nw * acx[10], B.nw * acx[11]);
    Xo[3] = make_float4(B.nw * acx[12], B.nw * acx[13], B.nw * acx[14], B.nw * acx[15]);
  };
  constexpr bool X_LDS = XL && NV == 24 && WN_FILL && WN_X_LDS;
  float4* const xs = (float4*)wn_lds + (4 * nl * NX4) * 16 + lane;                              // (behind the tier) pair j, quarter c of the lane at xs[(4 j + c) * 64]
#pragma unroll
  for (int j = 0; j < NW / 2; j++) if (2 * j + 1 < nwmax) {
    make_cross(win[2 * j], win[2 * j + 1], X[j]);
    if constexpr (X_LDS) {
#pragma unroll
      for (int c = 0; c < 4; c++) xs[(4 * j + c) * 64] = X[j][c];
    }
  }
  for (int w = NW; w < nwl; w++) { WnWin<NV> W; load_rows(W, w); make_tile(W); store_ext(xl + (w - NW) * NX4 * 16, W, 0.0f); }
  for (int w = nwl; w < nwmax; w++) { WnWin<NV> W; load_rows(W, w); make_tile(W); if (mine) store_ext(xg + w * NX4 * 16, W, 0.0f); }
  // a tier window for a sweep: record and force (a lane without an env, or with one that finished in the assemble launch, sweeps zero rows)
  auto load_tier = [&](const int w, WnWin<NV>& W, float& fw) __attribute__((always_inline)) {
    if (w < nwl) { const float4* t = xl + (w - NW) * NX4 * 16; load_ext(t, W); fw = WN_XF(t); }
    else {
      const float4* t = xg + w * NX4 * 16; fw = 0.0f;
      if (mine) { load_ext(t, W); fw = WN_XF(t); } else { load_rows(W, WN_MAXW); W.A0 = W.A1 = W.A2 = W.A3 = make_float4(0, 0, 0, 0); W.nw = 0; W.half = 0; }
    }
  };
  auto store_tier_force = [&](const int w, const float fw) __attribute__((always_inline)) {
    if (w < nwl) WN_XF(xl + (w - NW) * NX4 * 16) = fw; else if (mine) WN_XF(xg + w * NX4 * 16) = fw;
  };
  // the tiers' windows are swept in pairs too (the filled chain, 24-slot instance): cross tile of pair (NW + 2 j, NW + 2 j + 1) at xx + j * 64,
  // in the env's slice behind the records
  constexpr bool TIER_PAIRS = NV == 24 && WN_FILL && WN_TIER_PAIRS;
  float4* const xx = xg + M.win_maxw * NX4 * 16;
  if constexpr (TIER_PAIRS) for (int w = NW; w + 1 < nwmax; w += 2) {
    WnWin<NV> A, B; float fdum; float4 Xo[4];
    load_tier(w, A, fdum); load_tier(w + 1, B, fdum);
    make_cross(A, B, Xo);
    if (mine) { float4* tx = xx + ((w - NW) >> 1) * 64; tx[0] = Xo[0]; tx[16] = Xo[1]; tx[32] = Xo[2]; tx[48] = Xo[3]; }
  }
  // every window of the wave, register-resident ones first; the body sees the window W and its force fw
#define WN_FOR_WINDOWS(...) do { \
    _Pragma("unroll") for (int w = 0; w < NW; w++) if (w < nwmax) { WnWin<NV>& W = win[w]; float& fw = f[w]; __VA_ARGS__ } \
    for (int w = NW; w < nwl; w++) { WnWin<NV> W; float4* t = xl + (w - NW) * NX4 * 16; load_ext(t, W); float fw = WN_XF(t); __VA_ARGS__ WN_XF(t) = fw; } \
    for (int w = nwl; w < nwmax; w++) { WnWin<NV> W; float4* t = xg + w * NX4 * 16; float fw = 0.0f; if (mine) { load_ext(t, W); fw = WN_XF(t); } else { load_rows(W, WN_MAXW); W.A0 = W.A1 = W.A2 = W.A3 = make_float4(0, 0, 0, 0); W.nw = 0; W.half = 0; } \
                                          __VA_ARGS__ if (mine) WN_XF(t) = fw; } } while (0)
#define WN_ZERO_EXT_FORCES() do { for (int w = NW; w < nwl; w++) WN_XF(xl + (w - NW) * NX4 * 16) = 0.0f; \
                                  if (mine) for (int w = nwl; w < nwmax; w++) WN_XF(xg + w * NX4 * 16) = 0.0f; } while (0)

  // ---- warm start (mj_fwdConstraint): f = max(0, -(J a_ws - aref) / R), kept if the dual cost is not positive
  float a_lo = as_lo, a_hi = as_hi;
#pragma unroll
  for (int w = 0; w < NW; w++) f[w] = 0.0f;
  if (!(M.disableflags & MJH_DSBL_WARMSTART)) {
    float da_lo = 0.0f, da_hi = 0.0f;
    WN_FOR_WINDOWS({
      const float jar = wn_dot<NV>(W.J, ws_lo, ws_hi) - W.aref;
      fw = (jar < 0.0f && W.R > 0.0f) ? -jar / W.R : 0.0f;
      wn_jt<NV>(W.J, fw, da_lo, da_hi);
    });
    float cost = 0.0f;
    WN_FOR_WINDOWS({
      const float jda = wn_dot<NV>(W.J, da_lo, da_hi), bb = wn_dot<NV>(W.J, as_lo, as_hi) - W.aref;
      cost += fw * (0.5f * (jda + W.R * fw) + bb);
    });
    cost = wn_rowsum_f(cost);
    if (cost > 0.0f) {
#pragma unroll
      for (int w = 0; w < NW; w++) f[w] = 0.0f;
      WN_ZERO_EXT_FORCES();
    } else { a_lo += da_lo; a_hi += da_hi; }
  }
  // ---- sweeps: every env (16-lane row) until ITS improvement falls below the tolerance
  const ImpQ iq = imp_quantum(1.0f / (M.meaninertia * (float)(nv > 1 ? nv : 1)), M.tolerance);
  const int itmax = M.iterations;
  int niter = 0;
  // one sweep of a PAIR of windows with the chains' wait states filled (see WN_ROWF above): the register-resident pairs, and the pairs of the
  // tiers beyond them (records and cross tile loaded for the sweep)
  int impl = 0;
  auto sweep_pair_fill = [&](WnWin<NV>& A, WnWin<NV>& B, const float4& X0, const float4& X1, const float4& X2, const float4& X3, float& fa, float& fb) __attribute__((always_inline)) {
    if constexpr (NV == 24) {
      typedef float v2f __attribute__((ext_vector_type(2)));
      const float ua = wn_dot<NV>(A.J, a_lo, a_hi);
      float ub1;                                   // window B's dot over dofs 16 .. 23 (a_hi: lane 2j carries dof 16 + j)
      asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                   WN_FM("%0", "%1", "%3", 2) WN_FM("%0", "%1", "%4", 4) WN_FM("%0", "%1", "%5", 6) WN_FM("%0", "%1", "%6", 8) WN_FM("%0", "%1", "%7", 10)
                   WN_FM("%0", "%1", "%8", 12) WN_FM("%0", "%1", "%9", 14)
                   : "=&v"(ub1) : "v"(a_hi), "v"(B.J[16]), "v"(B.J[17]), "v"(B.J[18]), "v"(B.J[19]), "v"(B.J[20]), "v"(B.J[21]), "v"(B.J[22]), "v"(B.J[23]));
      const float foa = fa, fob = fb;
      float tta, dla, dlb, ub0;
      {
        float tt = ((ua - A.aref) + A.R * foa) * A.nw;
        const float nf = -foa;
        float dl;
        // (row 0's filler opens window B's dot: a product, not a multiply-add)
        asm volatile(WN_ROWF(0, "%[a0]", "v_mul_f32_dpp %[ub], %[al], %[b0] row_newbcast:0 row_mask:0xf bank_mask:0xf") WN_ROWF(1, "%[a1]", WN_FDOT(1, "%[b1]")) WN_ROWF(2, "%[a2]", WN_FDOT(2, "%[b2]")) WN_ROWF(3, "%[a3]", WN_FDOT(3, "%[b3]"))
                     : [t] "+v"(tt), [d] "=&v"(dl), [ub] "=&v"(ub0) : [nf] "v"(nf), [a0] "v"(A.A0.x), [a1] "v"(A.A0.y), [a2] "v"(A.A0.z), [a3] "v"(A.A0.w), [al] "v"(a_lo), [b0] "v"(B.J[0]), [b1] "v"(B.J[1]), [b2] "v"(B.J[2]), [b3] "v"(B.J[3]));
        WN_ROWS4_FA(4, 5, 6, 7, A.A1, B.J[4], B.J[5], B.J[6], B.J[7]); WN_ROWS4_FA(8, 9, 10, 11, A.A2, B.J[8], B.J[9], B.J[10], B.J[11]); WN_ROWS4_FA(12, 13, 14, 15, A.A3, B.J[12], B.J[13], B.J[14], B.J[15]);
        asm volatile("v_max_f32 %0, %1, %2" : "=v"(dl) : "v"(tt), "v"(nf));
        fa = foa + dl; dla = dl; tta = tt;
      }
      v2f p[NV / 2];
      float e1, e2;
      {
        const float ub = ub0 + ub1;
        float tt = ((ub - B.aref) + B.R * fob) * B.nw;
        float dx = dla;
        asm volatile("s_nop 1" : "+v"(dx));
        WN_CROSS4A(0, 1, 2, 3, X0); WN_CROSS4A(4, 5, 6, 7, X1); WN_CROSS4A(8, 9, 10, 11, X2); WN_CROSS4A(12, 13, 14, 15, X3);
        const float nf = -fob;
        float dl;
        const v2f xa2 = {dla, dla};
#define WN_J2(W, k) v2f{W.J[2 * (k)], W.J[2 * (k) + 1]}
        WN_ROWS4_FB(0, 1, 2, 3, B.A0, p[0], p[1], p[2], p[3], WN_J2(A, 0), WN_J2(A, 1), WN_J2(A, 2), WN_J2(A, 3));
        WN_ROWS4_FB(4, 5, 6, 7, B.A1, p[4], p[5], p[6], p[7], WN_J2(A, 4), WN_J2(A, 5), WN_J2(A, 6), WN_J2(A, 7));
        WN_ROWS4_FB(8, 9, 10, 11, B.A2, p[8], p[9], p[10], p[11], WN_J2(A, 8), WN_J2(A, 9), WN_J2(A, 10), WN_J2(A, 11));
        WN_ROWS4_FC(12, 13, 14, 15, B.A3);
        asm volatile("v_max_f32 %0, %1, %2" : "=v"(dl) : "v"(tt), "v"(nf));
        impl += (int)__builtin_amdgcn_fmed3f(e1, -(float)(2 << MJH_IMP_BITS), (float)(2 << MJH_IMP_BITS));
        impl += imp_fixed((B.half * dl) * (2.0f * tt - dl), iq.qs);
        fb = fob + dl; dlb = dl;
      }
      {   // a^ += J_A^T delta_A + J_B^T delta_B: A's products are there, B's join them, one transpose-reduce (wn_jt2)
        const v2f xb2 = {dlb, dlb};
        float pp[NV];
#pragma unroll
        for (int k = 0; k < NV / 2; k++) { const v2f pr = __builtin_elementwise_fma(WN_J2(B, k), xb2, p[k]); pp[2 * k] = pr.x; pp[2 * k + 1] = pr.y; }
#undef WN_J2
        a_lo += wn_fold16(pp);
        a_hi += wn_fold8(pp + 16);
      }
    }
  };
  bool act = nrow > 0;
  while (__ballot(act) != 0ull) {
    if (act) {
      impl = 0;
#define WN_SWEEP_ONE(W, fw) do { \
        const float u = wn_dot<NV>(W.J, a_lo, a_hi); \
        const float fo = fw; \
        float tt = ((u - W.aref) + W.R * fo) * W.nw; \
        const float nf = -fo; \
        float dl; \
        PP_ROWS4(0, 1, 2, 3, W.A0); PP_ROWS4(4, 5, 6, 7, W.A1); PP_ROWS4(8, 9, 10, 11, W.A2); PP_ROWS4(12, 13, 14, 15, W.A3); \
        asm volatile("v_max_f32 %0, %1, %2" : "=v"(dl) : "v"(tt), "v"(nf)); \
        impl += imp_fixed((W.half * dl) * (2.0f * tt - dl), iq.qs); \
        fw = fo + dl; \
        wn_jt<NV>(W.J, dl, a_lo, a_hi); } while (0)
      // register-resident windows: pairs (see the cross tiles above), a last odd one alone
#pragma unroll
      for (int j = 0; j < NW / 2; j++) if (2 * j < nwmax) {
        if (2 * j + 1 < nwmax) {
          WnWin<NV>& A = win[2 * j]; WnWin<NV>& B = win[2 * j + 1];
          if constexpr (NV == 24 && WN_FILL) {
            if constexpr (X_LDS) {
              const float4 x0 = xs[(4 * j) * 64], x1 = xs[(4 * j + 1) * 64], x2 = xs[(4 * j + 2) * 64], x3 = xs[(4 * j + 3) * 64];      // (requested here: they arrive during window A's chain)
              sweep_pair_fill(A, B, x0, x1, x2, x3, f[2 * j], f[2 * j + 1]);
            } else
            sweep_pair_fill(A, B, X[j][0], X[j][1], X[j][2], X[j][3], f[2 * j], f[2 * j + 1]);
          } else {
          const float ua = wn_dot<NV>(A.J, a_lo, a_hi), ub = wn_dot<NV>(B.J, a_lo, a_hi);
          float dla, dlb;
          {
            const float fo = f[2 * j];
            float tt = ((ua - A.aref) + A.R * fo) * A.nw;
            const float nf = -fo;
            float dl;
            PP_ROWS4(0, 1, 2, 3, A.A0); PP_ROWS4(4, 5, 6, 7, A.A1); PP_ROWS4(8, 9, 10, 11, A.A2); PP_ROWS4(12, 13, 14, 15, A.A3);
            asm volatile("v_max_f32 %0, %1, %2" : "=v"(dl) : "v"(tt), "v"(nf));
            impl += imp_fixed((A.half * dl) * (2.0f * tt - dl), iq.qs);
            f[2 * j] = fo + dl; dla = dl;
          }
          {
            const float fo = f[2 * j + 1];
            float tt = ((ub - B.aref) + B.R * fo) * B.nw;
            float dx = dla;
            asm volatile("s_nop 1" : "+v"(dx));
            WN_CROSS4A(0, 1, 2, 3, X[j][0]); WN_CROSS4A(4, 5, 6, 7, X[j][1]); WN_CROSS4A(8, 9, 10, 11, X[j][2]); WN_CROSS4A(12, 13, 14, 15, X[j][3]);
            const float nf = -fo;
            float dl;
            PP_ROWS4(0, 1, 2, 3, B.A0); PP_ROWS4(4, 5, 6, 7, B.A1); PP_ROWS4(8, 9, 10, 11, B.A2); PP_ROWS4(12, 13, 14, 15, B.A3);
            asm volatile("v_max_f32 %0, %1, %2" : "=v"(dl) : "v"(tt), "v"(nf));
            impl += imp_fixed((B.half * dl) * (2.0f * tt - dl), iq.qs);
            f[2 * j + 1] = fo + dl; dlb = dl;
          }
          wn_jt2<NV>(A.J, dla, B.J, dlb, a_lo, a_hi);
          }
        } else { WnWin<NV>& W = win[2 * j]; WN_SWEEP_ONE(W, f[2 * j]); }
      }
      // the tiers beyond them: in pairs as well (a last odd one alone), or one window at a time
      int wt = NW;
      if constexpr (TIER_PAIRS) for (; wt + 1 < nwmax; wt += 2) {
        WnWin<NV> A, B; float fa, fb;
        load_tier(wt, A, fa); load_tier(wt + 1, B, fb);
        const float4* tx = xx + ((wt - NW) >> 1) * 64;
        float4 X0 = make_float4(0, 0, 0, 0), X1 = X0, X2 = X0, X3 = X0;
        if (mine) { X0 = tx[0]; X1 = tx[16]; X2 = tx[32]; X3 = tx[48]; }
        sweep_pair_fill(A, B, X0, X1, X2, X3, fa, fb);
        store_tier_force(wt, fa); store_tier_force(wt + 1, fb);
      }
      // (two loops: the LDS tier's sweep waits on the LDS counter only, the global tier's on the memory counter only — one loop over both: S24D 5.32 -> 5.06 M)
      for (; wt < nwl; wt++) { WnWin<NV> W; float4* t = xl + (wt - NW) * NX4 * 16; load_ext(t, W); float fw = WN_XF(t); WN_SWEEP_ONE(W, fw); WN_XF(t) = fw; }
      for (; wt < nwmax; wt++) {
        WnWin<NV> W; float4* t = xg + wt * NX4 * 16; float fw = 0.0f;
        if (mine) { load_ext(t, W); fw = WN_XF(t); } else { load_rows(W, WN_MAXW); W.A0 = W.A1 = W.A2 = W.A3 = make_float4(0, 0, 0, 0); W.nw = 0; W.half = 0; }
        WN_SWEEP_ONE(W, fw);
        if (mine) WN_XF(t) = fw;
      }
#undef WN_SWEEP_ONE
      niter++;
      if (wn_rowsum_i(impl) < iq.thr || niter >= itmax) act = false;
    }
  }
  // ---- qacc, mj_checkAcc, semi-implicit Euler (mj_Euler; free joints only), state and statistics
  const float sv_lo = (mine && lo_on) ? wb[WN_SINV + q] : 0.0f, sv_hi = (mine && hi_on) ? wb[WN_SINV + dhi] : 0.0f;
  float qa_lo = a_lo * sv_lo, qa_hi = a_hi * sv_hi;
  float qv_lo = (mine && lo_on) ? wb[WN_QVEL + q] : 0.0f, qv_hi = (mine && hi_on) ? wb[WN_QVEL + dhi] : 0.0f;
  int flags = mine ? wh[3] : 0;
  if (mine && (xflags & XF_FORCE)) {   // the split API's exports (mjh_step2): qacc_smooth and qfrc_constraint = M (qacc - qacc_smooth), indexed by the row inside the launch's range
    const size_t xe = (size_t)(env - env0) * M.nvp;
    if (lo_on) { if (S.x_smooth) S.x_smooth[xe + q] = as_lo * sv_lo; if (S.x_constraint) S.x_constraint[xe + q] = (a_lo - as_lo) / sv_lo; }
    if (hi_on && (NV != 24 || !(q & 1))) { if (S.x_smooth) S.x_smooth[xe + dhi] = as_hi * sv_hi; if (S.x_constraint) S.x_constraint[xe + dhi] = (a_hi - as_hi) / sv_hi; }
  }
  const bool badl = !(qa_lo == qa_lo) || fabsf(qa_lo) > MJ_MAXVAL || !(qa_hi == qa_hi) || fabsf(qa_hi) > MJ_MAXVAL;
  const bool bad = ((__ballot(badl) >> (16 * rho)) & 0xffffull) != 0ull;
  const size_t qrow = (size_t)env * M.nqp, vrow = (size_t)env * M.nvp;
  if (bad) { qa_lo = qa_hi = 0.0f; qv_lo = qv_hi = 0.0f; flags |= 4; }
  const float h = M.timestep;
  const Tab<int> dof_bodyid{M.I, M.o_dof_bodyid}, jnt_qposadr{M.I, M.o_jnt_qposadr}, jnt_dofadr{M.I, M.o_jnt_dofadr};
  const Tab<float> dof_damping{M.F, M.o_dof_damping};
  const unsigned slotmask = S.slot_mask ? S.slot_mask[env] : 0u;
  const int sbase = M.nbody > 32 ? M.nbody - 32 : 0;
  __shared__ float s_v[4][32];
  __shared__ float s_qp[4][40];
  const bool has_odom = M.I[M.o_odom + 9] != 0;
  auto advance = [&](const int d, float& qa, float& qv) __attribute__((always_inline)) {
    float qint = qa;
    if (M.has_damping && !(M.disableflags & MJH_DSBL_EULERDAMP)) {
      // (M + h D) qacc' = M qacc, M diagonal: qacc' = qacc - h D qacc / (M_dd + h D)
      const float sv = wb[WN_SINV + d], Mdd = 1.0f / (sv * sv), D = dof_damping[d];
      qint = qa - h * (D * qa) / (Mdd + h * D);
    }
    const unsigned rb = (unsigned)(dof_bodyid[d] - sbase);
    const bool parked = rb < 32u && ((slotmask >> rb) & 1u);
    qv = parked ? 0.0f : qv + h * qint;
    if (parked) qa = 0.0f;
    S.qvel[vrow + d] = qv; S.qacc_ws[vrow + d] = qa;
    if (bad && (xflags & XF_SPLIT2)) S.qfrc_applied[vrow + d] = 0.0f;
    s_v[rho][d] = qv;
  };
  if (mine && lo_on) advance(q, qa_lo, qv_lo);
  if (mine && hi_on && (NV != 24 || !(q & 1))) advance(dhi, qa_hi, qv_hi);
  __syncthreads();
  if (mine && q < M.njnt) {
    const int qadr = jnt_qposadr[q], da = jnt_dofadr[q];
    float p[7];
#pragma unroll
    for (int k = 0; k < 7; k++) p[k] = bad ? S.initial_qpos[qrow + qadr + k] : wb[WN_QPOS + qadr + k];
    const float* v = s_v[rho] + da;
    p[0] += h * v[0]; p[1] += h * v[1]; p[2] += h * v[2];
    float w3[3] = {v[3], v[4], v[5]};
    quat_integrate(p + 3, w3, h);
#pragma unroll
    for (int k = 0; k < 7; k++) { S.qpos[qrow + qadr + k] = p[k]; s_qp[rho][qadr + k] = p[k]; }
  }
  if (has_odom) {
    __syncthreads();
    if (mine && q == 0) wn_odom(M, S, env, s_qp[rho], S.qvel + vrow);
  }
  if (mine && q == 0) {
    S.time[env] += M.timestep_d;
    // launch-order hint: sweeps x windows in units of the fused kernel's hint (patch_pgs.h: about four instructions)
    const int cost_hint = wn_cost_hint(M, nwin, niter);
    S.stats[4 * env] = WN_STAT0(wh[1]); S.stats[4 * env + 1] = wh[2]; S.stats[4 * env + 2] = niter;
    S.stats[4 * env + 3] = ((S.stats[4 * env + 3] | flags) & 0xff) | (cost_hint << 8);
  }
#undef WN_FOR_WINDOWS
#undef WN_ZERO_EXT_FORCES
#undef WN_XF
}
