// dev_math.h — fp32 device helpers: small vector/quaternion algebra, spatial algebra,
// wave-level (64-lane) reductions through DPP.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#define DEV __device__ __forceinline__
#define MJ_MINVAL 1e-15f
#define MJ_MAXVAL 1e10f

DEV float dot3(const float* a, const float* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
DEV void cross3(float* r, const float* a, const float* b) {
  float x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2], z = a[0]*b[1] - a[1]*b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
DEV float norm3(const float* a) { return sqrtf(dot3(a, a)); }
DEV float normalize3(float* a) {
  float n = norm3(a);
  if (n < MJ_MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; } else { float s = 1.0f / n; a[0] *= s; a[1] *= s; a[2] *= s; }
  return n;
}
DEV void normalize4(float* q) {
  const float n2 = q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3];
  // a quaternion within a few ulp of unit length is left alone (as mju_normalize4 leaves |n - 1| < mjMINVAL alone): re-scaling it
  // flips it between two neighbouring representations, so that the state a launch stores would depend on HOW MANY launches have
  // normalised it (split API: step1 | inverse | step2, each launch runs mj_kinematics) — with this the operation is idempotent
  if (fabsf(n2 - 1.0f) <= 4.8e-7f) return;
  const float n = sqrtf(n2);
  if (n < MJ_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else { float s = 1.0f / n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
}
DEV void mulquat(float* r, const float* a, const float* b) {
  float w = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3];
  float x = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
  float y = a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1];
  float z = a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
DEV void quat2mat(float* m, const float* q) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w*w + x*x - y*y - z*z; m[1] = 2*(x*y - w*z);         m[2] = 2*(x*z + w*y);
  m[3] = 2*(x*y + w*z);         m[4] = w*w - x*x + y*y - z*z; m[5] = 2*(y*z - w*x);
  m[6] = 2*(x*z - w*y);         m[7] = 2*(y*z + w*x);         m[8] = w*w - x*x - y*y + z*z;
}
DEV void rotvec(float* r, const float* m, const float* v) {  // r = M v
  float x = m[0]*v[0] + m[1]*v[1] + m[2]*v[2], y = m[3]*v[0] + m[4]*v[1] + m[5]*v[2], z = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
DEV void rotvecT(float* r, const float* m, const float* v) {  // r = M^T v
  float x = m[0]*v[0] + m[3]*v[1] + m[6]*v[2], y = m[1]*v[0] + m[4]*v[1] + m[7]*v[2], z = m[2]*v[0] + m[5]*v[1] + m[8]*v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
DEV void axisangle2quat(float* q, const float* axis, float angle) {
  float s, c; __sincosf(0.5f * angle, &s, &c);
  q[0] = c; q[1] = axis[0]*s; q[2] = axis[1]*s; q[3] = axis[2]*s;
}
// q <- q * exp(h w / 2), w in the body frame
DEV void quat_integrate(float* q, const float* w, float h) {
  float ax[3] = {w[0], w[1], w[2]};
  float n = norm3(ax);
  if (n < MJ_MINVAL) { normalize4(q); return; }
  float inv = 1.0f / n; ax[0] *= inv; ax[1] *= inv; ax[2] *= inv;
  float dq[4], r[4];
  float s = sinf(0.5f * h * n), c = cosf(0.5f * h * n);
  dq[0] = c; dq[1] = ax[0]*s; dq[2] = ax[1]*s; dq[3] = ax[2]*s;
  normalize4(q); mulquat(r, q, dq); normalize4(r);
  q[0] = r[0]; q[1] = r[1]; q[2] = r[2]; q[3] = r[3];
}
// spatial vectors: (rotational 3, translational 3)
DEV void cross_motion(float* r, const float* v, const float* x) {
  float a[3], b[3], c[3];
  cross3(a, v, x); cross3(b, v, x + 3); cross3(c, v + 3, x);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
DEV void cross_force(float* r, const float* v, const float* f) {
  float a[3], b[3], c[3];
  cross3(a, v, f); cross3(b, v + 3, f + 3); cross3(c, v, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
// 10-number spatial inertia about a point offset `off` from the COM
DEV void inert_com(float* res, const float* diagI, const float* mat, const float* off, float mass) {
  float I[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) I[3*r+c] = mat[3*r]*diagI[0]*mat[3*c] + mat[3*r+1]*diagI[1]*mat[3*c+1] + mat[3*r+2]*diagI[2]*mat[3*c+2];
  float d2 = dot3(off, off);
  res[0] = I[0] + mass * (d2 - off[0]*off[0]); res[1] = I[4] + mass * (d2 - off[1]*off[1]); res[2] = I[8] + mass * (d2 - off[2]*off[2]);
  res[3] = I[1] - mass * off[0]*off[1]; res[4] = I[2] - mass * off[0]*off[2]; res[5] = I[5] - mass * off[1]*off[2];
  res[6] = mass * off[0]; res[7] = mass * off[1]; res[8] = mass * off[2]; res[9] = mass;
}
DEV void mul_inert_vec(float* res, const float* i, const float* v) {
  float r0 = i[0]*v[0] + i[3]*v[1] + i[4]*v[2] - i[8]*v[4] + i[7]*v[5];
  float r1 = i[3]*v[0] + i[1]*v[1] + i[5]*v[2] + i[8]*v[3] - i[6]*v[5];
  float r2 = i[4]*v[0] + i[5]*v[1] + i[2]*v[2] - i[7]*v[3] + i[6]*v[4];
  float r3 = i[8]*v[1] - i[7]*v[2] + i[9]*v[3];
  float r4 = i[6]*v[2] - i[8]*v[0] + i[9]*v[4];
  float r5 = i[7]*v[0] - i[6]*v[1] + i[9]*v[5];
  res[0] = r0; res[1] = r1; res[2] = r2; res[3] = r3; res[4] = r4; res[5] = r5;
}
DEV float sel3(float a, float b, float c, int k) { return k == 0 ? a : (k == 1 ? b : c); }

// ---- wave64 cross-lane primitives (DPP; no LDS traffic) ----
#define MJH_DPP_ADD(v, ctrl, rm, bc) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rm, 0xf, bc))
DEV float readlane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
// sum over lanes [0, 16*NROW): result broadcast (uniform) to every lane
template <int NROW> DEV float wave_sum(float v) {
  MJH_DPP_ADD(v, 0x111, 0xf, true);  // row_shr:1
  MJH_DPP_ADD(v, 0x112, 0xf, true);  // row_shr:2
  MJH_DPP_ADD(v, 0x114, 0xf, true);  // row_shr:4
  MJH_DPP_ADD(v, 0x118, 0xf, true);  // row_shr:8   -> lane 15 of every row holds its row total
  if (NROW == 1) return readlane_f(v, 15);
  MJH_DPP_ADD(v, 0x142, 0xa, false);  // row_bcast:15 into rows 1,3
  if (NROW == 2) return readlane_f(v, 31);
  MJH_DPP_ADD(v, 0x143, 0xc, false);  // row_bcast:31 into rows 2,3
  return readlane_f(v, 63);
}
// N simultaneous sums over lanes [0,16*NROW): the DPP steps of the N chains are issued interleaved so
// that each chain's two wait states are filled by the others (results broadcast, uniform)
#define MJH_DPP_STEP4(v, N, ctrl, rm, bc) do { MJH_DPP_ADD(v[0], ctrl, rm, bc); if (N > 1) MJH_DPP_ADD(v[1], ctrl, rm, bc); \
    if (N > 2) MJH_DPP_ADD(v[2], ctrl, rm, bc); if (N > 3) MJH_DPP_ADD(v[3], ctrl, rm, bc); } while (0)
template <int NROW, int N> DEV void wave_sum4(float* v) {
  MJH_DPP_STEP4(v, N, 0x111, 0xf, true);
  MJH_DPP_STEP4(v, N, 0x112, 0xf, true);
  MJH_DPP_STEP4(v, N, 0x114, 0xf, true);
  MJH_DPP_STEP4(v, N, 0x118, 0xf, true);
  if (NROW >= 2) MJH_DPP_STEP4(v, N, 0x142, 0xa, false);
  if (NROW >= 4) MJH_DPP_STEP4(v, N, 0x143, 0xc, false);
  const int src = NROW == 1 ? 15 : (NROW == 2 ? 31 : 63);
#pragma unroll
  for (int j = 0; j < N; j++) v[j] = readlane_f(v[j], src);
}
// inclusive prefix sum over the 64 lanes: Hillis-Steele inside every 16-lane row by four row shifts (lanes without a source add 0), then the rows'
// totals handed on (lane 15 -> the next odd row, lane 31 -> rows 2 and 3).  Six data-parallel adds instead of six ds_bpermute round trips with a
// select each (integers: the same sums in any order).  Every lane of the wave must be active.
#ifndef MJH_DPP_SCAN
#define MJH_DPP_SCAN 1
#endif
DEV int wave_incl_scan_i(int v, int lane) {
#if MJH_DPP_SCAN
  (void)lane;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true); v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
#else
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(v, o); if (lane >= o) v += t; }
  return v;
#endif
}
DEV int wave_last_i(int v) { return __builtin_amdgcn_readlane(v, 63); }      // lane 63's value in every lane (the total of an inclusive scan)
DEV int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// integer sum over the 64 lanes through DPP (uniform result): order-independent, so a convergence test built on it does not depend
// on which lane carried which term
DEV int wave_sum_dpp_i(int v) {
#define MJH_IADD(ctrl, rm, bc) v += __builtin_amdgcn_update_dpp(0, v, ctrl, rm, 0xf, bc)
  MJH_IADD(0x111, 0xf, true); MJH_IADD(0x112, 0xf, true); MJH_IADD(0x114, 0xf, true); MJH_IADD(0x118, 0xf, true);
  MJH_IADD(0x142, 0xa, false); MJH_IADD(0x143, 0xc, false);
#undef MJH_IADD
  return __builtin_amdgcn_readlane(v, 63);
}
// One term of a sweep's cost decrease as a fixed-point integer: x = decrease * (scale / tolerance) (the sweep has converged when the
// terms sum to less than 1), clamped to +-2 (a single term of 2 decides the test) and counted in units of 2^-MJH_IMP_BITS.  Integer
// addition is associative: the total does not depend on the order in which a schedule visits independent blocks, which float
// addition would (a sequential sweep and a side-by-side schedule of the same Gauss-Seidel order must stop after the same sweep).
#define MJH_IMP_BITS 18     // 2048 terms x 2 x 2^18 < 2^31; sweeps over more terms count in coarser units (imp_quantum: nterms)
DEV int imp_fixed(const float decrease, const float qscale, const float clampv = (float)(2 << MJH_IMP_BITS)) {   // qscale = scale / tolerance * 2^bits
  return (int)__builtin_amdgcn_fmed3f(decrease * qscale, -clampv, clampv);
}
struct ImpQ { float qs; int thr; float cl; };     // thr: the sweep has converged when the fixed-point total is below it; cl: the clamp of one term (2 units)
// nterms: an upper bound of the terms one sweep adds up (blocks of the env).  Up to 2047 the unit is 2^-18, beyond it the unit grows so
// that nterms terms, every one at the clamp, still fit 31 bits (a many-body model with thousands of active blocks: the int32 total must
// not wrap in the first sweeps, when every term sits at the clamp)
DEV ImpQ imp_quantum(const float scale, const float tol, const int nterms = 0) {      // tolerance 0: never converged (the float test improvement * scale < 0 never holds either)
  ImpQ q; const bool on = tol > 0.0f;
  int bits = MJH_IMP_BITS;
  if (nterms >= 2048) { bits = 29 - (32 - __builtin_clz((unsigned)nterms)); if (bits < 2) bits = 2; }
  q.qs = on ? scale / tol * (float)(1 << bits) : 0.0f; q.thr = on ? (1 << bits) : (int)0x80000000; q.cl = (float)(2 << bits);
  return q;
}
DEV bool wave_any(bool p) { return __ballot(p) != 0ull; }
// maximum / minimum over the 64 lanes (uniform result).  A lane without a source in a DPP step keeps its own value.
#define MJH_DPP_KEEP(v, ctrl, rm) __builtin_amdgcn_update_dpp(v, v, ctrl, rm, 0xf, false)
DEV float wave_max_f(float v) {
#define MJH_MAXSTEP(ctrl, rm) v = fmaxf(v, __builtin_bit_cast(float, MJH_DPP_KEEP(__builtin_bit_cast(int, v), ctrl, rm)))
  MJH_MAXSTEP(0x111, 0xf); MJH_MAXSTEP(0x112, 0xf); MJH_MAXSTEP(0x114, 0xf); MJH_MAXSTEP(0x118, 0xf); MJH_MAXSTEP(0x142, 0xa); MJH_MAXSTEP(0x143, 0xc);
#undef MJH_MAXSTEP
  return readlane_f(v, 63);
}
DEV int wave_min_i(int v) {
#define MJH_MINSTEP(ctrl, rm) v = min(v, MJH_DPP_KEEP(v, ctrl, rm))
  MJH_MINSTEP(0x111, 0xf); MJH_MINSTEP(0x112, 0xf); MJH_MINSTEP(0x114, 0xf); MJH_MINSTEP(0x118, 0xf); MJH_MINSTEP(0x142, 0xa); MJH_MINSTEP(0x143, 0xc);
#undef MJH_MINSTEP
  return __builtin_amdgcn_readlane(v, 63);
}

// Visits the vertices of a convex hull in index order, eight at a time: the loads of a batch are issued together (a plain loop
// waits out the full memory latency per vertex — the hulls are read from global memory, a different one per lane).
template <class F>
DEV void mesh_scan(const float* vert, const int nvert, F f) {
  for (int i0 = 0; i0 < nvert; i0 += 8) {
    float x[8], y[8], z[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { const int i = min(i0 + u, nvert - 1); x[u] = vert[3*i]; y[u] = vert[3*i+1]; z[u] = vert[3*i+2]; }
#pragma unroll
    for (int u = 0; u < 8; u++) if (i0 + u < nvert) f(i0 + u, x[u], y[u], z[u]);
  }
}
