// step_kernel.h — the many-environment stepping kernel: one 64-lane wavefront per environment,
// the whole mj_step1 -> [mj_inverse] -> mj_step2 pipeline of the reference loop
// (/root/reference/src/mj_main.cpp:82-112) fused in a single launch with every per-env
// intermediate (body frames, spatial quantities, mass matrix, contacts, constraint rows) in LDS.
// HBM traffic per env-step is the state row only.
//
// Stage map (reference call site -> block below):
//   mj_step1 (mj_main.cpp:83)  FK, COM/cdof, CRBA, L'DL, collision, constraint rows, velocity stage
//   MjSim::controller (mj_sim.cpp:1055-1077)            "controller"
//   mj_inverse (mj_hw_interface.cpp:61)                  "inverse"
//   mj_step2 (mj_main.cpp:108)  smooth acceleration, PGS, implicit-damping Euler
//   MjSim::set_odom_vels (mj_sim.cpp:1079-1153)          "odom"
#pragma once
#include "dev_collide.h"
#include "dev_convex.h"
#include "dev_types.h"

#define WSYNC() __syncthreads()
// stage boundary k: optional shader-clock stamp (XF_PROF) and optional early exit (xflags bits 8..11 = k, debug:
// per-stage instruction counts are differences of the hardware counters of launches that stop at successive stages)
#define PROF(k) do { if ((xflags & XF_PROF) && lane == 0) S.x_prof[(size_t)blockIdx.x * PROF_STRIDE + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
                     if (((xflags >> 8) & 15) == (k) && (k) > 0) return; } while (0)
#define MINIMP 0.0001f
#define MAXIMP 0.9999f

DEV float get_impedance(const float* si, float pos, float margin) {
  float s0 = fminf(MAXIMP, fmaxf(MINIMP, si[0])), s1 = fminf(MAXIMP, fmaxf(MINIMP, si[1]));
  float s2 = fmaxf(0.0f, si[2]), s3 = fminf(MAXIMP, fmaxf(MINIMP, si[3])), s4 = fmaxf(1.0f, si[4]);
  if (s0 == s1 || s2 <= MJ_MINVAL) return 0.5f * (s0 + s1);
  float x = fabsf((pos - margin) / s2);
  if (x >= 1) return s1;
  if (x <= 0) return s0;
  float y;
  if (s4 == 1) y = x;
  else if (x <= s3) y = powf(x, s4) / powf(s3, s4 - 1);
  else y = 1 - powf(1 - x, s4) / powf(1 - s3, s4 - 1);
  return s0 + y * (s1 - s0);
}

template <class T> struct Tab {
  const T* base; int off;
  DEV T operator[](int i) const { return base[off + i]; }
  DEV const T* operator+(int k) const { return base + off + k; }
};
// ... or read through the model descriptor at the place of use (LAZY): the table's offset is a scalar load where the table is read instead of one of ~70 scalars
// loaded at the kernel's entry and kept — spilled to vector-register lanes and reloaded by v_readlane — until their last use.  Per instance of the kernel: the
// assemble-only instances of the window chain take it (S24's: 29.9 k -> 28.0 k instructions, 3 967 -> 2 274 v_readlane, 88 -> 264 scalar loads; states bitwise);
// the other instances keep the eager form and their code (the many-body instances' rounding moves with it, C3 loses 2 %: HISTORY.md Round 6)
template <bool LAZY, class T, int DModel::*OFF> struct XTab {
  const DModel* m; const T* base; int off;
  DEV explicit XTab(const DModel& M) : m(&M), base(nullptr), off(0) { if constexpr (!LAZY) { base = tb(); off = M.*OFF; } }
  DEV const T* tb() const { if constexpr (T(1.5) == T(1)) return (const T*)m->I; else return (const T*)m->F; }
#ifdef MJH_TAB_PROBE_L2     // probe: every table read of the lazy instances past the L1 (volatile: glc) — what a table read's latency is worth to the assemble launch
  DEV T operator[](int i) const { if constexpr (LAZY) return *(volatile const T*)(tb() + (m->*OFF + i)); else return base[off + i]; }
#else
  DEV T operator[](int i) const { if constexpr (LAZY) return tb()[m->*OFF + i]; else return base[off + i]; }
#endif
  DEV const T* operator+(int k) const { if constexpr (LAZY) return tb() + (m->*OFF + k); else return base + off + k; }
};

// the per-env LDS arrays the same way: `lds + L.n` formed where the array is used (LAZY) instead of 43 LDS addresses formed at the kernel's entry and kept
template <bool LAZY, int Lay::*OFF> struct XArr {
  float* p; const Lay* l; float* b;
  DEV XArr(float* lds, const Lay& L) : p(nullptr), l(&L), b(lds) { if constexpr (!LAZY) p = lds + L.*OFF; }
  DEV float* get() const { if constexpr (LAZY) return b + l->*OFF; else return p; }
  DEV operator float*() const { return get(); }
  template <class U> DEV explicit operator U*() const { return (U*)get(); }
};

// in-place x <- M^-1 x over one tree's contiguous dof range, x addressed by absolute dof index
DEV void solve_tree(float* x, const float* qLD, const float* qLDinv, const int* dof_parentid, const int* dof_Madr, int adr, int num, const int STRIDE = 1) {
  for (int k = adr + num - 1; k >= adr; k--) {
    float xk = x[k * STRIDE];
    if (xk == 0) continue;
    int a = dof_Madr[k] + 1;
    for (int i = dof_parentid[k]; i >= 0; i = dof_parentid[i]) x[i * STRIDE] -= qLD[a++] * xk;
  }
  for (int k = adr; k < adr + num; k++) x[k * STRIDE] *= qLDinv[k];
  for (int k = adr; k < adr + num; k++) {
    int a = dof_Madr[k] + 1; float xk = x[k * STRIDE];
    for (int i = dof_parentid[k]; i >= 0; i = dof_parentid[i]) xk -= qLD[a++] * x[i * STRIDE];
    x[k * STRIDE] = xk;
  }
}
// The first half of solve_tree for the dense row-space solver (dense_pgs.h), which works with Y = J L^-1 instead of B = J M^-1
// (M^-1 = L^-1 D^-1 L^-T, so J M^-1 J^T = Y D^-1 Y^T): x <- L^-T x keeps a row's sparsity — only the ancestors of its nonzero
// dofs are touched, a contact row of a 49-dof robot costs ~170 updates instead of ~1200 — and tree_l_levels (x <- L^-1 x) finishes a solve.
DEV void solve_tree_lt(float* x, const float* qLD, const int* dof_parentid, const int* dof_Madr, int adr, int num, const int STRIDE = 1) {
  for (int k = adr + num - 1; k >= adr; k--) {
    float xk = x[k * STRIDE];
    if (xk == 0) continue;
    int a = dof_Madr[k] + 1;
    for (int i = dof_parentid[k]; i >= 0; i = dof_parentid[i]) x[i * STRIDE] -= qLD[a++] * xk;
  }
}
DEV void factor_tree(float* qLD, float* qLDinv, const int* dof_parentid, const int* dof_Madr, int adr, int num) {
  for (int k = adr + num - 1; k >= adr; k--) {
    int Mkk_a = dof_Madr[k], Mki = Mkk_a + 1;
    float Mkk = qLD[Mkk_a];
    float inv = 1.0f / Mkk;
    for (int i = dof_parentid[k]; i >= 0; i = dof_parentid[i]) {
      float tmp = qLD[Mki] * inv;
      int cnt = 0, ai = dof_Madr[i];
      for (int j = i; j >= 0; j = dof_parentid[j]) { qLD[ai + cnt] -= tmp * qLD[Mki + cnt]; cnt++; }
      qLD[Mki] = tmp;
      Mki++;
    }
    qLDinv[k] = inv;
  }
}

// ---- wave-cooperative versions for long kinematic trees (articulated robots: one 20-50 dof tree per env, for which the
// lane-per-tree routines above leave 63 lanes idle during O(n depth^2) work).  Sequential over the dofs k of the tree,
// lanes = ancestors a_p of k (p = 1..depth); every update of one k touches a different row / entry per lane.
#define MJH_WAVE_TREE_MIN 12
// anc[dof_Madr[k] + p] = (p-th ancestor a of dof k) | dof_Madr[a] << 16  (p = 0: k itself): same layout as the rows of qLD, built
// once per launch.  The depth of k is the length of its qLD row minus one.
#define ANC_DOF(w) ((w) & 0xffff)
#define ANC_MADR(w) ((int)((unsigned)(w) >> 16))
// One dof k at a time, from the leaves up (mj_factorM's order, so the result is the sequential one bit for bit); lanes = the
// pairs (p, q), 1 <= p <= q <= depth, of ancestors of k:  M[a_p][a_q] -= (M_kp / M_kk) M_kq.  Row k is only read, so everything a
// lane needs from it is fetched in one go (row starts and depths come out of registers by v_readlane), then the targets, then the
// stores: two LDS round trips per dof instead of one per dependent index (PR2: 110 k -> 25 k clocks for 49 dofs).
// the pairs of one dof of depth d <= 4 NP, d <= 16 NQ:  p = 1 + 4 it + (lane >> 4), it < NP;  q = 1 + 16 c + (lane & 15), c < NQ.
// Every load is unconditional (a lane without a pair reads an address that is valid anyway): a predicated load becomes a
// branch with its own s_waitcnt, and the point is to have all of a round's loads in flight together.
template <int NP, int NQ>
DEV void factor_dof_pairs(float* qLD, float* qLDinv, const int* anc, const int Mk, const int d, const int k, const int lane) {
  const int q = 1 + (lane & 15), pr = lane >> 4;
  const float dk = qLD[Mk];
  float aq[NQ], ap[NP], tg[NP][NQ]; int am[NP];
#pragma unroll
  for (int c = 0; c < NQ; c++) aq[c] = qLD[Mk + min(q + 16 * c, d)];
#pragma unroll
  for (int it = 0; it < NP; it++) { const int pp = min(1 + 4 * it + pr, d); ap[it] = qLD[Mk + pp]; am[it] = ANC_MADR(anc[Mk + pp]); }
  const float inv = 1.0f / dk;
#pragma unroll
  for (int it = 0; it < NP; it++)
#pragma unroll
    for (int c = 0; c < NQ; c++) {
      tg[it][c] = 0.0f;
      if (16 * c + 16 > 4 * it) {      // (the chunk holds some q >= p)
        const int pp = 1 + 4 * it + pr, qq = q + 16 * c; const bool v = pp <= d && qq >= pp && qq <= d;
        tg[it][c] = qLD[v ? am[it] + (qq - pp) : Mk];
      }
    }
#pragma unroll
  for (int it = 0; it < NP; it++)
#pragma unroll
    for (int c = 0; c < NQ; c++)
      if (16 * c + 16 > 4 * it) {
        const int pp = 1 + 4 * it + pr, qq = q + 16 * c; const bool v = pp <= d && qq >= pp && qq <= d;
        if (v) qLD[am[it] + (qq - pp)] = tg[it][c] - ap[it] * inv * aq[c];
      }
#pragma unroll
  for (int c = 0; c < NQ; c++) if (pr == 0 && q + 16 * c <= d) qLD[Mk + q + 16 * c] = aq[c] * inv;
  if (lane == 0) qLDinv[k] = inv;
}
// depth <= 15: the d (d + 1) / 2 <= 120 pairs enumerated along the triangle (q-major: pair t = q (q - 1) / 2 + p - 1, so the pairs of a
// dof of depth d are exactly t < d (d + 1) / 2), two per lane (t = lane, lane + 64) — the 4 x 16 tiling above keeps a fifth of its lanes busy
DEV void factor_dof_tri(float* qLD, float* qLDinv, const int* anc, const int Mk, const int d, const int k, const int lane,
                        const int pA, const int qA, const int pB, const int qB) {
  const int np = (d * (d + 1)) >> 1;
  const bool vA = lane < np, vB = lane + 64 < np;
  const float dk = qLD[Mk];
  const float apA = qLD[Mk + min(pA, d)], aqA = qLD[Mk + min(qA, d)], apB = qLD[Mk + min(pB, d)], aqB = qLD[Mk + min(qB, d)];
  const int amA = ANC_MADR(anc[Mk + min(pA, d)]), amB = ANC_MADR(anc[Mk + min(pB, d)]);
  const float row = qLD[Mk + min(1 + lane, d)];                      // row k itself: scaled by 1 / M_kk at the end
  const float inv = 1.0f / dk;
  const float tgA = qLD[vA ? amA + (qA - pA) : Mk], tgB = qLD[vB ? amB + (qB - pB) : Mk];
  if (vA) qLD[amA + (qA - pA)] = tgA - apA * inv * aqA;
  if (vB) qLD[amB + (qB - pB)] = tgB - apB * inv * aqB;
  if (lane < d) qLD[Mk + 1 + lane] = row * inv;
  if (lane == 0) qLDinv[k] = inv;
}
DEV void factor_tree_wave(float* qLD, float* qLDinv, const int* anc, const int* dof_Madr, int adr, int num, int nM, int nv, const int lane) {
  int chunk = -1, MkL = 0, dL = 0;
  int pA, qA, pB, qB;       // the lane's two pairs of the triangular enumeration
  { int t = lane, q = 1; while (t >= q) { t -= q; q++; } pA = t + 1; qA = q;
    t = lane + 64; q = 1; while (t >= q) { t -= q; q++; } pB = t + 1; qB = q; }
  for (int k = adr + num - 1; k >= adr; k--) {
    const int ck = (k - adr) >> 6;
    if (ck != chunk) {
      chunk = ck;
      const int kk = adr + 64 * ck + lane;
      MkL = kk < adr + num ? dof_Madr[kk] : 0;
      dL = kk < adr + num ? (kk + 1 < nv ? dof_Madr[kk + 1] : nM) - MkL - 1 : 0;
    }
    const int Mk = __builtin_amdgcn_readlane(MkL, (k - adr) & 63), d = __builtin_amdgcn_readlane(dL, (k - adr) & 63);
    if (d <= 15) factor_dof_tri(qLD, qLDinv, anc, Mk, d, k, lane, pA, qA, pB, qB);
    else if (d <= 16) factor_dof_pairs<4, 1>(qLD, qLDinv, anc, Mk, d, k, lane);
    else if (d <= 32) factor_dof_pairs<8, 2>(qLD, qLDinv, anc, Mk, d, k, lane);
    else {
      const int q = 1 + (lane & 15), pr = lane >> 4;
      const float inv = 1.0f / qLD[Mk];
      for (int p0 = 1; p0 <= d; p0 += 4) {
        const int pp = p0 + pr;
        for (int q0 = 0; q0 < d; q0 += 16) {
          const int qq = q0 + q;
          if (pp <= d && qq >= pp && qq <= d) {
            const int ai = ANC_MADR(anc[Mk + pp]);
            qLD[ai + (qq - pp)] -= qLD[Mk + pp] * inv * qLD[Mk + qq];
          }
        }
      }
      __syncthreads();
      for (int p = 1 + lane; p <= d; p += 64) qLD[Mk + p] *= inv;
      if (lane == 0) qLDinv[k] = inv;
    }
    __syncthreads();
  }
}
// Short trees (fewer than MJH_WAVE_TREE_MIN dofs: the arms of C3, free bodies beside a robot), four at a time: tree t of a
// group on the 16-lane row t & 3 of the wave, its dofs from the leaf up in lockstep with the other rows; lanes of a row = the
// ancestors q of the current dof (depth <= 10), the pairs (p, q) for all p with their loads issued together.  Same arithmetic
// per entry as factor_tree (bit-identical), a quarter of its dependent LDS round trips.
template <int PMAX>      // deepest dof the step handles
DEV void factor_short_step(float* qLD, float* qLDinv, const int* anc, const int* dof_Madr, const bool on, const int k, const int nM, const int nv, const int q) {
  const int Mk = dof_Madr[k], d = on ? (k + 1 < nv ? dof_Madr[k + 1] : nM) - Mk - 1 : 0;
  const float dk = qLD[Mk], aq = qLD[Mk + min(q, d)];
  float ap[PMAX], tg[PMAX]; int am[PMAX];
#pragma unroll
  for (int p = 1; p <= PMAX; p++) { const int pp = min(p, d); ap[p-1] = qLD[Mk + pp]; am[p-1] = ANC_MADR(anc[Mk + pp]); }
  const float inv = 1.0f / dk;
#pragma unroll
  for (int p = 1; p <= PMAX; p++) { const bool v = p <= d && q >= p && q <= d; tg[p-1] = qLD[v ? am[p-1] + (q - p) : Mk]; }
#pragma unroll
  for (int p = 1; p <= PMAX; p++) { const bool v = p <= d && q >= p && q <= d; if (v) qLD[am[p-1] + (q - p)] = tg[p-1] - ap[p-1] * inv * aq; }
  if (on && q <= d) qLD[Mk + q] = aq * inv;
  if (on && q == 1) qLDinv[k] = inv;
}
template <class TA, class TN>
DEV void factor_trees_short(float* qLD, float* qLDinv, const int* anc, const int* dof_Madr, const TA& tree_dofadr, const TN& tree_dofnum,
                            const int ntree, const int nM, const int nv, const int lane) {
  const int q = 1 + (lane & 15), row = lane >> 4;
  for (int t0 = 0; t0 < ntree; t0 += 4) {
    const int t = t0 + row;
    int adr = 0, num = 0;
    if (t < ntree) { num = tree_dofnum[t]; adr = tree_dofadr[t]; if (num >= MJH_WAVE_TREE_MIN) num = 0; }
    int steps = 0;
    { const int n0 = __builtin_amdgcn_readlane(num, 0), n1 = __builtin_amdgcn_readlane(num, 16), n2 = __builtin_amdgcn_readlane(num, 32), n3 = __builtin_amdgcn_readlane(num, 48);
      steps = max(max(n0, n1), max(n2, n3)); }
    for (int st = 0; st < steps; st++) {
      const bool on = st < num;
      const int k = on ? adr + num - 1 - st : 0;
      // (a dof of a short tree with `num` dofs has at most num - 1 ancestors: the later steps of a group are shallow)
      const int deep = steps - 1 - st;
      if (deep <= 3) factor_short_step<3>(qLD, qLDinv, anc, dof_Madr, on, k, nM, nv, q);
      else if (deep <= 6) factor_short_step<6>(qLD, qLDinv, anc, dof_Madr, on, k, nM, nv, q);
      else factor_short_step<MJH_WAVE_TREE_MIN - 2>(qLD, qLDinv, anc, dof_Madr, on, k, nM, nv, q);
      __syncthreads();
    }
  }
}
// Solves with the factor, lanes = dofs (nv <= 128: two per lane), level by level instead of dof by dof: a dof with d ancestors
// sits at depth d, and all dofs of one depth are independent of each other.  One wavefront.
struct TreeLanes { int Mk[2], dep[2]; };
DEV TreeLanes tree_lanes(const int* dof_Madr, const int nv, const int nM, const int lane) {
  TreeLanes t;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int k = lane + 64 * h;
    t.Mk[h] = k < nv ? dof_Madr[k] : 0;
    t.dep[h] = k < nv ? (k + 1 < nv ? dof_Madr[k + 1] : nM) - t.Mk[h] - 1 : 0;
  }
  return t;
}
// x <- L^-1 x: a dof's value needs its ancestors' final values: levels from the roots down, every dof pulls
DEV void tree_l_levels(float* x, const float* qLD, const int* anc, const int* dof_Madr, const int nv, const int nM, const int lane) {
  const TreeLanes t = tree_lanes(dof_Madr, nv, nM, lane);
  for (int lev = 1; __ballot(t.dep[0] >= lev || t.dep[1] >= lev) != 0; lev++) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (t.dep[h] == lev) {
        const int k = lane + 64 * h; float xk = x[k];
        for (int p = 1; p <= lev; p++) xk -= qLD[t.Mk[h] + p] * x[ANC_DOF(anc[t.Mk[h] + p])];
        x[k] = xk;
      }
    }
    __syncthreads();
  }
}
// x <- L^-T x: a dof's final value goes to all its ancestors: levels from the leaves up, every dof pushes (LDS atomics: dofs
// of one depth on different branches share ancestors; one wavefront issues them in a fixed order)
DEV void tree_lt_levels(float* x, const float* qLD, const int* anc, const int* dof_Madr, const int nv, const int nM, const int lane) {
  const TreeLanes t = tree_lanes(dof_Madr, nv, nM, lane);
  int maxd = 0;
  while (__ballot(t.dep[0] > maxd || t.dep[1] > maxd) != 0) maxd++;
  for (int lev = maxd; lev >= 1; lev--) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (t.dep[h] == lev) {
        const float xk = x[lane + 64 * h];
        if (xk != 0.0f) for (int p = 1; p <= lev; p++) atomicAdd(&x[ANC_DOF(anc[t.Mk[h] + p])], -qLD[t.Mk[h] + p] * xk);
      }
    }
    __syncthreads();
  }
}
// x <- M^-1 x = L^-1 D^-1 L^-T x for every tree of the model at once
DEV void solve_trees_levels(float* x, const float* qLD, const float* qLDinv, const int* anc, const int* dof_Madr, const int nv, const int nM, const int lane) {
  tree_lt_levels(x, qLD, anc, dof_Madr, nv, nM, lane);
  for (int k = lane; k < nv; k += 64) x[k] *= qLDinv[k];
  __syncthreads();
  tree_l_levels(x, qLD, anc, dof_Madr, nv, nM, lane);
}
DEV void solve_tree_wave(float* x, const float* qLD, const float* qLDinv, const int* anc, const int* dof_Madr, int adr, int num, int nM, int nv, const int lane) {
  for (int k = adr + num - 1; k >= adr; k--) {                                // x <- L^-T x
    const float xk = x[k];
    if (xk != 0) {
      const int Mk = dof_Madr[k], d = (k + 1 < nv ? dof_Madr[k + 1] : nM) - Mk - 1;
      for (int p = 1 + lane; p <= d; p += 64) x[ANC_DOF(anc[Mk + p])] -= qLD[Mk + p] * xk;
    }
    __syncthreads();
  }
  for (int k = adr + lane; k < adr + num; k += 64) x[k] *= qLDinv[k];          // D^-1
  __syncthreads();
  for (int k = adr; k < adr + num; k++) {                                     // x <- L^-1 x
    const int Mk = dof_Madr[k], d = (k + 1 < nv ? dof_Madr[k + 1] : nM) - Mk - 1;
    float part = 0;
    for (int p = 1 + lane; p <= d; p += 64) part += qLD[Mk + p] * x[ANC_DOF(anc[Mk + p])];
    part = wave_sum<4>(part);
    if (lane == 0) x[k] -= part;
    __syncthreads();
  }
}

// row header: hd[0] = type | sub << 8, hd[1] = id, hd[2] = a1 | n1 << 16, hd[3] = a2 | n2 << 16
// where (a1,n1) ++ (a2,n2) are the contiguous dof ranges of the (up to two) kinematic trees the row touches
#define ROW_TREES(hd2, hd3) const int a1 = (hd2) & 0xffff, n1 = (hd2) >> 16, a2 = (hd3) & 0xffff, n2 = (hd3) >> 16
// Jacobian storage of a block: hd.x >> 16 = offset (in float4 units) into the J / B pools; a contact block keeps its
// (up to) 4 base rows interleaved J[k][4], every other block is a single row J[k]
#define BLK_JOFF(hx) ((((unsigned)(hx)) >> 16) << 2)
#define BLK_SLOTS(hy) (((((hy) >> 24) & 15) == RT_CONTACT) ? 4 : 1)
// offset of dof d inside the compact storage of a row spanning trees (a1,n1) ++ (a2,n2); -1 if outside
DEV int row_off(int d, int a1, int n1, int a2, int n2) {
  unsigned r1 = (unsigned)(d - a1), r2 = (unsigned)(d - a2);
  return r1 < (unsigned)n1 ? (int)r1 : (r2 < (unsigned)n2 ? n1 + (int)r2 : -1);
}


// Row-space solver data of a block (written once per step by the "AR" pass, HISTORY.md §5):
//   Q[16] (block floats 16..31) and, for models with condim-4 contacts, X[12] (s_ext):
//     1/AR_rr      : r < 4 -> Q[r]     ; r = 4,5 -> Q[14], Q[15]
//     AR_rr / 2    : r < 4 -> Q[4 + r] ; r = 4,5 -> X[0], X[1]
//     AR_rs, r < s : 01 02 03 12 13 23 -> Q[8..13] ; 04 05 14 15 24 25 34 35 45 -> X[2..10]
//   (AR = E A_c E^T + R I over the block's rows e_r = e_n +- e_k; a block with fewer rows has zeros in the unused
//   slots, which makes the extra unrolled rows of a wider template inert.)
#define SOLQ_N 16
#define SOLX_N 12
DEV constexpr int ar_off_slot(int r, int s) {   // r < s; < 16: Q index, >= 16: 16 + X index
  return r == 0 ? (s <= 3 ? 7 + s : 16 + s - 2) : r == 1 ? (s <= 3 ? 9 + s : 16 + s) : r == 2 ? (s == 3 ? 13 : 16 + s + 2) : r == 3 ? 16 + s + 4 : 16 + 10;
}
#define AR_INV(Q, X, r) ((r) < 4 ? (Q)[r] : (Q)[10 + (r)])
#define AR_HALF(Q, X, r) ((r) < 4 ? (Q)[4 + (r)] : (X)[(r) - 4])
#define AR_OFF(Q, X, r, s) (ar_off_slot(r, s) < 16 ? (Q)[ar_off_slot(r, s) & 15] : (X)[ar_off_slot(r, s) & 15])
// The NR rows of one block, Gauss-Seidel in row space.  u[j] = (J_base a - aref)_j on entry.  Everything here is
// uniform over the lanes that own the block.  Returns the cost decrease; dphi = E^T delta (per base).
template <int NB, int NR>
DEV float pgs_rows(const float R, const float lo, const float hi, const float* u, float* f, const float* Q, const float* X, float* dphi) {
  float res[NR], dl[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int k = NB > 1 ? 1 + (r >> 1) : 0;
    res[r] = R * f[r] + (NB > 1 ? ((r & 1) ? u[0] - u[k] : u[0] + u[k]) : u[0]);
  }
  float imp = 0;
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const float fn = __builtin_amdgcn_fmed3f(f[r] - res[r] * AR_INV(Q, X, r), lo, hi);
    const float delta = fn - f[r];
    // cost change delta*(res + delta*AR/2) <= 0 for every projected scalar update (HISTORY.md §5), so the
    // reference's "revert if the cost went up" guard is dead code and is not evaluated here
    imp -= delta * (res[r] + AR_HALF(Q, X, r) * delta);
#pragma unroll
    for (int q = r + 1; q < NR; q++) res[q] += AR_OFF(Q, X, r, q) * delta;
    f[r] = fn; dl[r] = delta;
  }
  dphi[0] = dl[0];
#pragma unroll
  for (int r = 1; r < NR; r++) dphi[0] += dl[r];
#pragma unroll
  for (int k = 1; k < NB; k++) dphi[k] = dl[2*k - 2] - dl[2*k - 1];
  return imp;
}
// The same block in a noslip sweep (mj_solNoSlip; model/ontology/scene.xml:2-3): friction dimensions only, without the
// regulariser.  u[j] = (J_base a - aref)_j as above (no R f term).  A dof friction-loss row is re-solved with
// A_rr - R; each opposing pair of pyramid edges (2k-2, 2k-1) is re-solved along f_j - f_q with f_j + f_q held;
// equality / limit / frictionless rows are left alone (dphi = 0).  Same definition as the oracle.
template <int NB, int NR>
DEV float noslip_rows(const float R, const float lo, const float hi, const float* u, float* f, const float* Q, const float* X, float* dphi) {
  float imp = 0;
  // (the four-row sweep runs single-row blocks through the widest template of their group: same test there)
  if (NB == 1 || (lo < 0.0f && lo > -1.0e38f)) {
#pragma unroll
    for (int k = 0; k < NB; k++) dphi[k] = 0;
    const float A = 2.0f * AR_HALF(Q, X, 0) - R;
    if (lo < 0.0f && lo > -1.0e38f && A > MJ_MINVAL) {      // [-frictionloss, frictionloss]
      const float fn = __builtin_amdgcn_fmed3f(f[0] - u[0] / A, lo, hi), delta = fn - f[0];
      imp = -delta * (u[0] + 0.5f * A * delta);
      f[0] = fn; dphi[0] = delta;
    }
    return imp;
  }
  float res[NR], dy[NR / 2 + 1];
#pragma unroll
  for (int r = 0; r < NR; r++) { const int k = 1 + (r >> 1); res[r] = (r & 1) ? u[0] - u[k] : u[0] + u[k]; }
#pragma unroll
  for (int p = 0; p < NR; p += 2) {
    const float K1 = 2.0f * (AR_HALF(Q, X, p) + AR_HALF(Q, X, p + 1) - R - AR_OFF(Q, X, p, p + 1));
    const float mid = 0.5f * (f[p] + f[p + 1]), y0 = 0.5f * (f[p] - f[p + 1]), dr = res[p] - res[p + 1];
    const float y = K1 > MJ_MINVAL ? __builtin_amdgcn_fmed3f(y0 - dr / K1, -mid, mid) : y0;
    const float d = y - y0;
    imp -= d * (dr + 0.5f * K1 * d);
    f[p] = mid + y; f[p + 1] = mid - y; dy[p / 2] = d;
#pragma unroll
    for (int q = p + 2; q < NR; q++) res[q] += (AR_OFF(Q, X, p, q) - AR_OFF(Q, X, p + 1, q)) * d;
  }
  dphi[0] = 0;
#pragma unroll
  for (int k = 1; k < NB; k++) dphi[k] = 2.0f * dy[k - 1];
  return imp;
}
// row math of one block in the current sweep mode (ns: noslip sweep; only the EXTRA kernel instances carry that code)
template <int NB, int NR, bool EXTRA>
DEV float solve_rows(const bool ns, const float R, const float lo, const float hi, const float* u, float* f, const float* Q, const float* X, float* dphi) {
  if (EXTRA && ns) return noslip_rows<NB, NR>(R, lo, hi, u, f, Q, X, dphi);
  return pgs_rows<NB, NR>(R, lo, hi, u, f, Q, X, dphi);
}
// One block of the single-block sweep (nv > 32): NB full-wave reductions, then uniform row math.
template <int NB, int NR, int NROW, bool EXTRA>
DEV void pgs_block(const bool ns, const float R, const float lo, const float hi, const float* ab, float* f, const float* Q, const float* X, const float* Jd, const float* Bd,
                   const float bscale, float& a, float& improvement) {
  float u[4] = {Jd[0] * a, NB > 1 ? Jd[1] * a : 0.0f, NB > 2 ? Jd[2] * a : 0.0f, NB > 3 ? Jd[3] * a : 0.0f}, dphi[4];
  constexpr int WR = NROW > 4 ? 4 : NROW;   // NROW 8 (nv > 64): full-wave sums over the block's compact dofs
  if (NB == 1) u[0] = wave_sum<WR>(u[0]); else wave_sum4<WR, NB>(u);
#pragma unroll
  for (int j = 0; j < NB; j++) u[j] -= ab[j];
  improvement += solve_rows<NB, NR, EXTRA>(ns, R, lo, hi, u, f, Q, X, dphi);
  float da = Bd[0] * dphi[0];
#pragma unroll
  for (int j = 1; j < NB; j++) da += Bd[j] * dphi[j];
  a += da * bscale;
}


// Keeps every component of a loaded vector formally alive until this point.  Without it the register allocator
// reuses the unused components of an in-flight ds_read_b128 as scratch, and the resulting write-after-write
// hazard makes the compiler wait for the load right after issuing it (defeating the software pipeline).
#define KEEP4(v) asm volatile("" :: "v"((v).x), "v"((v).y), "v"((v).z), "v"((v).w))
// ---- dual-block PGS (wavefronts with nv <= 32): the two 32-lane halves of the wave solve the two blocks of an
// independent pair at the same time.  Every "uniform" quantity of the single-block solver becomes uniform
// per half; a half-wide sum is a 4-step DPP butterfly inside each 16-lane row plus one v_permlane16_swap.
DEV void swap16_sum4(float* v, const int n) {   // v[j] <- v[j] + (v[j] of the partner 16-lane row), j < n
  float y0 = v[0], y1 = v[1], y2 = v[2], y3 = v[3];
  if (n == 4) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                           : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
  else if (n == 3) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %3\n\tv_permlane16_swap_b32 %1, %4\n\tv_permlane16_swap_b32 %2, %5\n\ts_nop 1"
                                : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(y0), "+v"(y1), "+v"(y2));
  else asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(v[0]), "+v"(y0));
  v[0] += y0; if (n > 1) v[1] += y1; if (n > 2) v[2] += y2; if (n > 3) v[3] += y3;
}
DEV float swap32_sum(float x) {                  // x + (x of the other 32-lane half)
  float y = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  return x + y;
}
#define MJH_DPP_BF4(v, N, ctrl) do { MJH_DPP_ADD(v[0], ctrl, 0xf, true); if (N > 1) MJH_DPP_ADD(v[1], ctrl, 0xf, true); \
    if (N > 2) MJH_DPP_ADD(v[2], ctrl, 0xf, true); if (N > 3) MJH_DPP_ADD(v[3], ctrl, 0xf, true); } while (0)
template <int N> DEV void half_sum4(float* v, const int lq) {   // sums over each 32-lane half, result in every lane of the half (lq = lane & 3)
  MJH_DPP_BF4(v, N, 0xB1);    // quad_perm [1,0,3,2]
  MJH_DPP_BF4(v, N, 0x4E);    // quad_perm [2,3,0,1]   -> every lane holds its quad's sum of each component
  if (N == 1) {
    MJH_DPP_BF4(v, N, 0x141);   // row_half_mirror
    MJH_DPP_BF4(v, N, 0x140);   // row_mirror
    swap16_sum4(v, N);
    return;
  }
  // transpose: lane (4q + j) carries component j from here on, so the cross-quad part runs on ONE register
  float w = v[0];
  w = lq == 1 ? v[1] : w;
  if (N > 2) w = lq == 2 ? v[2] : w;
  if (N > 3) w = lq == 3 ? v[3] : w;
  MJH_DPP_ADD(w, 0x124, 0xf, true);   // row_ror:4  (rotations by whole quads keep lane & 3)
  MJH_DPP_ADD(w, 0x128, 0xf, true);   // row_ror:8  -> sum over the 4 quads of the 16-lane row
  swap16_sum4(&w, 1);                 // + the partner row of the half
  // broadcast component j from lane 4q + j to its quad
#define MJH_QBCAST(j) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, w), (j) * 0x55, 0xf, 0xf, true))
  v[0] = MJH_QBCAST(0); v[1] = MJH_QBCAST(1);
  if (N > 2) v[2] = MJH_QBCAST(2);
  if (N > 3) v[3] = MJH_QBCAST(3);
#undef MJH_QBCAST
}
template <int NB, int NR, bool EXTRA>
DEV float pgs_dual(const bool ns, const float R, const float lo, const float hi, const float* ab, float* f, const float* Q, const float* X, const float* Jd, const float* Bd,
                   const float bscale, const int lq, float& a) {
  float u[4] = {Jd[0] * a, NB > 1 ? Jd[1] * a : 0.0f, NB > 2 ? Jd[2] * a : 0.0f, NB > 3 ? Jd[3] * a : 0.0f}, dphi[4];
  // (opaque products: otherwise u + dpp(u) is contracted to fma(J, a, dpp(u)), which costs an extra v_mov_dpp per base)
  asm volatile("" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]));
  half_sum4<NB>(u, lq);
#pragma unroll
  for (int j = 0; j < NB; j++) u[j] -= ab[j];
  const float imp = solve_rows<NB, NR, EXTRA>(ns, R, lo, hi, u, f, Q, X, dphi);
  float da = Bd[0] * dphi[0];
#pragma unroll
  for (int j = 1; j < NB; j++) da += Bd[j] * dphi[j];
  a += swap32_sum(da * bscale);       // the pair's blocks touch disjoint dofs: both halves end up with the same `a`
  return imp;
}

// ---- many-body sweep (nv > 64 or pools in global memory): the running acceleration lives in LDS (c.qacc), lanes = the
// compact dofs of ONE block (at most 64: rowW), gathered before and scattered after the block's update; the block's
// operands come from the pools and are prefetched one block ahead.  Shared by the fused step kernel and mjh_solve_kernel.
struct ManyCtx {
  const float *J, *B, *blkq, *ext; float* blkf; const int *blki, *order, *gstart;   // order / gstart: LDS copies (visiting order, group starts)
  int ngrp;
  float* qacc; const float* qLDinv;
  int nblk, nfixblk, rowW, iterations; bool has_dim4; float scale, tolerance;
  int noslip_iterations; float noslip_tolerance;     // noslip post-pass (EXTRA instances only)
  int nwave, wid; float* red;                        // mjh_solve_kernel with wide groups: waves per environment, this wave, LDS partial sums
  bool order_packed;    // the LDS copy of the order carries block | kind << 16 | ndof << 20: the fetch skips what the block does not have
  // mjh_solve_kernel (BUF instances): every pool is inside the env's global slice, addressed through ONE buffer descriptor with the
  // pool's byte offset as the scalar offset and a 32-bit per-lane offset - one shift per block instead of a 64-bit address per load,
  // and a lane that has nothing to fetch / store points beyond the slice (reads 0, stores nothing) instead of being masked off
  __amdgpu_buffer_rsrc_t rsrc; int oJ, oB, oblkf, oblkq, oext, oblki;
  // quad sweep (free-body piles, groups of up to 16 blocks): padded LDS copies of the running acceleration and of 1 / M_dd, four floats
  // per 3 dofs (one ds_read_b128 per lane); null: not used
  float* a4; const float* m4;
};
typedef unsigned mjh_v4u __attribute__((vector_size(16)));   // (the builtins' own vector types: an ext_vector_type of the same size converts by value, not by bits)
typedef unsigned mjh_v2u __attribute__((vector_size(8)));
#define MJH_BUF_OOB 0x7ffffff0u
DEV float4 buf_load4(const __amdgpu_buffer_rsrc_t r, const unsigned voff, const int soff) {
  const mjh_v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
  float4 f; __builtin_memcpy(&f, &v, 16);
  return f;
}
// LDS order word of block b with header hd (kind: BK_*, ndof: dofs of the one or two trees it touches)
#define MJH_ORDER_WORD(b, hd) ((b) | (((hd).x & 15) << 16) | (((((unsigned)(hd).z) >> 16) + ((((unsigned)(hd).w) >> 16))) << 20))
template <bool DIAGM, bool EXTRA, bool BUF = false>
DEV int pgs_many_body(const ManyCtx& c, const int lane) {
  int niter = 0, nmain = 0;
  // a four-wave workgroup (mjh_solve_kernel with wide groups) on an environment whose sweep is sequential (few blocks): wave 0
  // works alone and meets the others at the kernel's final barrier; no workgroup barrier may then be executed in here
  const bool wide = c.nblk > 64 && c.rowW <= 16 && c.ngrp >= 2;
  const bool solo = c.nwave > 1 && !wide;
  if (solo && c.wid != 0) return 0;
  // sweep mode: the main PGS sweeps, then (EXTRA instances, option noslip_iterations) the noslip sweeps over the same
  // schedule with the friction-only row math
  bool ns = false;
  int itmax = c.iterations; float tol = c.tolerance;
  const int nmode = (EXTRA && c.noslip_iterations > 0) ? 2 : 1;
  for (int mode = 0; mode < nmode; mode++) {
  if (mode == 1) { ns = true; nmain = niter; niter = 0; itmax = c.noslip_iterations; tol = c.noslip_tolerance; if (!solo) __syncthreads(); }
  // (the side-by-side forms below sum a sweep's cost decrease as fixed-point integers: the total must not depend on which lane
  //  carried which block — dev_math.h: imp_fixed)
  const ImpQ iq = imp_quantum(c.scale, tol, c.nblk);
  // operands of one block; every address follows from the block index alone (contact blocks are laid out
  // regularly behind the c.nfixblk non-contact ones), so all loads of block k+1 are in flight while block k is solved
  struct MOp { int4 hd; int b; float4 J, B, p0, r0, r1, r2, A0, A1, A2, A3, X0, X1, X2; };
  auto blockAt = [&](int k) __attribute__((always_inline)) { return c.order_packed ? (c.order[k] & 0xffff) : c.order[k]; };   // visiting order (LDS copy)
  auto fetch8 = [&](int b) __attribute__((always_inline)) {
    MOp op; op.b = b;
    const bool quad = b >= c.nfixblk;
    const int jo = (quad ? c.nfixblk + 4 * (b - c.nfixblk) : b) * c.rowW;      // = BLK_JOFF(hd.x), see put_block
    if constexpr (BUF) {
      const unsigned ob64 = (unsigned)b * 64u;
      { const mjh_v4u h = __builtin_amdgcn_raw_buffer_load_b128(c.rsrc, (int)((unsigned)b * 16u), c.oblki, 0); __builtin_memcpy(&op.hd, &h, 16); }
      if (quad) {
        const unsigned oj = lane < c.rowW ? (unsigned)(jo + 4 * lane) * 4u : MJH_BUF_OOB;
        op.J = buf_load4(c.rsrc, oj, c.oJ);
        op.B = DIAGM ? op.J : buf_load4(c.rsrc, oj, c.oB);
      } else {
        const unsigned oj = lane < c.rowW ? (unsigned)(jo + lane) * 4u : MJH_BUF_OOB;
        op.J = make_float4(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rsrc, (int)oj, c.oJ, 0)), 0, 0, 0);
        op.B = DIAGM ? op.J : make_float4(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rsrc, (int)oj, c.oB, 0)), 0, 0, 0);
      }
      op.p0 = buf_load4(c.rsrc, ob64, c.oblkf); op.r0 = buf_load4(c.rsrc, ob64 + 16u, c.oblkf); op.r1 = buf_load4(c.rsrc, ob64 + 32u, c.oblkf); op.r2 = buf_load4(c.rsrc, ob64 + 48u, c.oblkf);
      op.A0 = buf_load4(c.rsrc, ob64, c.oblkq); op.A1 = buf_load4(c.rsrc, ob64 + 16u, c.oblkq); op.A2 = buf_load4(c.rsrc, ob64 + 32u, c.oblkq); op.A3 = buf_load4(c.rsrc, ob64 + 48u, c.oblkq);
      op.X0 = make_float4(0, 0, 0, 0); op.X1 = op.X0; op.X2 = op.X0;
      if (c.has_dim4) { const unsigned ox = (unsigned)b * (unsigned)(SOLX_N * 4); op.X0 = buf_load4(c.rsrc, ox, c.oext); op.X1 = buf_load4(c.rsrc, ox + 16u, c.oext); op.X2 = buf_load4(c.rsrc, ox + 32u, c.oext); }
      return op;
    }
    op.hd = ((const int4*)c.blki)[b];
    op.J = make_float4(0, 0, 0, 0); op.B = op.J;
    if (lane < c.rowW) {                                                  // dofs beyond the block's trees hold zeros
      if (quad) op.J = *(const float4*)(c.J + jo + 4*lane); else op.J.x = c.J[jo + lane];
      if (!DIAGM) { if (quad) op.B = *(const float4*)(c.B + jo + 4*lane); else op.B.x = c.B[jo + lane]; }
    }
    op.p0 = ((const float4*)c.blkf)[4*b]; op.r0 = ((const float4*)c.blkf)[4*b+1]; op.r1 = ((const float4*)c.blkf)[4*b+2]; op.r2 = ((const float4*)c.blkf)[4*b+3];
    op.A0 = ((const float4*)c.blkq)[4*b]; op.A1 = ((const float4*)c.blkq)[4*b+1]; op.A2 = ((const float4*)c.blkq)[4*b+2]; op.A3 = ((const float4*)c.blkq)[4*b+3];
    op.X0 = make_float4(0, 0, 0, 0); op.X1 = op.X0; op.X2 = op.X0;
    if (c.has_dim4) { const float4* x4 = (const float4*)(c.ext + b * SOLX_N); op.X0 = x4[0]; op.X1 = x4[1]; op.X2 = x4[2]; }
    return op;
  };
  auto process8 = [&](MOp& op, float& improvement) __attribute__((always_inline)) {
    KEEP4(op.hd); KEEP4(op.J); KEEP4(op.p0); KEEP4(op.r0); KEEP4(op.r1); KEEP4(op.r2); KEEP4(op.A0); KEEP4(op.A1); KEEP4(op.A2); KEEP4(op.A3);
    if (!DIAGM) KEEP4(op.B);
    if (c.has_dim4) { KEEP4(op.X0); KEEP4(op.X1); KEEP4(op.X2); }
    ROW_TREES(op.hd.z, op.hd.w);
    const bool on = lane < n1 + n2;
    const int d = lane < n1 ? a1 + lane : a2 + lane - n1;
    float ak = on ? c.qacc[d] : 0.0f;                                    // gather
    const int kind = __builtin_amdgcn_readfirstlane(op.hd.x & 15);
    float f[6] = {op.r1.x, op.r1.y, op.r1.z, op.r1.w, op.r2.x, op.r2.y};
    const float aref[4] = {op.r0.x, op.r0.y, op.r0.z, op.r0.w};
    const float Q[16] = {op.A0.x, op.A0.y, op.A0.z, op.A0.w, op.A1.x, op.A1.y, op.A1.z, op.A1.w,
                         op.A2.x, op.A2.y, op.A2.z, op.A2.w, op.A3.x, op.A3.y, op.A3.z, op.A3.w};
    const float X[12] = {op.X0.x, op.X0.y, op.X0.z, op.X0.w, op.X1.x, op.X1.y, op.X1.z, op.X1.w, op.X2.x, op.X2.y, op.X2.z, op.X2.w};
    const float Jd[4] = {on ? op.J.x : 0.0f, on ? op.J.y : 0.0f, on ? op.J.z : 0.0f, on ? op.J.w : 0.0f};
    const float Bd[4] = {on ? op.B.x : 0.0f, on ? op.B.y : 0.0f, on ? op.B.z : 0.0f, on ? op.B.w : 0.0f};
    const float* Bp = DIAGM ? Jd : Bd;
    const float bs = DIAGM ? (on ? c.qLDinv[d] : 0.0f) : 1.0f;
    const float R = op.p0.x, lo = op.r2.z, hi = op.r2.w;
    if (kind == BK_PYR4) pgs_block<4, 6, 8, EXTRA>(ns, R, lo, hi, aref, f, Q, X, Jd, Bp, bs, ak, improvement);
    else if (kind == BK_PYR3) pgs_block<3, 4, 8, EXTRA>(ns, R, lo, hi, aref, f, Q, X, Jd, Bp, bs, ak, improvement);
    else pgs_block<1, 1, 8, EXTRA>(ns, R, lo, hi, aref, f, Q, X, Jd, Bp, bs, ak, improvement);
    if (on) c.qacc[d] = ak;                                              // scatter
    if constexpr (BUF) {
      const unsigned of = lane == 0 ? (unsigned)op.b * 64u + (unsigned)(BF_F * 4) : MJH_BUF_OOB;
      mjh_v4u f4; mjh_v2u f2;
      __builtin_memcpy(&f4, f, 16); __builtin_memcpy(&f2, f + 4, 8);
      __builtin_amdgcn_raw_buffer_store_b128(f4, c.rsrc, (int)of, c.oblkf, 0);
      __builtin_amdgcn_raw_buffer_store_b64(f2, c.rsrc, (int)(of + 16u), c.oblkf, 0);
    } else
    if (lane == 0) {
      float* bf = c.blkf + op.b * BLKF_STRIDE + BF_F;
      *(float4*)(bf) = make_float4(f[0], f[1], f[2], f[3]);
      *(float2*)(bf + 4) = make_float2(f[4], f[5]);
    }
  };
  if (BUF && DIAGM && wide && c.a4 != nullptr && c.nwave == 1) {
    // ======== SIXTEEN blocks per wave-step (free-body piles: BASELINE config C2): a whole group of up to 16 mutually independent
    //          blocks at once, four lanes per block — lane (s, h) of the wave owns dofs 3h .. 3h + 2 of the compact dofs of block
    //          s of the group (h = 0, 1: body 1 translation / rotation; h = 2, 3: body 2).  The four-blocks-per-step form above
    //          spends ~150 instructions per step doing the uniform row math redundantly on the 16 lanes of each block; here the
    //          redundancy is 4 and a sweep over 200 blocks is ~14 steps instead of ~55.  Same groups, same order, same row math
    //          (solve_rows) as the other forms and the oracle.  u_j = sum over the lane's 3 dofs, then two quad_perm adds.
    const int s = lane >> 2, h = lane & 3;
    struct SOp { int4 hd; int b; float act; float4 J0, J1, J2, p0, r0, r1, r2, A0, A1, A2, A3, X0, X1, X2; };
    auto fetchS = [&](int g) __attribute__((always_inline)) {
      SOp op;
      const int st = c.gstart[g], sz = c.gstart[g + 1] - st;
      const bool act = s < sz;
      const int ow = c.order[act ? st + s : st];
      const int b = ow & 0xffff, okind = (ow >> 16) & 15, ondof = ow >> 20;
      op.b = b; op.act = act ? 1.0f : 0.0f;
      const bool quad = b >= c.nfixblk;
      const int jo = (quad ? c.nfixblk + 4 * (b - c.nfixblk) : b) * c.rowW;
      const unsigned ob64 = (unsigned)b * 64u;
      { const mjh_v4u hh = __builtin_amdgcn_raw_buffer_load_b128(c.rsrc, (int)((unsigned)b * 16u), c.oblki, 0); __builtin_memcpy(&op.hd, &hh, 16); }
      const bool jon = act && 3 * h < ondof;
      if (quad) {
        const unsigned oj = jon ? (unsigned)(jo + 12 * h) * 4u : MJH_BUF_OOB;
        op.J0 = buf_load4(c.rsrc, oj, c.oJ); op.J1 = buf_load4(c.rsrc, oj + 16u, c.oJ); op.J2 = buf_load4(c.rsrc, oj + 32u, c.oJ);
      } else {
        const unsigned oj = jon ? (unsigned)(jo + 3 * h) * 4u : MJH_BUF_OOB;
        op.J0 = make_float4(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rsrc, (int)oj, c.oJ, 0)), 0, 0, 0);
        op.J1 = make_float4(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rsrc, (int)(oj + 4u), c.oJ, 0)), 0, 0, 0);
        op.J2 = make_float4(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rsrc, (int)(oj + 8u), c.oJ, 0)), 0, 0, 0);
      }
      const unsigned ob = act ? ob64 : MJH_BUF_OOB;
      op.p0 = buf_load4(c.rsrc, ob, c.oblkf); op.r0 = buf_load4(c.rsrc, ob + 16u, c.oblkf); op.r1 = buf_load4(c.rsrc, ob + 32u, c.oblkf); op.r2 = buf_load4(c.rsrc, ob + 48u, c.oblkf);
      op.A0 = buf_load4(c.rsrc, ob, c.oblkq); op.A1 = buf_load4(c.rsrc, ob + 16u, c.oblkq); op.A2 = buf_load4(c.rsrc, ob + 32u, c.oblkq); op.A3 = buf_load4(c.rsrc, ob + 48u, c.oblkq);
      op.X0 = make_float4(0, 0, 0, 0); op.X1 = op.X0; op.X2 = op.X0;
      if (c.has_dim4) {
        const unsigned ox = (okind == BK_PYR4 && act) ? (unsigned)b * (unsigned)(SOLX_N * 4) : MJH_BUF_OOB;
        op.X0 = buf_load4(c.rsrc, ox, c.oext); op.X1 = buf_load4(c.rsrc, ox + 16u, c.oext); op.X2 = buf_load4(c.rsrc, ox + 32u, c.oext);
      }
      return op;
    };
    auto processS = [&](SOp& op, int& impl) __attribute__((always_inline)) {
      KEEP4(op.hd); KEEP4(op.J0); KEEP4(op.J1); KEEP4(op.J2); KEEP4(op.p0); KEEP4(op.r0); KEEP4(op.r1); KEEP4(op.r2); KEEP4(op.A0); KEEP4(op.A1); KEEP4(op.A2); KEEP4(op.A3);
      if (c.has_dim4) { KEEP4(op.X0); KEEP4(op.X1); KEEP4(op.X2); }
      ROW_TREES(op.hd.z, op.hd.w);
      const bool actb = op.act > 0.0f;
      const bool on = actb && 3 * h < n1 + n2;
      // triple index of the lane's dofs in the padded vectors: dofs a1 + 3h (h < n1 / 3) or a2 + 3h - n1  (x / 3 = x * 43691 >> 17)
      const int d0 = 3 * h < n1 ? a1 + 3 * h : a2 + 3 * h - n1;
      const int t4 = on ? (int)(((unsigned)d0 * 43691u) >> 17) << 2 : 0;
      const float4 av = on ? *(const float4*)(c.a4 + t4) : make_float4(0, 0, 0, 0);
      const float4 mv = on ? *(const float4*)(c.m4 + t4) : make_float4(0, 0, 0, 0);
      const int kk = actb ? (op.hd.x & 15) : 0;
      const int kind = __ballot(kk == BK_PYR4) ? BK_PYR4 : (__ballot(kk == BK_PYR3) ? BK_PYR3 : BK_SINGLE);    // rows of smaller blocks are inert
      float f[6] = {op.r1.x, op.r1.y, op.r1.z, op.r1.w, op.r2.x, op.r2.y};
      const float aref[4] = {op.r0.x, op.r0.y, op.r0.z, op.r0.w};
      const float Q[16] = {op.A0.x, op.A0.y, op.A0.z, op.A0.w, op.A1.x, op.A1.y, op.A1.z, op.A1.w,
                           op.A2.x, op.A2.y, op.A2.z, op.A2.w, op.A3.x, op.A3.y, op.A3.z, op.A3.w};
      const float X[12] = {op.X0.x, op.X0.y, op.X0.z, op.X0.w, op.X1.x, op.X1.y, op.X1.z, op.X1.w, op.X2.x, op.X2.y, op.X2.z, op.X2.w};
      const float R = op.p0.x, lo = op.r2.z, hi = op.r2.w;
      float u[4] = {op.J0.x * av.x + op.J1.x * av.y + op.J2.x * av.z, op.J0.y * av.x + op.J1.y * av.y + op.J2.y * av.z,
                    op.J0.z * av.x + op.J1.z * av.y + op.J2.z * av.z, op.J0.w * av.x + op.J1.w * av.y + op.J2.w * av.z}, dphi[4] = {0, 0, 0, 0};
      asm volatile("" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]));
      float imp;
#define MJH_QUADSUM(N) do { MJH_DPP_BF4(u, N, 0xB1); MJH_DPP_BF4(u, N, 0x4E); } while (0)     // quad_perm [1,0,3,2], [2,3,0,1]: every lane of the quad holds the block's sum
      if (kind == BK_PYR4) { MJH_QUADSUM(4); for (int j = 0; j < 4; j++) u[j] -= aref[j]; imp = solve_rows<4, 6, EXTRA>(ns, R, lo, hi, u, f, Q, X, dphi); }
      else if (kind == BK_PYR3) { MJH_QUADSUM(3); for (int j = 0; j < 3; j++) u[j] -= aref[j]; imp = solve_rows<3, 4, EXTRA>(ns, R, lo, hi, u, f, Q, X, dphi); }
      else { MJH_QUADSUM(1); u[0] -= aref[0]; imp = solve_rows<1, 1, EXTRA>(ns, R, lo, hi, u, f, Q, X, dphi); }
#undef MJH_QUADSUM
      if (on) {
        float4 an = av;
        an.x += (op.J0.x * dphi[0] + op.J0.y * dphi[1] + op.J0.z * dphi[2] + op.J0.w * dphi[3]) * mv.x;
        an.y += (op.J1.x * dphi[0] + op.J1.y * dphi[1] + op.J1.z * dphi[2] + op.J1.w * dphi[3]) * mv.y;
        an.z += (op.J2.x * dphi[0] + op.J2.y * dphi[1] + op.J2.z * dphi[2] + op.J2.w * dphi[3]) * mv.z;
        *(float4*)(c.a4 + t4) = an;                                           // (the group's blocks touch disjoint bodies)
      }
      const bool head = actb && h == 0;
      impl += imp_fixed(head ? imp : 0.0f, iq.qs, iq.cl);
      const unsigned of = head ? (unsigned)op.b * 64u + (unsigned)(BF_F * 4) : MJH_BUF_OOB;
      mjh_v4u f4; mjh_v2u f2;
      __builtin_memcpy(&f4, f, 16); __builtin_memcpy(&f2, f + 4, 8);
      __builtin_amdgcn_raw_buffer_store_b128(f4, c.rsrc, (int)of, c.oblkf, 0);
      __builtin_amdgcn_raw_buffer_store_b64(f2, c.rsrc, (int)(of + 16u), c.oblkf, 0);
    };
    // operands two groups ahead (three register sets in rotation), as in the four-block form
    int gF = 0, gP = 0;
    auto nextFetch = [&]() __attribute__((always_inline)) { SOp o = fetchS(gF); gF = gF + 1 == c.ngrp ? 0 : gF + 1; return o; };
    int impl = 0; bool done = false;
    auto stepDone = [&]() __attribute__((always_inline)) {
      if (++gP < c.ngrp) return;
      gP = 0; niter++;
      const int improvement = wave_sum_dpp_i(impl);
      impl = 0;
      done = improvement < iq.thr || niter >= itmax;
    };
#ifndef MJH_QUAD_DEPTH
#define MJH_QUAD_DEPTH 3      // operand sets in flight: 3 = two groups ahead (~210 VGPRs, two waves per SIMD): C2 0.658 M env-steps/s; 2 = one ahead (149 VGPRs, three waves per SIMD): 0.618 M — the fetch latency matters more than the third wave
#endif
    if (MJH_QUAD_DEPTH >= 3 && c.ngrp >= 3) {
      SOp o0 = nextFetch(), o1 = nextFetch(), o2;
      while (true) {
        o2 = nextFetch(); processS(o0, impl); stepDone(); if (done) break;
        o0 = nextFetch(); processS(o1, impl); stepDone(); if (done) break;
        o1 = nextFetch(); processS(o2, impl); stepDone(); if (done) break;
      }
    } else {
      SOp o0 = nextFetch(), o1;
      while (true) {
        o1 = nextFetch(); processS(o0, impl); stepDone(); if (done) break;
        o0 = nextFetch(); processS(o1, impl); stepDone(); if (done) break;
      }
    }
  } else
  if (wide) {
    // ======== four blocks per wave-step: the blocks of a group (mutually independent by construction of the order) sit on the
    //          four 16-lane rows of a wave; lanes of a row = the compact dofs of its block.  Everything "uniform" of the
    //          single-block step is uniform per row; a row-wide sum is a 4-step DPP butterfly inside the row.  A group holds
    //          up to 16 blocks: chunk c4 of group g = positions gstart[g] + 4 c4 ...; ONE wave walks the chunks of a group one
    //          after the other (fused kernel); the c.nwave waves of mjh_solve_kernel share them (chunk c4 goes to wave c4 % nwave) and meet at a
    //          workgroup barrier after every group.
    const int row = lane >> 4, l = lane & 15;
    const bool multi = c.nwave > 1;     // (solo is false here)
    struct QOp { int4 hd; int b; float act; float4 J, B, p0, r0, r1, r2, A0, A1, A2, A3, X0, X1, X2; };
    auto fetchQ = [&](int g, int c4) __attribute__((always_inline)) {
      QOp op;
      const int st = c.gstart[g] + 4 * c4, sz = c.gstart[g + 1] - st;       // sz <= 0: this wave has no chunk in this group
      const bool act = row < sz;
      const int ow = c.order[act ? st + row : c.gstart[g]];
      const int b = c.order_packed ? (ow & 0xffff) : ow;
      // what the block has (packed order word) bounds what is fetched: the condim-4 extension for condim-4 blocks only, the
      // Jacobian for the block's own dofs only (a box-floor contact touches 6 of the 12 row slots)
      const int okind = c.order_packed ? ((ow >> 16) & 15) : BK_PYR4, ondof = c.order_packed ? (ow >> 20) : c.rowW;
      op.b = b; op.act = act ? 1.0f : 0.0f;
      const bool quad = b >= c.nfixblk;
      const int jo = (quad ? c.nfixblk + 4 * (b - c.nfixblk) : b) * c.rowW;
      if constexpr (BUF) {
        const unsigned ob64 = (unsigned)b * 64u;
        { const mjh_v4u h = __builtin_amdgcn_raw_buffer_load_b128(c.rsrc, (int)((unsigned)b * 16u), c.oblki, 0); __builtin_memcpy(&op.hd, &h, 16); }
        const bool jon = l < ondof && act;
        if (quad) {
          const unsigned oj = jon ? (unsigned)(jo + 4 * l) * 4u : MJH_BUF_OOB;
          op.J = buf_load4(c.rsrc, oj, c.oJ);
          op.B = DIAGM ? op.J : buf_load4(c.rsrc, oj, c.oB);
        } else {
          const unsigned oj = jon ? (unsigned)(jo + l) * 4u : MJH_BUF_OOB;
          op.J = make_float4(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rsrc, (int)oj, c.oJ, 0)), 0, 0, 0);
          op.B = DIAGM ? op.J : make_float4(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rsrc, (int)oj, c.oB, 0)), 0, 0, 0);
        }
        op.p0 = buf_load4(c.rsrc, ob64, c.oblkf); op.r0 = buf_load4(c.rsrc, ob64 + 16u, c.oblkf); op.r1 = buf_load4(c.rsrc, ob64 + 32u, c.oblkf); op.r2 = buf_load4(c.rsrc, ob64 + 48u, c.oblkf);
        op.A0 = buf_load4(c.rsrc, ob64, c.oblkq); op.A1 = buf_load4(c.rsrc, ob64 + 16u, c.oblkq); op.A2 = buf_load4(c.rsrc, ob64 + 32u, c.oblkq); op.A3 = buf_load4(c.rsrc, ob64 + 48u, c.oblkq);
        op.X0 = make_float4(0, 0, 0, 0); op.X1 = op.X0; op.X2 = op.X0;
        if (c.has_dim4) {
          const unsigned ox = (okind == BK_PYR4 && act) ? (unsigned)b * (unsigned)(SOLX_N * 4) : MJH_BUF_OOB;
          op.X0 = buf_load4(c.rsrc, ox, c.oext); op.X1 = buf_load4(c.rsrc, ox + 16u, c.oext); op.X2 = buf_load4(c.rsrc, ox + 32u, c.oext);
        }
        return op;
      }
      op.hd = ((const int4*)c.blki)[b];
      op.J = make_float4(0, 0, 0, 0); op.B = op.J;
      if (l < ondof && act) {
        if (quad) op.J = *(const float4*)(c.J + jo + 4*l); else op.J.x = c.J[jo + l];
        if (!DIAGM) { if (quad) op.B = *(const float4*)(c.B + jo + 4*l); else op.B.x = c.B[jo + l]; }
      }
      op.p0 = ((const float4*)c.blkf)[4*b]; op.r0 = ((const float4*)c.blkf)[4*b+1]; op.r1 = ((const float4*)c.blkf)[4*b+2]; op.r2 = ((const float4*)c.blkf)[4*b+3];
      op.A0 = ((const float4*)c.blkq)[4*b]; op.A1 = ((const float4*)c.blkq)[4*b+1]; op.A2 = ((const float4*)c.blkq)[4*b+2]; op.A3 = ((const float4*)c.blkq)[4*b+3];
      op.X0 = make_float4(0, 0, 0, 0); op.X1 = op.X0; op.X2 = op.X0;
      if (c.has_dim4 && okind == BK_PYR4 && act) { const float4* x4 = (const float4*)(c.ext + b * SOLX_N); op.X0 = x4[0]; op.X1 = x4[1]; op.X2 = x4[2]; }
      return op;
    };
    auto processQ = [&](QOp& op, int& impl) __attribute__((always_inline)) {
      KEEP4(op.hd); KEEP4(op.J); KEEP4(op.p0); KEEP4(op.r0); KEEP4(op.r1); KEEP4(op.r2); KEEP4(op.A0); KEEP4(op.A1); KEEP4(op.A2); KEEP4(op.A3);
      if (!DIAGM) KEEP4(op.B);
      if (c.has_dim4) { KEEP4(op.X0); KEEP4(op.X1); KEEP4(op.X2); }
      ROW_TREES(op.hd.z, op.hd.w);
      const bool on = op.act > 0.0f && l < n1 + n2;
      const int d = l < n1 ? a1 + l : a2 + l - n1;
      float ak = on ? c.qacc[d] : 0.0f;                                       // gather
      const int kk = op.act > 0.0f ? (op.hd.x & 15) : 0;
      const int k01 = max(__builtin_amdgcn_readlane(kk, 0), __builtin_amdgcn_readlane(kk, 16));
      const int k23 = max(__builtin_amdgcn_readlane(kk, 32), __builtin_amdgcn_readlane(kk, 48));
      const int kind = max(k01, k23);                                          // rows of smaller blocks are inert
      float f[6] = {op.r1.x, op.r1.y, op.r1.z, op.r1.w, op.r2.x, op.r2.y};
      const float aref[4] = {op.r0.x, op.r0.y, op.r0.z, op.r0.w};
      const float Q[16] = {op.A0.x, op.A0.y, op.A0.z, op.A0.w, op.A1.x, op.A1.y, op.A1.z, op.A1.w,
                           op.A2.x, op.A2.y, op.A2.z, op.A2.w, op.A3.x, op.A3.y, op.A3.z, op.A3.w};
      const float X[12] = {op.X0.x, op.X0.y, op.X0.z, op.X0.w, op.X1.x, op.X1.y, op.X1.z, op.X1.w, op.X2.x, op.X2.y, op.X2.z, op.X2.w};
      const float Jd[4] = {on ? op.J.x : 0.0f, on ? op.J.y : 0.0f, on ? op.J.z : 0.0f, on ? op.J.w : 0.0f};
      const float Bd[4] = {on ? op.B.x : 0.0f, on ? op.B.y : 0.0f, on ? op.B.z : 0.0f, on ? op.B.w : 0.0f};
      const float* Bp = DIAGM ? Jd : Bd;
      const float bs = DIAGM ? (on ? c.qLDinv[d] : 0.0f) : 1.0f;
      const float R = op.p0.x, lo = op.r2.z, hi = op.r2.w;
      float u[4] = {Jd[0] * ak, Jd[1] * ak, Jd[2] * ak, Jd[3] * ak}, dphi[4] = {0, 0, 0, 0};
      asm volatile("" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]));
      float imp;
#define MJH_ROWSUM(N) do { MJH_DPP_BF4(u, N, 0xB1); MJH_DPP_BF4(u, N, 0x4E); MJH_DPP_BF4(u, N, 0x141); MJH_DPP_BF4(u, N, 0x140); } while (0)
      if (kind == BK_PYR4) { MJH_ROWSUM(4); for (int j = 0; j < 4; j++) u[j] -= aref[j]; imp = solve_rows<4, 6, EXTRA>(ns, R, lo, hi, u, f, Q, X, dphi); }
      else if (kind == BK_PYR3) { MJH_ROWSUM(3); for (int j = 0; j < 3; j++) u[j] -= aref[j]; imp = solve_rows<3, 4, EXTRA>(ns, R, lo, hi, u, f, Q, X, dphi); }
      else { MJH_ROWSUM(1); u[0] -= aref[0]; imp = solve_rows<1, 1, EXTRA>(ns, R, lo, hi, u, f, Q, X, dphi); }
#undef MJH_ROWSUM
      ak += (Bp[0] * dphi[0] + Bp[1] * dphi[1] + Bp[2] * dphi[2] + Bp[3] * dphi[3]) * bs;
      if (on) c.qacc[d] = ak;                                                 // scatter (the group's blocks touch disjoint dofs)
      impl += imp_fixed(op.act * imp, iq.qs, iq.cl);
      if constexpr (BUF) {
        const unsigned of = (l == 0 && op.act > 0.0f) ? (unsigned)op.b * 64u + (unsigned)(BF_F * 4) : MJH_BUF_OOB;
        mjh_v4u f4; mjh_v2u f2;
        __builtin_memcpy(&f4, f, 16); __builtin_memcpy(&f2, f + 4, 8);
        __builtin_amdgcn_raw_buffer_store_b128(f4, c.rsrc, (int)of, c.oblkf, 0);
        __builtin_amdgcn_raw_buffer_store_b64(f2, c.rsrc, (int)(of + 16u), c.oblkf, 0);
      } else
      if (l == 0 && op.act > 0.0f) {
        float* bf = c.blkf + op.b * BLKF_STRIDE + BF_F;
        *(float4*)(bf) = make_float4(f[0], f[1], f[2], f[3]);
        *(float2*)(bf + 4) = make_float2(f[4], f[5]);
      }
    };
    // operands are requested two steps ahead (three register sets in rotation): the pools of a cohort exceed the L2 and one
    // step is shorter than a miss.  A wave's items of one sweep: for every group g its chunks c4 = wid, wid + nwave, ... (at
    // least one, possibly idle, so that every wave meets the barrier that ends the group).  Two iterators over that cyclic
    // sequence: gF / cF fetches, gP / cP processes.
    const int nw = c.nwave;
    int gF = 0, cF = c.wid, gP = 0, cP = c.wid;
    auto nextFetch = [&]() __attribute__((always_inline)) {
      QOp o = fetchQ(gF, cF);
      cF += nw;
      if (c.gstart[gF] + 4 * cF >= c.gstart[gF + 1]) { cF = c.wid; gF = gF + 1 == c.ngrp ? 0 : gF + 1; }
      return o;
    };
    int impl = 0; bool done = false;
    auto stepDone = [&]() __attribute__((always_inline)) {          // end of a step; at the end of a sweep: convergence test
      cP += nw;
      if (c.gstart[gP] + 4 * cP < c.gstart[gP + 1]) return;         // more chunks of this group for this wave
      cP = c.wid;
      if (multi) __syncthreads();                                     // the group's scatters are visible to every wave
      if (++gP < c.ngrp) return;
      gP = 0; niter++;
      int improvement = __builtin_amdgcn_readlane(impl, 0) + __builtin_amdgcn_readlane(impl, 16) + __builtin_amdgcn_readlane(impl, 32) + __builtin_amdgcn_readlane(impl, 48);
      if (multi) {
        if (lane == 0) ((int*)c.red)[c.wid] = improvement;
        __syncthreads();
        improvement = 0;
        for (int w = 0; w < nw; w++) improvement += ((const int*)c.red)[w];
        __syncthreads();
      }
      impl = 0;
      done = improvement < iq.thr || niter >= itmax;
    };
    if (MJH_QUAD_DEPTH >= 3 && c.ngrp >= 3) {
      QOp o0 = nextFetch(), o1 = nextFetch(), o2;
      while (true) {
        o2 = nextFetch(); processQ(o0, impl); stepDone(); if (done) break;
        o0 = nextFetch(); processQ(o1, impl); stepDone(); if (done) break;
        o1 = nextFetch(); processQ(o2, impl); stepDone(); if (done) break;
      }
    } else {
      QOp o0 = nextFetch(), o1;
      while (true) {
        o1 = nextFetch(); processQ(o0, impl); stepDone(); if (done) break;
        o0 = nextFetch(); processQ(o1, impl); stepDone(); if (done) break;
      }
    }
  } else
  if (c.nblk < 3) {
    // (the prefetch would read the forces of a block before its pending update is stored)
    for (int it = 0; it < itmax; it++) {
      float improvement = 0;
      for (int k = 0; k < c.nblk; k++) { MOp op = fetch8(blockAt(k)); process8(op, improvement); if (!solo) __syncthreads(); }
      niter = it + 1;
      if (improvement * c.scale < tol) break;
    }
  } else {
    MOp opA = fetch8(blockAt(0)), opB;
    for (int it = 0; it < itmax; it++) {
      float improvement = 0;
      for (int k = 0; k < c.nblk; k += 2) {
        opB = fetch8(blockAt(k + 1 < c.nblk ? k + 1 : 0));
        process8(opA, improvement);
        if (k + 1 < c.nblk) {
          opA = fetch8(blockAt(k + 2 < c.nblk ? k + 2 : 0));
          process8(opB, improvement);
        } else opA = opB;                  // odd block count: block 0 of the next sweep was loaded into B
      }
      niter = it + 1;
      if (improvement * c.scale < tol) break;
    }
  }
  }   // sweep mode
  return niter + nmain;
}

// EXTRA: the model has a pair for the generic convex narrow phase (cylinder-x, capsule-box, ellipsoid-x, mesh geoms) or
// asks for noslip sweeps; a separate instantiation so that models without either (S24, box piles, the robots'
// primitive geometry) keep their code size and register allocation
#include "patch_pgs.h"
#include "window_pgs.h"

// WPRE: 0 the whole kernel; assemble-only instances of the window chain (no sweep code): 1 with the base-row pool in LDS, 2 with the pool in the
// env's window slice (models beyond 64 contacts: the pool is what decides how many envs a CU holds)
template <int NROW, bool DIAGM, bool EXTRA, int WPRE = 0>
#ifndef MJH_STEP_WAVES
// resident waves per SIMD the register allocation aims at: two for the instances that keep sweep records in registers (free-body
// patch sweep) or long dense stages (many-body chain), three for the small articulated models (C3, C5: latency-bound, every extra
// resident wave counts) (A/B builds: -DMJH_STEP_WAVES=3 with -DPP_NRC=0)
#define MJH_STEP_WAVES ((!DIAGM && NROW <= 4) ? 3 : 2)
#endif
// the assemble-only instances of the window chain: waves per SIMD their register allocation aims at.  6 = 80 registers: such a wave fits on a
// SIMD BESIDE a window wavefront (422 of the 512 registers), i.e. on every SIMD of the chip instead of only on those the other cohorts'
// window waves have left
#ifndef MJH_WPRE_WAVES
#define MJH_WPRE_WAVES 2
#endif
#define MJH_STEP_WAVES_T (WPRE ? MJH_WPRE_WAVES : MJH_STEP_WAVES)
__global__ __launch_bounds__(64, MJH_STEP_WAVES_T) void mjh_step_kernel(const DConst* __restrict__ C, const DState S_arg, int env0, int nsteps, int ph, int xflags) {
  // the device-state descriptor (32 scalars: the pointers of the per-env arrays): by value in the kernel-argument segment.  As a plain argument all of it is
  // loaded at the kernel's entry, and what the LAST stage stores through is spilled across the whole kernel; the assemble-only instances of the window chain
  // read it through the segment pointer where a field is used (same segment, same values)
#ifndef MJH_LAZY_STATE
#define MJH_LAZY_STATE true
#endif
  static_assert(alignof(DState) <= alignof(const DConst*), "the descriptor follows the first argument in the kernel-argument segment without padding");
  const DState& S = MJH_LAZY_STATE ? *(const DState*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(const DConst*)) : S_arg;
  // model descriptor + LDS layout live in device memory (uploaded once): uniform scalar loads on demand instead
  // of a by-value kernarg struct that the lambdas below would force into a private (scratch) copy
  const DModel& M = C->M;
  const Lay& L = C->L;
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  // longest-job-first dispatch: workgroups are issued in blockIdx order, so the envs that needed the most solver work
  // in the previous step go first (S.env_order, built by mjh_order_kernel); results do not depend on the order
  const int env = S.env_order ? S.env_order[env0 + blockIdx.x] : env0 + (int)blockIdx.x;
  const int xrow = env - env0;   // row of this env in the x_* export buffers of a ranged launch
  const int nq = M.nq, nv = M.nv, nbody = M.nbody, njnt = M.njnt, ngeom = M.ngeom;

  // model tables: one base pointer per element type + a kernarg-resident offset per table (kept as
  // (base, offset) pairs so that ~70 table addresses do not each pin an SGPR pair for the whole kernel)
#ifndef MJH_LAZY_TABLES
#define MJH_LAZY_TABLES (WPRE != 0 || (!DIAGM && NROW <= 4 && EXTRA))
#endif
#define IT(n) const XTab<MJH_LAZY_TABLES, int, &DModel::o_##n> n(M);
  MJH_INT_TABLES(IT)
#undef IT
#define FT(n) const XTab<MJH_LAZY_TABLES, float, &DModel::o_##n> n(M);
  MJH_FLT_TABLES(FT)
#undef FT
  float* const gs = NROW == 8 ? S.gscratch + (size_t)env * (size_t)S.gstride : nullptr;   // many-body models: big pools in global memory
  // (compile-time choice per instantiation: the address space of every array is known to the compiler)
#ifndef MJH_LAZY_LDS
#define MJH_LAZY_LDS (WPRE != 0)
#endif
#define LA(n) const XArr<MJH_LAZY_LDS, &Lay::n> s_##n(lds, L);
  MJH_LDS_SMALL(LA)
#undef LA
#define LA(n) float* s_##n = NROW == 8 ? gs + (-1 - L.n) : lds + L.n;
  MJH_LDS_POOLS(LA)
#undef LA
  int* s_blki_i = (int*)s_blki; int* s_sched_i = (int*)s_sched; int* s_order_i = (int*)s_order; 
  // chain-walk tables: LDS copies for articulated models; free-body models (DIAGM) walk 6-dof chains a few times per step
  // and read the shared tables instead (their LDS space is not allocated)
  const int* s_dofpar_i = DIAGM ? (dof_parentid + 0) : (const int*)s_dofpar; const int* s_dofMadr_i = DIAGM ? (dof_Madr + 0) : (const int*)s_dofMadr;
  int* s_anc_i = (int*)s_anc;
  // articulated models with at most 128 dofs: solves with the factor level by level (all trees at once), short trees factored four at a time
  const bool level_solves = !DIAGM && nv <= 128;
  // ... up to four trees (one group of factor_trees_short: C3's four arms); more of them are better off one per lane
  const bool short_rows = level_solves && M.ntree <= 4;
  // M after the factorisation: articulated models in the many-body layout keep only the factor in LDS (it is built in M's place)
  const float* qM_ro = (NROW == 8 && !DIAGM) ? gs + L.g_qM : s_qM;
  // assemble-only launches of the window chain of models beyond 64 contacts: the base-row pool (and the raw-contact staging that aliases it) — the largest array by far,
  // 192 B per contact — lives in the env's slice of the window buffer (global memory, L2-resident) instead of LDS: the launch is bound by
  // the latency of one wave per env, and its LDS sets how many envs a CU holds (S24D at 96 contacts: 44 KB -> 25 KB, 3 -> 6 per CU)
  if constexpr (WPRE == 2) { s_J = S.wbuf + (size_t)env * (size_t)S.wstride + S.wj_off; s_B = s_J; }
  // ... and the per-base scratch vectors bv / phi (velocity stage: J qvel; mj_inverse: J qacc and the base forces) in the contact records, which
  // nothing reads once the rows are made (the last reader is the row-parameter stage): this instance's LDS ends in front of their own slots (engine.hip)
  if constexpr (WPRE == 2) if (8 * max(M.maxblk, 1) <= M.maxcon * CON_STRIDE) { s_bv = s_con; s_phi = s_con + 4 * max(M.maxblk, 1); }
  float* s_stage = s_J;  // raw-contact staging aliases the (not yet built) base-row storage
  const int rowW = M.rowW;

  const size_t qrow = (size_t)env * M.nqp, vrow = (size_t)env * M.nvp;
  const float h = M.timestep;
  const float grav[3] = {(M.disableflags & MJH_DSBL_GRAVITY) ? 0.0f : M.gravity[0], (M.disableflags & MJH_DSBL_GRAVITY) ? 0.0f : M.gravity[1],
                         (M.disableflags & MJH_DSBL_GRAVITY) ? 0.0f : M.gravity[2]};

  // ------------------------------------------------------------------ load state
  if (ph & PH_RESET) {
    for (int i = lane; i < nq; i += 64) S.qpos[qrow + i] = S.initial_qpos[qrow + i];
    for (int i = lane; i < nv; i += 64) { S.qvel[vrow+i] = 0; S.qacc_ws[vrow+i] = 0; S.qvel_ref[vrow+i] = 0; S.qfrc_applied[vrow+i] = 0; S.ddq[vrow+i] = 0; S.dq[vrow+i] = 0; }
    if (lane == 0) { S.time[env] = 0; S.stats[4*env] = 0; S.stats[4*env+1] = 0; S.stats[4*env+2] = 0; S.stats[4*env+3] = 0; }
    return;
  }
  for (int i = lane; i < nq; i += 64) s_qpos[i] = S.qpos[qrow + i];
  for (int i = lane; i < nv; i += 64) {
    s_qvel[i] = S.qvel[vrow + i]; s_ws[i] = S.qacc_ws[vrow + i];
    // last step's qacc is an input of mj_inverse only; a launch that will not overwrite it (no step2) carries it through
    s_qacc[i] = ((ph & PH_INV) || !(ph & PH_STEP2)) ? S.qacc[vrow + i] : 0.0f;
    // qvel_ref / qfrc_applied only cross launches in the split API (step1 | inverse | step2 as separate calls)
    if (!(ph & PH_STEP1)) { s_qvref[i] = S.qvel_ref[vrow + i]; s_applied[i] = S.qfrc_applied[vrow + i]; }
  }
  if (lane < (M.patch ? PP_ZERO : 4)) s_zero[lane] = 0;   // what lanes outside a block read instead of its Jacobian (patch sweep: instead of a row record)
  // hot chain-walk tables and the (possibly per-env) model parameters go to LDS once per launch
  if (!DIAGM) {
    for (int i = lane; i < nv; i += 64) { ((int*)s_dofpar)[i] = dof_parentid[i]; ((int*)s_dofMadr)[i] = dof_Madr[i]; }
    WSYNC();
    for (int i = lane; i < nv; i += 64) {       // ancestor lists (wave-cooperative factor / solve): chain walks on the LDS copies
      int a = s_dofMadr_i[i]; s_anc_i[a] = i | (a << 16);
      for (int j = s_dofpar_i[i]; j >= 0; j = s_dofpar_i[j]) s_anc_i[++a] = j | (s_dofMadr_i[j] << 16);
    }
  }
  for (int i = lane; i < 3 * ngeom; i += 64) s_p_gsize[i] = S.p_geom_size ? S.p_geom_size[(size_t)env * S.p_stride + i] : geom_size[i];
  for (int i = lane; i < ngeom; i += 64) s_p_rbound[i] = S.p_geom_rbound ? S.p_geom_rbound[(size_t)env * S.p_stride + i] : geom_rbound[i];
  for (int i = lane; i < nbody; i += 64) s_p_mass[i] = S.p_body_mass ? S.p_body_mass[(size_t)env * S.p_stride + i] : body_mass[i];
  for (int i = lane; i < 3 * nbody; i += 64) s_p_inertia[i] = S.p_body_inertia ? S.p_body_inertia[(size_t)env * S.p_stride + i] : body_inertia[i];
  // many-body layout, three-launch step (engine.hip): PH_PRE stops in front of the solver sweeps and hands over through the
  // env's scratch slice; PH_POST skips everything between the factorisation and the end of the sweeps
  const bool pre = NROW == 8 && (ph & PH_PRE), post = NROW == 8 && (ph & PH_POST);
  // window sweep (window_pgs.h): assemble launch of a patch-eligible free-body model; an env WITHOUT constraint rows (or with more than
  // the window kernel takes) finishes the step in this launch and leaves 0 in its hand-over header
  bool wpre = false;
  if constexpr (DIAGM && NROW <= 2) wpre = M.window != 0 && (ph & PH_PRE) && S.wbuf != nullptr;
  if (wpre && lane == 0) { int* wh0 = (int*)(S.wbuf + (size_t)env * (size_t)S.wstride); wh0[0] = 0; wh0[4] = 0; wh0[5] = 0; }
  if (post) for (int i = lane; i < nv; i += 64) s_qvel[i] = gs[L.g_qvel + i];     // (the controller may have overridden velocities)
  double time = S.time[env];
  // spawn/destroy as slot toggling: bit b set = body b is an INACTIVE slot in this env
  const unsigned slotmask = S.slot_mask ? (unsigned)__builtin_amdgcn_readfirstlane((int)S.slot_mask[env]) : 0u;
  const int sbase = nbody > 32 ? nbody - 32 : 0;   // mask bit i = body sbase + i: the LAST 32 bodies of a big model are the toggleable slots
  int flags = 0, ncon = 0, nefc = 0, niter = 0, cost_hint = 0;
  PROF(0);
  if ((xflags & XF_PROF) && lane == 0) {   // 100 MHz wall clock (comparable across CUs) and where this env ran: HW_ID | XCC_ID << 32
    S.x_prof[(size_t)blockIdx.x * PROF_STRIDE + 16] = (long long)__builtin_amdgcn_s_memrealtime();
    S.x_prof[(size_t)blockIdx.x * PROF_STRIDE + 19] = env;
    S.x_prof[(size_t)blockIdx.x * PROF_STRIDE + 18] = (long long)__builtin_amdgcn_s_getreg(63492) | ((long long)__builtin_amdgcn_s_getreg(63508) << 32);
  }
  WSYNC();
  PROF(1);

  // In-kernel step loop (SURVEY §7.3 "one launch per n steps"): the env's state stays in LDS between the steps of a launch — qpos, qvel,
  // the warm start and last step's qacc are exactly what the end of a step leaves there, the per-launch tables above are in spans no
  // stage aliases (engine.hip: LDS layout; the patch pool and the many-body / window chains alias or leave the kernel: nsteps = 1 there,
  // the host's rule) — commands are consumed by the first step, the in-engine PD law is evaluated every step.  Bitwise equal to nsteps
  // launches of one step (tests/test_gpu_round4.py).
  // Register discipline: what the body derives from the lane index (an LDS address per array, the lane predicates of every stage —
  // cheap arithmetic) is loop-invariant, and hoisted in front of the loop it stays live around the whole body: 256 VGPRs + 244 B of
  // scratch per lane when tried (C5 -21 %).  So every iteration takes the lane index through an opaque move: nothing derived from it
  // can leave the loop, live ranges are those of a one-step launch (168 VGPRs, three waves per SIMD).  Instances whose launches always
  // run one step (free-body patch / window models, the many-body chain) compile the loop away.
  constexpr bool LOOP = !DIAGM && NROW <= 4;
  int step = 0;
step_again:      // (a backward goto instead of a `for`: the instances without the loop keep the straight-line body they were tuned with)
  {
    int lane_it = (int)threadIdx.x;
    if constexpr (LOOP) asm volatile("" : "+v"(lane_it));
    const int lane = lane_it;
    cost_hint = 0;
    // ---- bad-state check (mj_checkPos / mj_checkVel): reset this env
    {
      bool badv = false;
      for (int i = lane; i < nq; i += 64) { float x = s_qpos[i]; badv |= !(x == x) || fabsf(x) > MJ_MAXVAL; }
      for (int i = lane; i < nv; i += 64) { float x = s_qvel[i], y = s_qacc[i]; badv |= !(x == x) || fabsf(x) > MJ_MAXVAL || !(y == y) || fabsf(y) > MJ_MAXVAL; }
      if (wave_any(badv)) {
        for (int i = lane; i < nq; i += 64) s_qpos[i] = S.initial_qpos[qrow + i];
        for (int i = lane; i < nv; i += 64) { s_qvel[i] = 0; s_qacc[i] = 0; s_ws[i] = 0; s_applied[i] = 0; }
        flags |= 4;
        WSYNC();
      }
    }
    // ================================================================ position stage
    if (post) {
      // three-launch step: the integrate launch only needs what mj_kinematics does to qpos (quaternion normalisation) and
      // the mass matrix for the implicit-damping solve, handed over by the assemble launch
      for (int j = lane; j < njnt; j += 64) {
        const int jt = jnt_type[j], qa = jnt_qposadr[j] + (jt == MJH_JNT_FREE ? 3 : 0);
        if (jt == MJH_JNT_FREE || jt == MJH_JNT_BALL) {
          float q[4] = {s_qpos[qa], s_qpos[qa+1], s_qpos[qa+2], s_qpos[qa+3]};
          normalize4(q);
          s_qpos[qa] = q[0]; s_qpos[qa+1] = q[1]; s_qpos[qa+2] = q[2]; s_qpos[qa+3] = q[3];
        }
      }
      for (int i = lane; i < M.nM; i += 64) s_qM[i] = gs[L.g_qM + i];
      WSYNC();
    } else {
    // ---- FK (mj_kinematics)
    if (lane == 0) {
      s_xpos[0] = s_xpos[1] = s_xpos[2] = 0; s_xquat[0] = 1; s_xquat[1] = s_xquat[2] = s_xquat[3] = 0;
      for (int k = 0; k < 9; k++) { s_xmat[k] = (k % 4 == 0) ? 1.0f : 0.0f; s_ximat[k] = s_xmat[k]; }
      s_xipos[0] = s_xipos[1] = s_xipos[2] = 0;
    }
    WSYNC();
    for (int lev = 1; lev <= M.maxlevel; lev++) {
      for (int b = lane; b < nbody; b += 64) {
        if (body_level[b] != lev) continue;
        float xpos[3], xquat[4], mat[9];
        const int jn = body_jntnum[b], ja = body_jntadr[b];
        if (jn == 1 && jnt_type[ja] == MJH_JNT_FREE) {
          const int qa = jnt_qposadr[ja];
          xpos[0] = s_qpos[qa]; xpos[1] = s_qpos[qa+1]; xpos[2] = s_qpos[qa+2];
          xquat[0] = s_qpos[qa+3]; xquat[1] = s_qpos[qa+4]; xquat[2] = s_qpos[qa+5]; xquat[3] = s_qpos[qa+6];
          normalize4(xquat);
          s_qpos[qa+3] = xquat[0]; s_qpos[qa+4] = xquat[1]; s_qpos[qa+5] = xquat[2]; s_qpos[qa+6] = xquat[3];
          s_xanchor[3*ja] = xpos[0]; s_xanchor[3*ja+1] = xpos[1]; s_xanchor[3*ja+2] = xpos[2];
          s_xaxis[3*ja] = 0; s_xaxis[3*ja+1] = 0; s_xaxis[3*ja+2] = 1;
        } else if (EXTRA && M.nmocap > 0 && M.I[M.o_body_mocapid + b] >= 0) {
          // mocap body (mj_kinematics: xpos / xquat = d->mocap_pos / mocap_quat; the *_ref bodies of mj_sim.cpp:903)
          const float* mp = S.mocap + ((size_t)env * M.nmocap + M.I[M.o_body_mocapid + b]) * 7;
          xpos[0] = mp[0]; xpos[1] = mp[1]; xpos[2] = mp[2]; xquat[0] = mp[3]; xquat[1] = mp[4]; xquat[2] = mp[5]; xquat[3] = mp[6];
        } else {
          const int p = body_parentid[b];
          float t[3]; rotvec(t, s_xmat + 9*p, body_pos + 3*b);
          xpos[0] = s_xpos[3*p] + t[0]; xpos[1] = s_xpos[3*p+1] + t[1]; xpos[2] = s_xpos[3*p+2] + t[2];
          mulquat(xquat, s_xquat + 4*p, body_quat + 4*b);
          for (int j = ja; j < ja + jn; j++) {
            const int qa = jnt_qposadr[j], jt = jnt_type[j];
            float vec[3], anchor[3], axis[3];
            quat2mat(mat, xquat);
            rotvec(vec, mat, jnt_pos + 3*j);
            anchor[0] = xpos[0] + vec[0]; anchor[1] = xpos[1] + vec[1]; anchor[2] = xpos[2] + vec[2];
            rotvec(axis, mat, jnt_axis + 3*j);
            if (jt == MJH_JNT_SLIDE) {
              float dq = s_qpos[qa] - qpos0[qa];
              xpos[0] += axis[0]*dq; xpos[1] += axis[1]*dq; xpos[2] += axis[2]*dq;
            } else if (jt == MJH_JNT_BALL || jt == MJH_JNT_HINGE) {
              float ql[4], r[4];
              if (jt == MJH_JNT_BALL) {
                ql[0] = s_qpos[qa]; ql[1] = s_qpos[qa+1]; ql[2] = s_qpos[qa+2]; ql[3] = s_qpos[qa+3];
                normalize4(ql);
                s_qpos[qa] = ql[0]; s_qpos[qa+1] = ql[1]; s_qpos[qa+2] = ql[2]; s_qpos[qa+3] = ql[3];
              } else axisangle2quat(ql, jnt_axis + 3*j, s_qpos[qa] - qpos0[qa]);
              mulquat(r, xquat, ql);
              xquat[0] = r[0]; xquat[1] = r[1]; xquat[2] = r[2]; xquat[3] = r[3];
              quat2mat(mat, xquat); rotvec(vec, mat, jnt_pos + 3*j);
              xpos[0] = anchor[0] - vec[0]; xpos[1] = anchor[1] - vec[1]; xpos[2] = anchor[2] - vec[2];
            }
            s_xanchor[3*j] = anchor[0]; s_xanchor[3*j+1] = anchor[1]; s_xanchor[3*j+2] = anchor[2];
            s_xaxis[3*j] = axis[0]; s_xaxis[3*j+1] = axis[1]; s_xaxis[3*j+2] = axis[2];
          }
        }
        normalize4(xquat);
        quat2mat(mat, xquat);
        float t[3], qi[4], imat[9];
        rotvec(t, mat, body_ipos + 3*b);
        mulquat(qi, xquat, body_iquat + 4*b); quat2mat(imat, qi);
#pragma unroll
        for (int k = 0; k < 3; k++) { s_xpos[3*b+k] = xpos[k]; s_xipos[3*b+k] = xpos[k] + t[k]; }
#pragma unroll
        for (int k = 0; k < 4; k++) s_xquat[4*b+k] = xquat[k];
#pragma unroll
        for (int k = 0; k < 9; k++) { s_xmat[9*b+k] = mat[k]; s_ximat[9*b+k] = imat[k]; }
      }
      WSYNC();
    }
    for (int g = lane; g < ngeom; g += 64) {
      const int b = geom_bodyid[g];
      float t[3], q[4], mat[9];
      rotvec(t, s_xmat + 9*b, geom_pos + 3*g);
      mulquat(q, s_xquat + 4*b, geom_quat + 4*g); quat2mat(mat, q);
#pragma unroll
      for (int k = 0; k < 3; k++) s_gpos[3*g+k] = s_xpos[3*b+k] + t[k];
#pragma unroll
      for (int k = 0; k < 9; k++) s_gmat[9*g+k] = mat[k];
    }
    if (EXTRA && M.nsite > 0) {          // site frames in the world (read by the force / torque sensors after the solve)
      for (int si = lane; si < M.nsite; si += 64) {
        const int b = M.I[M.o_site_bodyid + si];
        float t[3], q[4], mat[9];
        rotvec(t, s_xmat + 9*b, M.F + M.o_site_pos + 3*si);
        mulquat(q, s_xquat + 4*b, M.F + M.o_site_quat + 4*si); quat2mat(mat, q);
#pragma unroll
        for (int k = 0; k < 3; k++) s_site[12*si+k] = s_xpos[3*b+k] + t[k];
#pragma unroll
        for (int k = 0; k < 9; k++) s_site[12*si+3+k] = mat[k];
      }
    }
    if (xflags & (XF_BODY | XF_GEOM)) {
      WSYNC();
      const size_t e = xrow;
      if (S.x_xpos) for (int i = lane; i < 3*nbody; i += 64) S.x_xpos[e*3*nbody + i] = s_xpos[i];
      if (S.x_xquat) for (int i = lane; i < 4*nbody; i += 64) S.x_xquat[e*4*nbody + i] = s_xquat[i];
      if (S.x_gpos) for (int i = lane; i < 3*ngeom; i += 64) S.x_gpos[e*3*ngeom + i] = s_gpos[i];
      if (S.x_gmat) for (int i = lane; i < 9*ngeom; i += 64) S.x_gmat[e*9*ngeom + i] = s_gmat[i];
    }
    if (ph & PH_FKONLY) return;
    PROF(2);

    // ---- subtree COM of every tree root, COM-based inertias, motion axes (mj_comPos)
    for (int r = lane; r < nbody; r += 64) {
      if (r == 0 || body_parentid[r] != 0) continue;
      float sm = 0, c[3] = {0, 0, 0};
      for (int b = r; b < r + body_subtreesize[r]; b++) {
        float m = s_p_mass[b]; sm += m;
        c[0] += m * s_xipos[3*b]; c[1] += m * s_xipos[3*b+1]; c[2] += m * s_xipos[3*b+2];
      }
      if (sm < MJ_MINVAL) { c[0] = s_xipos[3*r]; c[1] = s_xipos[3*r+1]; c[2] = s_xipos[3*r+2]; }
      else { float inv = 1.0f / sm; c[0] *= inv; c[1] *= inv; c[2] *= inv; }
      s_com[3*r] = c[0]; s_com[3*r+1] = c[1]; s_com[3*r+2] = c[2];
    }
    WSYNC();
    if constexpr (DIAGM) {
      // every tree a free body about its own centre with body-aligned principal axes: M = diag(m, m, m, Ix, Iy, Iz) per body, and
      // nothing below needs the spatial inertias or motion axes (contact rows, bias forces, energy, gravity compensation and
      // Cartesian forces have closed forms for these models): cinert / cdof / crb are neither formed nor allocated
      for (int i = lane; i < M.nM; i += 64) { s_qM[i] = 0; }
      WSYNC();
      for (int d = lane; d < nv; d += 64) {
        const int b = dof_bodyid[d], k = d - body_dofadr[b];
        s_qM[s_dofMadr_i[d]] = (k < 3 ? s_p_mass[b] : s_p_inertia[3*b + k - 3]) + dof_armature[d];
      }
      WSYNC();
    } else {
    for (int b = lane; b < nbody; b += 64) {
      float ci[10];
      if (b == 0) { for (int k = 0; k < 10; k++) ci[k] = 0; }
      else {
        const float* com = s_com + 3*body_rootid[b];
        float off[3] = {s_xipos[3*b] - com[0], s_xipos[3*b+1] - com[1], s_xipos[3*b+2] - com[2]};
        inert_com(ci, s_p_inertia + 3*b, s_ximat + 9*b, off, s_p_mass[b]);
      }
#pragma unroll
      for (int k = 0; k < 10; k++) s_cinert[10*b+k] = ci[k];
    }
    for (int d = lane; d < nv; d += 64) {
      const int j = dof_jntid[d], b = dof_bodyid[d], jt = jnt_type[j];
      int k = d - jnt_dofadr[j];
      const float* com = s_com + 3*body_rootid[b];
      float off[3] = {com[0] - s_xanchor[3*j], com[1] - s_xanchor[3*j+1], com[2] - s_xanchor[3*j+2]};
      float cd[6];
      if (jt == MJH_JNT_FREE && k < 3) { cd[0] = cd[1] = cd[2] = 0; cd[3] = (k == 0); cd[4] = (k == 1); cd[5] = (k == 2); }
      else if (jt == MJH_JNT_FREE || jt == MJH_JNT_BALL) {
        if (jt == MJH_JNT_FREE) k -= 3;
        float ax[3] = {s_xmat[9*b + k], s_xmat[9*b + 3 + k], s_xmat[9*b + 6 + k]};
        cd[0] = ax[0]; cd[1] = ax[1]; cd[2] = ax[2]; cross3(cd + 3, ax, off);
      } else if (jt == MJH_JNT_SLIDE) { cd[0] = cd[1] = cd[2] = 0; cd[3] = s_xaxis[3*j]; cd[4] = s_xaxis[3*j+1]; cd[5] = s_xaxis[3*j+2]; }
      else { float ax[3] = {s_xaxis[3*j], s_xaxis[3*j+1], s_xaxis[3*j+2]}; cd[0] = ax[0]; cd[1] = ax[1]; cd[2] = ax[2]; cross3(cd + 3, ax, off); }
#pragma unroll
      for (int q = 0; q < 6; q++) s_cdof[6*d+q] = cd[q];
    }
    WSYNC();
    // ---- CRBA (mj_crb): composite inertia = sum over the body's (contiguous) subtree
    for (int b = lane; b < nbody; b += 64) {
      float acc[10];
#pragma unroll
      for (int k = 0; k < 10; k++) acc[k] = 0;
      if (b > 0) for (int c = b; c < b + body_subtreesize[b]; c++)
#pragma unroll
        for (int k = 0; k < 10; k++) acc[k] += s_cinert[10*c+k];
#pragma unroll
      for (int k = 0; k < 10; k++) s_crb[10*b+k] = acc[k];
    }
    WSYNC();
    for (int i = lane; i < nv; i += 64) {
      float buf[6], cd[6];
#pragma unroll
      for (int q = 0; q < 6; q++) cd[q] = s_cdof[6*i+q];
      mul_inert_vec(buf, s_crb + 10*dof_bodyid[i], cd);
      int adr = s_dofMadr_i[i];
      for (int j = i; j >= 0; j = s_dofpar_i[j]) {
        float v = 0;
#pragma unroll
        for (int q = 0; q < 6; q++) v += s_cdof[6*j+q] * buf[q];
        if (j == i) v += dof_armature[i];
        s_qM[adr] = v; s_qLD[adr] = v;
        if (NROW == 8) gs[L.g_qM + adr] = v;      // many-body layout: the factor takes M's place in LDS (L.qLD == L.qM), M itself lives in the env's scratch slice
        adr++;
      }
    }
    WSYNC();
    }   // !DIAGM
    if (ph & PH_MULM) {  // mj_mulM (mj_sim.cpp:1057) for the host API
      for (int i = lane; i < nv; i += 64) { s_tmpv[i] = S.x_vec[(size_t)blockIdx.x * M.nvp + i]; s_tmpv2[i] = 0; }
      WSYNC();
      for (int i = lane; i < nv; i += 64) {
        int adr = s_dofMadr_i[i]; float vi = s_tmpv[i], acc = s_qM[adr] * vi; int k = 1;
        for (int j = s_dofpar_i[i]; j >= 0; j = s_dofpar_i[j]) { float mij = s_qM[adr + k]; acc += mij * s_tmpv[j]; atomicAdd(&s_tmpv2[j], mij * vi); k++; }
        atomicAdd(&s_tmpv2[i], acc);
      }
      WSYNC();
      for (int i = lane; i < nv; i += 64) S.x_res[(size_t)blockIdx.x * M.nvp + i] = s_tmpv2[i];
      return;
    }
    PROF(3);
    // ---- L'DL factorisation (mj_factorM): one lane per kinematic tree
    if (DIAGM) { for (int d = lane; d < nv; d += 64) s_qLDinv[d] = 1.0f / s_qM[s_dofMadr_i[d]]; }   // every tree: M is diagonal (single free body about its COM)
    else {
      for (int t = 0; t < M.ntree; t++) if (tree_dofnum[t] >= MJH_WAVE_TREE_MIN) factor_tree_wave(s_qLD, s_qLDinv, s_anc_i, s_dofMadr_i, tree_dofadr[t], tree_dofnum[t], M.nM, nv, lane);
      if (short_rows) factor_trees_short(s_qLD, s_qLDinv, s_anc_i, s_dofMadr_i, tree_dofadr, tree_dofnum, M.ntree, M.nM, nv, lane);
      else for (int t = lane; t < M.ntree; t += 64) if (tree_dofnum[t] < MJH_WAVE_TREE_MIN) factor_tree(s_qLD, s_qLDinv, s_dofpar_i, s_dofMadr_i, tree_dofadr[t], tree_dofnum[t]);
    }
    WSYNC();
    }   // !post

    PROF(4);
    // ---- collision (mj_collision): lanes = candidate geom pairs of the static pair list
    ncon = 0;
    if (!post && !(M.disableflags & (MJH_DSBL_CONTACT | MJH_DSBL_CONSTRAINT))) {
      int conbase = 0;
      // one round: narrow phase of (up to) 64 candidate pairs, contacts appended in lane (= pair) order
      auto collide_round = [&](const int ip) __attribute__((always_inline)) {
        int n = 0, g1 = 0, g2 = 0; float margin = 0, gap = 0;
        float* st = s_stage;
        CvxGeom G1, G2; bool cvx = false;
        G1.type = G2.type = 0; G1.vert = G2.vert = nullptr; G1.nvert = G2.nvert = 0; G1.pad = G2.pad = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) { G1.mat[k] = G2.mat[k] = 0; if (k < 3) { G1.pos[k] = G2.pos[k] = 0; G1.size[k] = G2.size[k] = 0; } }
        if (ip >= 0) {
          g1 = pair_geom1[ip]; g2 = pair_geom2[ip];
          st = s_stage + pair_stageadr[ip] * RAW_STRIDE;
          const int t1 = geom_type[g1], t2 = geom_type[g2];
          margin = fmaxf(geom_margin[g1], geom_margin[g2]); gap = fmaxf(geom_gap[g1], geom_gap[g2]);
          float p1[3], p2[3], m1[9], m2[9], z1[3], z2[3];
#pragma unroll
          for (int k = 0; k < 3; k++) { p1[k] = s_gpos[3*g1+k]; p2[k] = s_gpos[3*g2+k]; z1[k] = s_p_gsize[3*g1+k]; z2[k] = s_p_gsize[3*g2+k]; }
#pragma unroll
          for (int k = 0; k < 9; k++) { m1[k] = s_gmat[9*g1+k]; m2[k] = s_gmat[9*g2+k]; }
          bool cull;
          const int sb1 = geom_bodyid[g1], sb2 = geom_bodyid[g2];
          const unsigned r1 = (unsigned)(sb1 - sbase), r2 = (unsigned)(sb2 - sbase);
          const bool parked = ((r1 < 32u && ((slotmask >> r1) & 1u)) || (r2 < 32u && ((slotmask >> r2) & 1u)));
          float tt[3] = {p2[0]-p1[0], p2[1]-p1[1], p2[2]-p1[2]};
          if (t1 == MJH_GEOM_PLANE) { float nn[3] = {m1[2], m1[5], m1[8]}; cull = dot3(tt, nn) > s_p_rbound[g2] + margin; }
          else { float bound = s_p_rbound[g1] + s_p_rbound[g2] + margin; cull = dot3(tt, tt) > bound * bound; }
          if (!cull && !parked) {
            if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_BOX) n = c_plane_box(p1, m1, p2, m2, z2, margin, st);
            else if (t1 == MJH_GEOM_BOX && t2 == MJH_GEOM_BOX) n = c_box_box(p1, m1, z1, p2, m2, z2, margin, st);
            else if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_SPHERE) n = c_plane_sphere(p1, m1, p2, z2[0], margin, st, 0);
            else if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_CAPSULE) n = c_plane_capsule(p1, m1, p2, m2, z2, margin, st);
            else if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_CYLINDER) n = c_plane_cylinder(p1, m1, p2, m2, z2, margin, st);
            else if (t1 == MJH_GEOM_SPHERE && t2 == MJH_GEOM_SPHERE) n = c_sphere_sphere(p1, z1[0], p2, z2[0], margin, st);
            else if (t1 == MJH_GEOM_SPHERE && t2 == MJH_GEOM_CAPSULE) n = c_sphere_capsule(p1, z1[0], p2, m2, z2, margin, st);
            else if (t1 == MJH_GEOM_CAPSULE && t2 == MJH_GEOM_CAPSULE) n = c_capsule_capsule(p1, m1, z1, p2, m2, z2, margin, st);
            else if (t1 == MJH_GEOM_SPHERE && t2 == MJH_GEOM_BOX) n = c_sphere_box(p1, z1[0], p2, m2, z2, margin, st);
            else if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_ELLIPSOID) n = c_plane_ellipsoid(p1, m1, p2, m2, z2, margin, st);
            else if (EXTRA && t2 == MJH_GEOM_MESH && t1 == MJH_GEOM_PLANE) {
              const Tab<int> geom_dataid{M.I, M.o_geom_dataid}, mesh_vertadr{M.I, M.o_mesh_vertadr}, mesh_vertnum{M.I, M.o_mesh_vertnum};
              const Tab<float> mesh_vert{M.F, M.o_mesh_vert};
              const int id = geom_dataid[g2];
              n = c_plane_mesh(p1, m1, p2, m2, mesh_vert + 3 * mesh_vertadr[id], mesh_vertnum[id], margin, st);
            } else if (EXTRA && pair_is_convex(t1, t2)) {
              const Tab<int> geom_dataid{M.I, M.o_geom_dataid}, mesh_vertadr{M.I, M.o_mesh_vertadr}, mesh_vertnum{M.I, M.o_mesh_vertnum};
              const Tab<float> mesh_vert{M.F, M.o_mesh_vert};
              cvx = true;
              G1.type = t1; G2.type = t2; G1.vert = G2.vert = nullptr; G1.nvert = G2.nvert = 0;
              if (t1 == MJH_GEOM_MESH) { const int id = geom_dataid[g1]; G1.vert = mesh_vert + 3 * mesh_vertadr[id]; G1.nvert = mesh_vertnum[id]; }
              if (t2 == MJH_GEOM_MESH) { const int id = geom_dataid[g2]; G2.vert = mesh_vert + 3 * mesh_vertadr[id]; G2.nvert = mesh_vertnum[id]; }
#pragma unroll
              for (int k = 0; k < 3; k++) { G1.pos[k] = p1[k]; G2.pos[k] = p2[k]; G1.size[k] = z1[k]; G2.size[k] = z2[k]; }
#pragma unroll
              for (int k = 0; k < 9; k++) { G1.mat[k] = m1[k]; G2.mat[k] = m2[k]; }
            }
          }
        }
        // generic convex pairs (dev_convex.h): the lanes run the portal algorithm on their pairs, the wave serves their mesh scans
        if constexpr (EXTRA) { if (wave_any(cvx)) { const int nc = c_convex_wave(G1, G2, margin, st, cvx, lane); if (cvx) n = nc; } }
        const int incl = wave_incl_scan_i(n, lane);
        const int total = wave_last_i(incl);
        const int first = conbase + incl - n;
        for (int q = 0; q < n; q++) {
          const int idx = first + q;
          if (idx >= M.maxcon) break;
          const float* r = st + q * RAW_STRIDE;
          float fr[9]; fr[0] = r[4]; fr[1] = r[5]; fr[2] = r[6];
          make_frame(fr);
          float* c = s_con + idx * CON_STRIDE;
          c[0] = r[0]; c[1] = r[1]; c[2] = r[2]; c[3] = r[3];
#pragma unroll
          for (int k = 0; k < 9; k++) c[4+k] = fr[k];
          c[CON_GEOMS] = __int_as_float(g1 | (g2 << 12) | (max(geom_condim[g1], geom_condim[g2]) << 24)); c[CON_MARGIN] = margin - gap;
        }
        conbase += total;
      };
      // bounding-sphere / plane-distance cull of one pair (the narrow phase repeats it; it is cheap)
      auto survives = [&](const int ip) __attribute__((always_inline)) {     // (branch-free: four of them are evaluated side by side)
        const int g1 = pair_geom1[ip], g2 = pair_geom2[ip];
        const float margin = fmaxf(geom_margin[g1], geom_margin[g2]);
        const int sb1 = geom_bodyid[g1], sb2 = geom_bodyid[g2];
        const unsigned r1 = (unsigned)(sb1 - sbase), r2 = (unsigned)(sb2 - sbase);
        const bool parked = (r1 < 32u && ((slotmask >> (r1 & 31u)) & 1u)) | (r2 < 32u && ((slotmask >> (r2 & 31u)) & 1u));
        const float tt[3] = {s_gpos[3*g2] - s_gpos[3*g1], s_gpos[3*g2+1] - s_gpos[3*g1+1], s_gpos[3*g2+2] - s_gpos[3*g1+2]};
        const float nn[3] = {s_gmat[9*g1+2], s_gmat[9*g1+5], s_gmat[9*g1+8]};
        const float rb2 = s_p_rbound[g2], bound = s_p_rbound[g1] + rb2 + margin;
        const bool far_plane = dot3(tt, nn) > rb2 + margin, far_sphere = dot3(tt, tt) > bound * bound;
        return !parked & !(geom_type[g1] == MJH_GEOM_PLANE ? far_plane : far_sphere);
      };
      // Many candidate pairs (64 free boxes: 2080): the divergent narrow phase would run once per 64 pairs whether or not
      // anything is close; first compact the pairs that pass the cull (pair order is preserved, so is the contact order),
      // then run the narrow phase on full rounds of survivors.  The list lives in dof vectors that are unused until the
      // velocity stage (smooth, asmooth, passive, bias: contiguous).
      // (many-body layout: in the per-block matrix pool of the env's scratch slice instead, dead until the solver starts — it
      //  holds every pair, so the divergent narrow phase runs once over ALL survivors: PR2 + objects, 1760 pairs, 47 survivors
      //  spread over five 384-pair chunks)
      const int poolcap = NROW == 8 ? (max(M.maxblk, 1) * BLKQ_STRIDE) & ~63 : 0;
      const bool pool_list = NROW == 8 && poolcap >= M.npair;
      const int listcap = pool_list ? poolcap : (4 * (((nv + 3) / 4) * 4)) & ~63;
      if (M.npair > 64 && listcap >= 64) {
        int* list = pool_list ? (int*)s_blkq : (int*)s_smooth;
        for (int c0 = 0; c0 < M.npair; c0 += listcap) {
          const int c1 = min(c0 + listcap, M.npair);
          int cnt = 0;
          for (int base = c0; base < c1; base += 256) {       // four rounds of 64 pairs at a time: their table loads overlap
            bool sv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int ip = base + 64 * u + lane; sv[u] = survives(min(ip, c1 - 1)) & (ip < c1); }
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const unsigned long long mk = __ballot(sv[u]);
              if (sv[u]) list[cnt + __popcll(mk & ((1ull << lane) - 1ull))] = base + 64 * u + lane;
              cnt += __popcll(mk);
            }
          }
          WSYNC();
          for (int b2 = 0; b2 < cnt; b2 += 64) collide_round(b2 + lane < cnt ? list[b2 + lane] : -1);
          WSYNC();
        }
      } else {
        for (int base = 0; base < M.npair; base += 64) collide_round(base + lane < M.npair ? base + lane : -1);
      }
      if (conbase > M.maxcon) { flags |= 1; conbase = M.maxcon; }
      ncon = __builtin_amdgcn_readfirstlane(conbase);
    }
    WSYNC();
    if ((xflags & XF_CON) && S.x_contacts) {   // contact snapshot (mjh_get_contacts): records + count, nothing else is written
      float* o = S.x_contacts + (size_t)blockIdx.x * ((size_t)M.maxcon * CON_STRIDE + 4);
      for (int i = lane; i < ncon * CON_STRIDE; i += 64) o[i] = s_con[i];
      if (lane == 0) { o[(size_t)M.maxcon * CON_STRIDE] = __int_as_float(ncon); o[(size_t)M.maxcon * CON_STRIDE + 1] = __int_as_float(flags); }
      if (xflags & XF_NOSTORE) return;
      WSYNC();
    }

    PROF(5);
    // ---- constraint blocks (mj_makeConstraint + mj_makeImpedance).  A block = the rows that share base
    //      Jacobians: 1 row (equality / friction loss / limit / frictionless contact) or the 2(dim-1)
    //      pyramid rows of a contact, which are all  J_n +- mu_k J_k  over nbase = dim base rows.
    nefc = 0;
    int nblk = 0, nbrow = 0, nfixblk = 0;   // nfixblk: number of non-contact blocks (they come first)
    if (!post && !(M.disableflags & MJH_DSBL_CONSTRAINT)) {
      // non-contact blocks come first and own one row of rowW floats each; contact block c owns 4*rowW floats
      auto put_block = [&](int b, int kind, int nrows, int nb, int clamp, int joff4, int id, int rtype, int side) __attribute__((always_inline)) {
        int* hd = s_blki_i + b * BLKI_STRIDE;
        hd[0] = kind | (nrows << 4) | (nb << 8) | (clamp << 12) | (joff4 << 16); hd[1] = id | (rtype << 24) | (side << 28);
      };
      const int w4 = rowW >> 2;
      if (!(M.disableflags & MJH_DSBL_EQUALITY))
        for (int e = 0; e < M.neq; e++) if (eq_active[e]) {
          // joint coupling: one row; connect / weld between bodies (EXTRA instances): 3 / 6 rows, one single-row block each
          const int et = (EXTRA && M.has_weld) ? M.I[M.o_eq_type + e] : MJH_EQ_JOINT;
          const int nr = et == MJH_EQ_WELD ? 6 : (et == MJH_EQ_CONNECT ? 3 : 1);
          if (lane < nr) put_block(nblk + lane, BK_SINGLE, 1, 1, 0, (nblk + lane) * w4, et == MJH_EQ_JOINT ? e : (e | (lane << 16)), et == MJH_EQ_JOINT ? RT_EQ : RT_WELD, 0);
          nblk += nr; nbrow += nr; nefc += nr;
        }
      if (!(M.disableflags & MJH_DSBL_FRICTIONLOSS)) {
        for (int f = lane; f < M.nfl; f += 64) put_block(nblk + f, BK_SINGLE, 1, 1, 2, (nblk + f) * w4, fl_dof[f], RT_FL, 0);
        nblk += M.nfl; nbrow += M.nfl; nefc += M.nfl;
      }
      if (M.has_limits && !(M.disableflags & MJH_DSBL_LIMIT)) {
        for (int base = 0; base < njnt; base += 64) {
          const int j = base + lane; int lo = 0, hi = 0;
          if (j < njnt && jnt_limited[j] && (jnt_type[j] == MJH_JNT_HINGE || jnt_type[j] == MJH_JNT_SLIDE)) {
            float val = s_qpos[jnt_qposadr[j]], mg = jnt_margin[j];
            lo = (val - jnt_range[2*j]) < mg; hi = (jnt_range[2*j+1] - val) < mg;
          }
          const int n = lo + hi, incl = wave_incl_scan_i(n, lane);
          int r = incl - n;
          if (lo) { put_block(nblk + r, BK_SINGLE, 1, 1, 1, (nblk + r) * w4, j, RT_LIMIT, 0); r++; }
          if (hi) put_block(nblk + r, BK_SINGLE, 1, 1, 1, (nblk + r) * w4, j, RT_LIMIT, 1);
          const int tot = wave_last_i(incl);
          nblk += tot; nbrow += tot; nefc += tot;
        }
      }
      // contacts: a contact whose rows do not fit drops it and every later contact (oracle rule)
      const int nfix = nblk; nfixblk = nblk;
      bool stop = false;
      for (int base = 0; base < ncon && !stop; base += 64) {
        const int ic = base + lane; int nr = 0, nb = 0, dim = 0;
        if (ic < ncon) {
          const float* c = s_con + ic * CON_STRIDE;
          dim = CON_DIM(c);
          if (c[0] < c[CON_MARGIN]) { nr = dim == 1 ? 1 : 2 * (dim - 1); nb = dim == 1 ? 1 : (dim == 3 ? 3 : 4); }
        }
        const int inc = nr > 0;
        const int sr = wave_incl_scan_i(nr, lane), sb = wave_incl_scan_i(nb, lane), sc = wave_incl_scan_i(inc, lane);
        const unsigned long long over = __ballot(inc && nefc + sr > M.maxefc);
        const int firstover = over ? __ffsll((long long)over) - 1 : 64;
        if (inc && lane < firstover)
          put_block(nblk + sc - 1, dim == 1 ? BK_SINGLE : (dim == 3 ? BK_PYR3 : BK_PYR4), nr, nb, 1, (nfix + 4 * (nblk + sc - 1 - nfix)) * w4, ic, RT_CONTACT, 0);
        int last = 63;
        if (over) { flags |= 2; stop = true; last = firstover - 1; }
        if (last >= 0) { const int ul = __builtin_amdgcn_readfirstlane(last); nefc += __builtin_amdgcn_readlane(sr, ul); nbrow += __builtin_amdgcn_readlane(sb, ul); nblk += __builtin_amdgcn_readlane(sc, ul); }   // (last is uniform: a v_readlane each instead of a ds_bpermute round trip)
      }
    }
    nefc = __builtin_amdgcn_readfirstlane(nefc); nblk = __builtin_amdgcn_readfirstlane(nblk); nbrow = __builtin_amdgcn_readfirstlane(nbrow);
    // assemble launch of a cohort whose solve runs the dense row-space solver (the host queues mjh_dense_build_kernel behind this
    // launch: same rule there).  Such an env stores Y = J L^-1 where the others store B = J M^-1, and needs no per-block matrices.
    const bool dense_pre = NROW == 8 && !DIAGM && pre && (xflags & XF_DENSE) && nefc > 0 && nefc <= M.dense_cap;
    WSYNC();
    PROF(6);
    // ---- base Jacobian rows: lanes = (block, base) tasks.  Storage is interleaved per block, J[b][k][4]
    //      (k = compact dof index over the up-to-two trees the block touches), so that the solver reads
    //      the 4 base entries of one dof with a single ds_read_b128.
    for (int t = lane; t < 4 * nblk; t += 64) {
      const int b = t >> 2, jb = t & 3;
      int* hd = s_blki_i + b * BLKI_STRIDE;
      const int nb = (hd[0] >> 8) & 15;
      const int SL = BLK_SLOTS(hd[1]);
      if (jb >= SL) continue;
      float* J = s_J + BLK_JOFF(hd[0]) + jb;
      for (int k = 0; k < rowW; k++) J[SL*k] = 0;
      if (jb >= nb) continue;
      const int id = hd[1] & 0xffffff, rtype = (hd[1] >> 24) & 15, side = (hd[1] >> 28) & 1;
      int t1 = -1, t2 = -1;
      if (rtype == RT_EQ) {
        const int j1 = eq_obj1id[id], j2 = eq_obj2id[id];
        const float* dat = eq_data + 11*id;
        const int d1 = jnt_dofadr[j1];
        t1 = dof_treeid[d1];
        J[SL*(d1 - tree_dofadr[t1])] = 1;
        if (j2 >= 0) {
          const int d2 = jnt_dofadr[j2];
          const float p2 = s_qpos[jnt_qposadr[j2]] - qpos0[jnt_qposadr[j2]];
          const float deriv = dat[1] + p2*(2*dat[2] + p2*(3*dat[3] + p2*4*dat[4]));
          const int tt = dof_treeid[d2];
          if (tt == t1) J[SL*(d2 - tree_dofadr[t1])] += -deriv;
          else { t2 = tt; J[SL*(tree_dofnum[t1] + d2 - tree_dofadr[t2])] = -deriv; }
        }
      } else if (rtype == RT_FL) { t1 = dof_treeid[id]; J[SL*(id - tree_dofadr[t1])] = 1; }
      else if (rtype == RT_LIMIT) { const int d = jnt_dofadr[id]; t1 = dof_treeid[d]; J[SL*(d - tree_dofadr[t1])] = side ? -1.0f : 1.0f; }
      else if (EXTRA && rtype == RT_WELD) {
        // one row of a connect / weld equality (conventions: include/mjhip.h, oracle: orc_make_constraint): rows 0..2 = world
        // components of p1 - p2 (point Jacobians), rows 3..5 = torquescale * vec(q1^-1 q2 relpose), whose rate is
        // 1/2 torquescale vec(q1^-1 (w2 - w1) q2 relpose)
        const int e = id & 0xffff, r = id >> 16;
        const int b1 = eq_obj1id[e], b2 = eq_obj2id[e];
        const bool weld = M.I[M.o_eq_type + e] == MJH_EQ_WELD;
        const float* dat = eq_data + 11*e;
        t1 = body_treeid[b1]; t2 = body_treeid[b2];
        if (t1 < 0) { t1 = t2; t2 = -1; }
        if (t2 == t1) t2 = -1;
        float q1i[4] = {s_xquat[4*b1], -s_xquat[4*b1+1], -s_xquat[4*b1+2], -s_xquat[4*b1+3]}, q2r[4];
        { const float rel[4] = {dat[6], dat[7], dat[8], dat[9]}; mulquat(q2r, s_xquat + 4*b2, rel); }
        const float ts = dat[10];
#pragma unroll
        for (int sd = 0; sd < 2; sd++) {
          const int bd = sd ? b2 : b1; const float ss = sd ? -1.0f : 1.0f;
          int i = bd > 0 ? body_lastdof[bd] : -1;
          if (i < 0 || t1 < 0) continue;
          const float* an = weld ? (sd ? dat : dat + 3) : (sd ? dat + 3 : dat);      // the anchor in THIS body's frame
          float pw[3]; rotvec(pw, s_xmat + 9*bd, an);
          const float* com = s_com + 3*body_rootid[bd];
          const float off[3] = {s_xpos[3*bd] + pw[0] - com[0], s_xpos[3*bd+1] + pw[1] - com[1], s_xpos[3*bd+2] + pw[2] - com[2]};
          const int tr = dof_treeid[i];
          const int o = (tr == t1) ? -tree_dofadr[t1] : tree_dofnum[t1] - tree_dofadr[t2];
          for (; i >= 0; i = s_dofpar_i[i]) {
            const float* cd = s_cdof + 6*i;
            float v;
            if (r < 3) { float cr[3]; cross3(cr, cd, off); v = ss * (cd[3 + r] + cr[r]); }
            else {
              const float w[4] = {0.0f, -ss * cd[0], -ss * cd[1], -ss * cd[2]};     // w2 - w1: body 2 enters with +, body 1 with -
              float u[4], vv[4]; mulquat(u, q1i, w); mulquat(vv, u, q2r);
              v = 0.5f * ts * vv[1 + (r - 3)];
            }
            J[SL*(o + i)] += v;
          }
        }
      }
      else {
        const float* c = s_con + id * CON_STRIDE;
        const int g1 = CON_G1(c), g2 = CON_G2(c);
        const int b1 = geom_bodyid[g1], b2 = geom_bodyid[g2];
        t1 = body_treeid[b1]; t2 = body_treeid[b2];
        const int t1raw = t1, t2raw = t2;                    // trees of geom1's / geom2's body (-1: static)
        if (t1 < 0) { t1 = t2; t2 = -1; }
        if (t2 == t1) t2 = -1;
        const float* dir = c + 4 + 3 * (jb < 3 ? jb : 0);   // base 0..2: translation along n,t1,t2 ; base 3: rotation about n
        // friction bases carry their coefficient (J_k <- mu_k J_k), so every pyramid row is simply J_n +- J_k
        const float musc = jb == 0 ? 1.0f : (jb < 3 ? fmaxf(geom_friction[3*g1], geom_friction[3*g2]) : fmaxf(geom_friction[3*g1+1], geom_friction[3*g2+1]));
#pragma unroll
        for (int sd = 0; sd < 2; sd++) {
          const int bd = sd ? b2 : b1; const float ss = (sd ? 1.0f : -1.0f) * musc;
          if constexpr (DIAGM) {
            // every tree a free body about its own centre (cdof: translations along the world axes, rotations about the body
            // axes through the centre): the point Jacobian in closed form, no walk over the dof chain
            const int tr = sd ? t2raw : t1raw;
            if (tr < 0) continue;
            const int o = (tr == t1) ? 0 : 6;
            const float* com = s_com + 3*bd;
            const float off[3] = {c[1] - com[0], c[2] - com[1], c[3] - com[2]};
            const float* xm = s_xmat + 9*bd;
            float tq[3];
            if (jb < 3) { cross3(tq, off, dir); J[SL*o] = ss * dir[0]; J[SL*(o+1)] = ss * dir[1]; J[SL*(o+2)] = ss * dir[2]; }
            else { tq[0] = dir[0]; tq[1] = dir[1]; tq[2] = dir[2]; }
#pragma unroll
            for (int k = 0; k < 3; k++) J[SL*(o+3+k)] = ss * (xm[k] * tq[0] + xm[3+k] * tq[1] + xm[6+k] * tq[2]);
            continue;
          }
          int i = body_lastdof[bd];
          if (i < 0) continue;
          const float* com = s_com + 3*body_rootid[bd];
          const float off[3] = {c[1] - com[0], c[2] - com[1], c[3] - com[2]};
          const int tr = dof_treeid[i];
          const int o = (tr == t1) ? -tree_dofadr[t1] : tree_dofnum[t1] - tree_dofadr[t2];
          for (; i >= 0; i = s_dofpar_i[i]) {
            const float* cd = s_cdof + 6*i;
            float v;
            if (jb < 3) { float cr[3]; cross3(cr, cd, off); const float jp[3] = {cd[3] + cr[0], cd[4] + cr[1], cd[5] + cr[2]}; v = dot3(dir, jp); }
            else v = dot3(dir, cd);
            J[SL*(o + i)] += ss * v;
          }
        }
      }
      if (jb == 0) {
        if (EXTRA && t1 < 0) { hd[2] = 0xffff; hd[3] = 0xffff; }     // (an equality between two static bodies: no dofs, an inert row)
        else { hd[2] = tree_dofadr[t1] | (tree_dofnum[t1] << 16); hd[3] = t2 >= 0 ? (tree_dofadr[t2] | (tree_dofnum[t2] << 16)) : 0xffff; }   // no second tree: n2 = 0, a2 matches no dof
      }
    }
    // ---- block parameters: impedance, regulariser R, reference gains (lanes = blocks)
    auto p_dinv = [&](int d) __attribute__((always_inline)) { return S.p_dof_invweight0 ? S.p_dof_invweight0[(size_t)env * S.p_stride + d] : dof_invweight0[d]; };
    auto p_binv = [&](int i) __attribute__((always_inline)) { return S.p_body_invweight0 ? S.p_body_invweight0[(size_t)env * S.p_stride + i] : body_invweight0[i]; };
    for (int b = lane; b < nblk; b += 64) {
      const int* hd = s_blki_i + b * BLKI_STRIDE;
      const int id = hd[1] & 0xffffff, rtype = (hd[1] >> 24) & 15, side = (hd[1] >> 28) & 1;
      float* bf = s_blkf + b * BLKF_STRIDE;
      for (int k = 0; k < BLKF_STRIDE; k++) bf[k] = 0;   // unused rows / bases must be inert (dual-block solver)
      float pos = 0, margin = 0, diagA = 0, fl = 0, mu1 = 0, mu3 = 0, rscale = -1, sr[2], si[5];
      if (rtype == RT_EQ) {
        const int j1 = eq_obj1id[id], j2 = eq_obj2id[id];
        const float* dat = eq_data + 11*id;
        const float pos1 = s_qpos[jnt_qposadr[j1]] - qpos0[jnt_qposadr[j1]];
        diagA = p_dinv(jnt_dofadr[j1]);
        if (j2 >= 0) {
          const float p2 = s_qpos[jnt_qposadr[j2]] - qpos0[jnt_qposadr[j2]];
          pos = pos1 - (dat[0] + p2*(dat[1] + p2*(dat[2] + p2*(dat[3] + p2*dat[4]))));
          diagA += p_dinv(jnt_dofadr[j2]);
        } else pos = pos1 - dat[0];
        sr[0] = eq_solref[2*id]; sr[1] = eq_solref[2*id+1];
#pragma unroll
        for (int q = 0; q < 5; q++) si[q] = eq_solimp[5*id+q];
      } else if (rtype == RT_FL) {
        diagA = p_dinv(id); fl = dof_frictionloss[id];
        sr[0] = dof_solref[2*id]; sr[1] = dof_solref[2*id+1];
#pragma unroll
        for (int q = 0; q < 5; q++) si[q] = dof_solimp[5*id+q];
      } else if (rtype == RT_LIMIT) {
        const float val = s_qpos[jnt_qposadr[id]];
        pos = side ? jnt_range[2*id+1] - val : val - jnt_range[2*id];
        margin = jnt_margin[id]; diagA = p_dinv(jnt_dofadr[id]);
        sr[0] = jnt_solref[2*id]; sr[1] = jnt_solref[2*id+1];
#pragma unroll
        for (int q = 0; q < 5; q++) si[q] = jnt_solimp[5*id+q];
      } else if (EXTRA && rtype == RT_WELD) {
        const int e = id & 0xffff, r = id >> 16;
        const int b1 = eq_obj1id[e], b2 = eq_obj2id[e];
        const bool weld = M.I[M.o_eq_type + e] == MJH_EQ_WELD;
        const float* dat = eq_data + 11*e;
        if (r < 3) {
          float p1[3], p2[3];
          rotvec(p1, s_xmat + 9*b1, weld ? dat + 3 : dat); rotvec(p2, s_xmat + 9*b2, weld ? dat : dat + 3);
          pos = (s_xpos[3*b1 + r] + p1[r]) - (s_xpos[3*b2 + r] + p2[r]);
          diagA = p_binv(2*b1) + p_binv(2*b2);
        } else {
          const float q1i[4] = {s_xquat[4*b1], -s_xquat[4*b1+1], -s_xquat[4*b1+2], -s_xquat[4*b1+3]}, rel[4] = {dat[6], dat[7], dat[8], dat[9]};
          float q2r[4], qe[4]; mulquat(q2r, s_xquat + 4*b2, rel); mulquat(qe, q1i, q2r);
          pos = dat[10] * qe[1 + (r - 3)];
          diagA = p_binv(2*b1+1) + p_binv(2*b2+1);
        }
        sr[0] = eq_solref[2*e]; sr[1] = eq_solref[2*e+1];
#pragma unroll
        for (int q = 0; q < 5; q++) si[q] = eq_solimp[5*e+q];
      } else {
        const float* c = s_con + id * CON_STRIDE;
        const int g1 = CON_G1(c), g2 = CON_G2(c), dim = CON_DIM(c);
        const int b1 = geom_bodyid[g1], b2 = geom_bodyid[g2];
        pos = c[0]; margin = c[CON_MARGIN];
        // contact parameters (mj_contactParam): max friction, solmix-weighted solref/solimp
        const float f0 = fmaxf(geom_friction[3*g1], geom_friction[3*g2]), f1 = fmaxf(geom_friction[3*g1+1], geom_friction[3*g2+1]);
        mu1 = f0; mu3 = f1;
        const float tran = p_binv(2*b1) + p_binv(2*b2);
        if (dim == 1) diagA = tran;
        else { diagA = tran + f0*f0*tran; const float mu0 = f0 * rsqrtf(M.impratio); rscale = 2 * mu0 * mu0; }  // Rpy = 2 mu^2 R(first row)
        const float a = geom_solmix[g1], bq = geom_solmix[g2];
        const float mix = (a >= MJ_MINVAL && bq >= MJ_MINVAL) ? a / (a + bq) : ((a < MJ_MINVAL && bq < MJ_MINVAL) ? 0.5f : (a < MJ_MINVAL ? 0.0f : 1.0f));
        sr[0] = mix*geom_solref[2*g1] + (1-mix)*geom_solref[2*g2]; sr[1] = mix*geom_solref[2*g1+1] + (1-mix)*geom_solref[2*g2+1];
#pragma unroll
        for (int q = 0; q < 5; q++) si[q] = mix*geom_solimp[5*g1+q] + (1-mix)*geom_solimp[5*g2+q];
      }
      const float imp = get_impedance(si, pos, margin);
      float R = fmaxf(MJ_MINVAL, (1 - imp) * diagA / imp);
      if (rscale > 0) R = fmaxf(MJ_MINVAL, rscale * R);
      float sr0 = sr[0], K, Bc; const float sr1 = sr[1], dmax = fminf(MAXIMP, fmaxf(MINIMP, si[1]));
      if (sr0 > 0) {
        if (!(M.disableflags & MJH_DSBL_REFSAFE)) sr0 = fmaxf(sr0, 2 * h);
        K = 1 / fmaxf(MJ_MINVAL, dmax*dmax * sr0*sr0 * sr1*sr1); Bc = 2 / fmaxf(MJ_MINVAL, dmax * sr0);
      } else { K = -sr0 / fmaxf(MJ_MINVAL, dmax*dmax); Bc = -sr1 / fmaxf(MJ_MINVAL, dmax); }
      if (rtype == RT_FL) K = 0;
      (void)mu1; (void)mu3;   // friction coefficients are folded into the base rows
      bf[0] = R; bf[1] = fl; bf[2] = K * imp * (pos - margin); bf[3] = Bc;
    }
    WSYNC();
    PROF(7);
    // ---- B = M^-1 J^T per base row (skipped when every tree has a diagonal M: B_d = J_d / M_dd on the fly)
    if (!DIAGM && NROW == 8 && M.k1_floats >= 16 * rowW) {
      // many-body layout: the rows live in global memory and the back-substitutions are long read-modify-write chains, so
      // every lane solves its row in a private LDS vector (dead position-stage arrays) and writes the result back once
      const int P = min(64, M.k1_floats / rowW);        // rows per pass
      for (int t0 = 0; t0 < 4 * nblk; t0 += P) {
        const int t = t0 + lane;
        if (lane >= P || t >= 4 * nblk) continue;
        const int b = t >> 2, jb = t & 3;
        const int* hd = s_blki_i + b * BLKI_STRIDE;
        const int SL = BLK_SLOTS(hd[1]);
        if (jb >= SL) continue;
        const float* J = s_J + BLK_JOFF(hd[0]) + jb; float* B = s_B + BLK_JOFF(hd[0]) + jb;
        if (jb >= ((hd[0] >> 8) & 15)) { for (int k = 0; k < rowW; k++) B[SL*k] = 0; continue; }
        ROW_TREES(hd[2], hd[3]);
        float* x = lds + M.scratch_off + lane;          // entry k of the lane's vector at x[k P]: consecutive lanes on consecutive banks
        for (int k = 0; k < rowW; k++) x[k * P] = J[SL*k];
        if (dense_pre) {
          solve_tree_lt(x - a1 * P, s_qLD, s_dofpar_i, s_dofMadr_i, a1, n1, P);
          if (n2 > 0) solve_tree_lt(x + (n1 - a2) * P, s_qLD, s_dofpar_i, s_dofMadr_i, a2, n2, P);
        } else {
          solve_tree(x - a1 * P, s_qLD, s_qLDinv, s_dofpar_i, s_dofMadr_i, a1, n1, P);
          if (n2 > 0) solve_tree(x + (n1 - a2) * P, s_qLD, s_qLDinv, s_dofpar_i, s_dofMadr_i, a2, n2, P);
        }
        for (int k = 0; k < rowW; k++) B[SL*k] = x[k * P];
      }
      WSYNC();
    } else if (!DIAGM) {
      for (int t = lane; t < 4 * nblk; t += 64) {
        const int b = t >> 2, jb = t & 3;
        const int* hd = s_blki_i + b * BLKI_STRIDE;
        const int SL = BLK_SLOTS(hd[1]);
        if (jb >= SL) continue;
        const float* J = s_J + BLK_JOFF(hd[0]) + jb; float* B = s_B + BLK_JOFF(hd[0]) + jb;
        if (jb >= ((hd[0] >> 8) & 15)) { for (int k = 0; k < rowW; k++) B[SL*k] = 0; continue; }
        ROW_TREES(hd[2], hd[3]);
        for (int k = 0; k < rowW; k++) B[SL*k] = J[SL*k];
        if (dense_pre) {
          solve_tree_lt(B - SL*a1, s_qLD, s_dofpar_i, s_dofMadr_i, a1, n1, SL);
          if (n2 > 0) solve_tree_lt(B + SL*(n1 - a2), s_qLD, s_dofpar_i, s_dofMadr_i, a2, n2, SL);
          continue;
        }
        solve_tree(B - SL*a1, s_qLD, s_qLDinv, s_dofpar_i, s_dofMadr_i, a1, n1, SL);
        if (n2 > 0) solve_tree(B + SL*(n1 - a2), s_qLD, s_qLDinv, s_dofpar_i, s_dofMadr_i, a2, n2, SL);
      }
      WSYNC();
    }
    // ---- Gauss-Seidel visiting order (shared with the oracle).  Sequence: the blocks that couple two kinematic trees
    //      first (they are the hard ones to pair), then the single-tree blocks, each group in constraint order.  Then
    //      greedy: a block, and the first later unvisited block of the sequence that shares no tree with it.  Pairs
    //      (i, q) are independent and can be solved side by side.
    int ngrp = 0;
    bool patch_order = false;    // contact-patch sweep (patch_pgs.h): it builds its own schedule
    if constexpr (DIAGM && NROW <= 2) patch_order = M.patch != 0;
    if constexpr (WPRE != 0) patch_order = true;      // (assemble-only instances: the window kernel sweeps in row order and needs no schedule — nothing of it is built, its LDS is not even allocated: engine.hip)
    if (!patch_order && M.pgs_row_order) {
      // mj_solPGS's own order: block after block as the rows were made.  Blocks without a common kinematic tree touch disjoint dofs, so
      // their updates commute exactly; pgs_row_order 1 list-schedules the sequence — block i goes to the first group (with a free place)
      // after every EARLIER block that shares a tree with it — and the sweeps that solve several blocks side by side (dual: 2, many-body:
      // 4 or 16 per wave-step) produce the sequential sweep's iterates bit for bit.  pgs_row_order 2 (and the forms that are
      // sequential anyway, and models with more than 64 trees): one block per group.
      bool listed = false;
      auto tree_of = [&](const int adr) __attribute__((always_inline)) { if constexpr (DIAGM) return adr / 6; else return (int)dof_treeid[adr]; };
      if constexpr (NROW <= 2) {
        if (M.pgs_row_order == 1 && nblk <= 64 && M.ntree <= 64) {
          // dual-block sweep: lanes = blocks; lastv: lane t = first group tree t is free again; cntv: lane g = blocks of group g
          listed = true;
          int t1 = -1, t2 = -1;
          if (lane < nblk) {
            const int* hd = s_blki_i + lane * BLKI_STRIDE;
            const int a = hd[2] & 0xffff;
            if (a != 0xffff) t1 = tree_of(a);
            if ((hd[3] >> 16) != 0) t2 = tree_of(hd[3] & 0xffff);
          }
          int lastv = 0, cntv = 0, mygrp = 0, myrank = 0;
          for (int i = 0; i < nblk; i++) {
            const int a = __builtin_amdgcn_readlane(t1, i), b = __builtin_amdgcn_readlane(t2, i);
            const int ea = a >= 0 ? __builtin_amdgcn_readlane(lastv, a) : 0, eb = b >= 0 ? __builtin_amdgcn_readlane(lastv, b) : 0;
            const int e = max(ea, eb);
            const unsigned long long bal = __ballot(cntv < 2 && lane >= e);
            const int g = __ffsll((long long)bal) - 1;
            const int c = __builtin_amdgcn_readlane(cntv, g);
            if (lane == i) { mygrp = g; myrank = c; }
            if (lane == g) cntv++;
            if (lane == a || lane == b) lastv = g + 1;
            ngrp = max(ngrp, g + 1);
          }
          const int gend = wave_incl_scan_i(lane < ngrp ? cntv : 0, lane);
          if (lane < ngrp && cntv < 2) s_sched_i[2*lane + 1] = -1;
          if (lane < nblk) { s_sched_i[2*mygrp + myrank] = lane; s_order_i[__shfl(gend, mygrp) - __shfl(cntv, mygrp) + myrank] = lane; }
        }
      }
      if constexpr (NROW == 8) {
        int* info = (int*)(lds + M.scratch_off);        // [nblk] group | rank << 16 of every block, then [nblk] blocks per group
        if (M.pgs_row_order == 1 && nblk > 64 && M.rowW <= 16 && M.ntree <= 64 && 2 * nblk <= M.k1_floats && nblk < 2048) {   // (rowW <= 16: the forms that solve blocks side by side)
          listed = true;
          int* gcnt = info + nblk;
          const int cap = M.group_max;
          for (int i = lane; i < nblk; i += 64) gcnt[i] = 0;
          WSYNC();
          int lastv = 0;
          for (int base = 0; base < nblk; base += 64) {
            int t1 = -1, t2 = -1;
            if (base + lane < nblk) {
              const int* hd = s_blki_i + (base + lane) * BLKI_STRIDE;
              const int a = hd[2] & 0xffff;
              if (a != 0xffff) t1 = tree_of(a);
              if ((hd[3] >> 16) != 0) t2 = tree_of(hd[3] & 0xffff);
            }
            const int nb = min(64, nblk - base);
            for (int i = 0; i < nb; i++) {
              const int a = __builtin_amdgcn_readlane(t1, i), b = __builtin_amdgcn_readlane(t2, i);
              const int ea = a >= 0 ? __builtin_amdgcn_readlane(lastv, a) : 0, eb = b >= 0 ? __builtin_amdgcn_readlane(lastv, b) : 0;
              int e = max(ea, eb), g = -1, c = 0;
              while (g < 0) {                           // first group >= e with a free place (group nblk - 1 at the latest)
                const int gg = e + lane;
                const int cc = gg < nblk ? gcnt[gg] : cap;
                const unsigned long long bal = __ballot(cc < cap);
                if (bal) { const int q = __ffsll((long long)bal) - 1; g = e + q; c = __builtin_amdgcn_readlane(cc, q); } else e += 64;
              }
              if (lane == 0) { gcnt[g] = c + 1; info[base + i] = g | (c << 16); }
              if (lane == a || lane == b) lastv = g + 1;
              ngrp = max(ngrp, g + 1);
              WSYNC();
            }
          }
          // group starts (exclusive prefix of the counts), then the blocks to their places
          int carry = 0;
          for (int base = 0; base < ngrp; base += 64) {
            const int g = base + lane;
            const int cnt = g < ngrp ? gcnt[g] : 0;
            const int incl = wave_incl_scan_i(cnt, lane);
            if (g < ngrp) s_sched_i[g] = carry + incl - cnt;
            carry += wave_last_i(incl);
          }
          if (lane == 0) s_sched_i[ngrp] = nblk;
          WSYNC();
          for (int i = lane; i < nblk; i += 64) { const int w = info[i]; s_order_i[s_sched_i[w & 0xffff] + (w >> 16)] = i; }
        }
      }
      if (!listed) {
        for (int i = lane; i < nblk; i += 64) {
          s_order_i[i] = i;
          if (NROW <= 2) { s_sched_i[2*i] = i; s_sched_i[2*i+1] = -1; } else if (nblk > 64) s_sched_i[i] = i;
        }
        if (NROW > 2 && nblk > 64 && lane == 0) s_sched_i[nblk] = nblk;
        ngrp = nblk;
      }
    } else if (!patch_order) {
      if (nblk > 64) {
        // many-block models: groups of up to 4 mutually independent blocks (the many-body solver puts one block on each
        // 16-lane row of the wave; sequential sweeps just follow the order).  Same rule as the oracle: two-tree blocks
        // first, then first fit: a block plus up to three later unvisited blocks of the sequence that share no tree with any
        // block already in the group.  s_order_i[k] = k-th block, s_sched_i[g] = first k of group g (s_sched_i[ngrp] = nblk).
        // Scratch: one packed word per sequence position (block | tree1 | tree2+1 | used) in LDS that is dead here.
        int* info = (int*)(NROW == 8 ? lds + M.scratch_off : s_bv);
        const bool fits = (NROW != 8 || nblk <= M.k1_floats) && nblk < 2048 && nv < 1023;
        if (!fits) {   // (cannot happen with the capacities the host accepts; keep a valid order)
          for (int i = lane; i < nblk; i += 64) { s_order_i[i] = i; s_sched_i[i] = i; }
          if (lane == 0) s_sched_i[nblk] = nblk;
          ngrp = nblk; flags |= 2;
        } else {
          int c = 0;
          for (int pass = 0; pass < 2; pass++)
            for (int base = 0; base < nblk; base += 64) {
              const int b = base + lane;
              int w = 0; bool mine = false;
              if (b < nblk) {
                const int* hd = s_blki_i + b * BLKI_STRIDE;
                const int ta = hd[2] & 0xffff, tb1 = (hd[3] >> 16) ? (hd[3] & 0xffff) + 1 : 0;
                mine = (tb1 != 0) == (pass == 0);
                w = b | (ta << 11) | (tb1 << 21);
              }
              const unsigned long long mk = __ballot(mine);
              if (mine) info[c + __popcll(mk & ((1ull << lane) - 1ull))] = w;
              c += __popcll(mk);
            }
          WSYNC();
          int k = 0, cursor = 0;
          while (k < nblk) {
            // first unvisited position
            int p0 = -1;
            for (int base = cursor; base < nblk && p0 < 0; base += 64) {
              const int p = base + lane;
              const unsigned long long mk = __ballot(p < nblk && info[p] >= 0);
              if (mk) p0 = base + __ffsll((long long)mk) - 1;
            }
            int w0 = info[p0];
            int gt0 = (w0 >> 11) & 1023, gt1 = ((w0 >> 21) & 1023) - 1, gt2 = -2, gt3 = -2, gt4 = -2, gt5 = -2;   // trees of the group (-2: none)
            if (lane == 0) { info[p0] = w0 | 0x80000000; s_sched_i[ngrp] = k; s_order_i[k] = w0 & 2047; }
            k++; cursor = p0 + 1;
            int cnt = 1, pos = p0 + 1;
            WSYNC();
            if (M.group_max == 16) {
              // wide groups: up to 16 blocks, the trees of the group as a bitmask (<= 64 trees by the host's rule)
              auto tmask = [&](int w) __attribute__((always_inline)) {
                const int ta = (w >> 11) & 1023, tb = ((w >> 21) & 1023) - 1;
                if constexpr (DIAGM) return (1ull << (ta / 6)) | (tb >= 0 ? (1ull << (tb / 6)) : 0ull);   // free bodies: tree t owns dofs 6t .. 6t+5 (no table look-up inside the loop)
                return (1ull << dof_treeid[ta]) | (tb >= 0 ? (1ull << dof_treeid[tb]) : 0ull);
              };
              unsigned long long gmask = tmask(w0);
              // first fit over the sequence, one 64-position window at a time: the window's unvisited blocks that do not touch the
              // group yet are tried in order (a block accepted earlier in the window may rule a later one out); the visited
              // marks go back to LDS once per window
              while (cnt < 16 && pos < nblk) {
                const int p = pos + lane;
                int w = -1; unsigned long long tm = 0;
                if (p < nblk) { w = info[p]; tm = tmask(w); }
                unsigned long long mk = __ballot(w >= 0 && !(tm & gmask));
                bool took = false;
                while (mk && cnt < 16) {
                  const int q = __ffsll((long long)mk) - 1;
                  mk &= mk - 1ull;
                  const unsigned long long tq = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(tm >> 32), q) << 32) | (unsigned)__builtin_amdgcn_readlane((int)tm, q);
                  if (tq & gmask) continue;
                  const int wq = __builtin_amdgcn_readlane(w, q);
                  if (lane == 0) s_order_i[k] = wq & 2047;
                  if (lane == q) { w |= 0x80000000; took = true; }
                  gmask |= tq;
                  k++; cnt++;
                }
                if (took) info[p] = w;
                pos += 64;
              }
              WSYNC();
            } else
            while (cnt < 4 && pos < nblk) {
              const int p = pos + lane;
              bool cand = false; int w = 0;
              if (p < nblk) {
                w = info[p];
                const int ta = (w >> 11) & 1023, tb = ((w >> 21) & 1023) - 1;
                const bool sh = ta == gt0 || ta == gt1 || ta == gt2 || ta == gt3 || ta == gt4 || ta == gt5 ||
                                (tb >= 0 && (tb == gt0 || tb == gt1 || tb == gt2 || tb == gt3 || tb == gt4 || tb == gt5));
                cand = w >= 0 && !sh;
              }
              const unsigned long long mk = __ballot(cand);
              if (!mk) { pos += 64; continue; }
              const int q = __ffsll((long long)mk) - 1;
              const int wq = __builtin_amdgcn_readlane(w, q);
              if (lane == 0) { info[pos + q] = wq | 0x80000000; s_order_i[k] = wq & 2047; }
              const int ta = (wq >> 11) & 1023, tb = ((wq >> 21) & 1023) - 1;
              if (cnt == 1) { gt2 = ta; gt3 = tb >= 0 ? tb : -2; } else { gt4 = ta; gt5 = tb >= 0 ? tb : -2; }
              k++; cnt++; pos = pos + q + 1;
              WSYNC();
            }
            ngrp++;
          }
          if (lane == 0) s_sched_i[ngrp] = nblk;
          if (NROW <= 2) {   // the dual-block sweep reads (block, partner) pairs: follow the order, no partners
            WSYNC();
            for (int i = lane; i < nblk; i += 64) { s_sched_i[2*i] = s_order_i[i]; s_sched_i[2*i+1] = -1; }
            ngrp = nblk;
          }
        }
      } else {
        {   // stable partition (two-tree blocks first): position of block `lane` in the sequence
          const bool two = lane < nblk && (s_blki_i[lane * BLKI_STRIDE + 3] >> 16) != 0;
          const unsigned long long m2 = __ballot(two), m1 = __ballot(lane < nblk && !two), lt = (1ull << lane) - 1ull;
          const int pos = two ? __popcll(m2 & lt) : __popcll(m2) + __popcll(m1 & lt);
          if (lane < nblk) s_order_i[pos] = lane;
        }
        WSYNC();
        int ta = -1, tb = -1, bid = -1;   // lanes = sequence positions
        if (lane < nblk) { bid = s_order_i[lane]; const int* hd = s_blki_i + bid * BLKI_STRIDE; ta = hd[2] & 0xffff; tb = (hd[3] >> 16) ? (hd[3] & 0xffff) : -1; }
        WSYNC();
        unsigned long long used = 0; int k = 0;
        for (int i = 0; i < nblk; i++) {
          if ((used >> i) & 1ull) continue;
          used |= 1ull << i;
          const int ia = __builtin_amdgcn_readlane(ta, i), ib = __builtin_amdgcn_readlane(tb, i);
          const bool share = ta == ia || (ib >= 0 && ta == ib) || (tb >= 0 && (tb == ia || tb == ib));
          const bool cand = lane < nblk && lane > i && !((used >> lane) & 1ull) && !share;
          const unsigned long long bal = __ballot(cand);
          int q = -1;
          if (bal) { q = __ffsll((long long)bal) - 1; used |= 1ull << q; }
          const int bi = __builtin_amdgcn_readlane(bid, i), bq = q >= 0 ? __builtin_amdgcn_readlane(bid, q) : -1;
          if (lane == 0) { s_sched_i[2*ngrp] = bi; s_sched_i[2*ngrp+1] = bq; s_order_i[k] = bi; if (q >= 0) s_order_i[k+1] = bq; }
          k += q >= 0 ? 2 : 1; ngrp++;
        }
      }
    }
    WSYNC();
    PROF(8);
    if (post) {   // the blocks were built by the PH_PRE launch and live in the pools
      const int* meta = (const int*)(gs + L.g_meta);
      nblk = __builtin_amdgcn_readfirstlane(meta[0]); nfixblk = __builtin_amdgcn_readfirstlane(meta[1]);
      nefc = __builtin_amdgcn_readfirstlane(meta[2]); ncon = __builtin_amdgcn_readfirstlane(meta[3]); flags |= __builtin_amdgcn_readfirstlane(meta[4]);
    }

    // dot products of every base row with a dof-space vector: out[4b+j] = J[b][.][j] . vec   (lanes = blocks: the four interleaved
    // base entries of a dof are one 16-byte read)
    auto base_dots = [&](const float* vec, float* out) __attribute__((always_inline)) {
      for (int b = lane; b < nblk; b += 64) {
        const int4 hd = *(const int4*)(s_blki_i + b * BLKI_STRIDE);
        ROW_TREES(hd.z, hd.w);
        float4 v = make_float4(0, 0, 0, 0);
        if (BLK_SLOTS(hd.y) == 4) {
          const float4* J4 = (const float4*)(s_J + BLK_JOFF(hd.x));
          for (int k = 0; k < n1; k++) { const float4 j = J4[k]; const float x = vec[a1 + k]; v.x += j.x * x; v.y += j.y * x; v.z += j.z * x; v.w += j.w * x; }
          for (int k = 0; k < n2; k++) { const float4 j = J4[n1 + k]; const float x = vec[a2 + k]; v.x += j.x * x; v.y += j.y * x; v.z += j.z * x; v.w += j.w * x; }
        } else {
          const float* J = s_J + BLK_JOFF(hd.x);
          for (int k = 0; k < n1; k++) v.x += J[k] * vec[a1 + k];
          for (int k = 0; k < n2; k++) v.x += J[n1 + k] * vec[a2 + k];
        }
        *(float4*)(out + 4 * b) = v;
      }
      WSYNC();
    };
    // out[d] = sum over blocks/bases of X[b][d][j] * phi[4b+j]   (lanes = dofs; X = J, or B = M^-1 J^T when useB)
    auto accum_T = [&](bool useB, const float* phi, float* out) __attribute__((always_inline)) {
      if constexpr (NROW == 8) {
        // many-body layout: the gather below is O(nv x nblk) header reads from the global pools; scatter instead, lanes =
        // blocks, each adding its <= rowW entries to the dof vector with LDS atomics (lane order: deterministic)
        for (int d = lane; d < nv; d += 64) out[d] = 0;
        WSYNC();
        for (int b = lane; b < nblk; b += 64) {
          const int4 hd = *(const int4*)(s_blki_i + b * BLKI_STRIDE);
          ROW_TREES(hd.z, hd.w);
          const float* X = ((useB && !DIAGM) ? s_B : s_J) + BLK_JOFF(hd.x);
          const float4 p = *(const float4*)(phi + 4*b);
          const bool quad = BLK_SLOTS(hd.y) == 4;
          for (int k = 0; k < n1 + n2; k++) {
            const int d = k < n1 ? a1 + k : a2 + k - n1;
            float v;
            if (quad) { const float4 x = *(const float4*)(X + 4*k); v = x.x*p.x + x.y*p.y + x.z*p.z + x.w*p.w; } else v = X[k] * p.x;
            if (v != 0) atomicAdd(&out[d], (useB && DIAGM) ? v * s_qLDinv[d] : v);
          }
        }
        WSYNC();
        return;
      }
      for (int d = lane; d < nv; d += 64) {
        float acc = 0;
        const float minv = s_qLDinv[d];
        for (int b = 0; b < nblk; b++) {
          const int4 hd = *(const int4*)(s_blki_i + b * BLKI_STRIDE);
          ROW_TREES(hd.z, hd.w);
          const int o = row_off(d, a1, n1, a2, n2);
          if (o < 0) continue;
          const float* X = ((useB && !DIAGM) ? s_B : s_J) + BLK_JOFF(hd.x);
          const float4 p = *(const float4*)(phi + 4*b);
          if (BLK_SLOTS(hd.y) == 4) { const float4 x = *(const float4*)(X + 4*o); acc += x.x*p.x + x.y*p.y + x.z*p.z + x.w*p.w; }
          else acc += X[o] * p.x;
        }
        out[d] = (useB && DIAGM) ? acc * minv : acc;
      }
      WSYNC();
    };
    // pyramid row r of a block: direction index k (1..3) and signed friction coefficient c
#define PYR_KC(r, mu1, mu3, k, c) const int k = 1 + ((r) >> 1); const float c = ((r) & 1) ? -1.0f : 1.0f
    // per-row force response to jar = J a - aref  (mj_constraintUpdate, pyramidal cones), and base forces phi
    auto forces_from = [&](const float* bv, bool keep) __attribute__((always_inline)) {
      for (int b = lane; b < nblk; b += 64) {
        const int* hd = s_blki_i + b * BLKI_STRIDE;
        float* bf = s_blkf + b * BLKF_STRIDE;
        const int kind = hd[0] & 15, nr = (hd[0] >> 4) & 15, clamp = (hd[0] >> 12) & 3, jadr = 4 * b;
        const float R = bf[0], D = 1.0f / R, flv = bf[1];
        float ph[4] = {0, 0, 0, 0};
        for (int r = 0; r < nr; r++) {
          PYR_KC(r, mu1, mu3, k, c);
          const bool pyr = kind != BK_SINGLE;
          const float jar = bv[jadr] + (pyr ? c * bv[jadr + k] : 0.0f) - (bf[BF_AREF] + (pyr ? c * bf[BF_AREF + k] : 0.0f));
          float f;
          if (clamp == 0) f = -D * jar;
          else if (clamp == 2) f = (jar <= -R*flv) ? flv : ((jar >= R*flv) ? -flv : -D * jar);
          else f = jar < 0 ? -D * jar : 0.0f;
          if (keep) bf[BF_F + r] = f;
          ph[0] += f;
          if (pyr) { if (k == 1) ph[1] += c * f; else if (k == 2) ph[2] += c * f; else ph[3] += c * f; }
        }
        for (int j = 0; j < 4; j++) s_phi[jadr + j] = ph[j];
      }
      WSYNC();
    };
    auto phi_from_forces = [&]() __attribute__((always_inline)) {
      for (int b = lane; b < nblk; b += 64) {
        const int* hd = s_blki_i + b * BLKI_STRIDE;
        const float* bf = s_blkf + b * BLKF_STRIDE;
        const int kind = hd[0] & 15, nr = (hd[0] >> 4) & 15, jadr = 4 * b;
        float ph[4] = {0, 0, 0, 0};
        for (int r = 0; r < nr; r++) {
          PYR_KC(r, bf[2], bf[3], k, c);
          const float f = bf[BF_F + r];
          ph[0] += f;
          if (kind != BK_SINGLE) { if (k == 1) ph[1] += c * f; else if (k == 2) ph[2] += c * f; else ph[3] += c * f; }
        }
        for (int j = 0; j < 4; j++) s_phi[jadr + j] = ph[j];
      }
      WSYNC();
    };

    // ================================================================ velocity stage (lambdas: used by step1, inverse, step2-alone)
    auto vel_levels = [&](const float* qv, const float* qa, float* out) __attribute__((always_inline)) {
      // mj_comVel + mj_rne forward/backward; qa != null adds cdof*qacc (flg_acc)
      if constexpr (DIAGM) {
        // every tree a free body about its own centre with body-aligned principal axes: RNE in closed form.  Translations (world
        // frame): m (a - g); rotations (body frame): I alpha + w x (I w).  (The spatial quantities cvel / cacc / cfrc are not
        // formed: their only other reader for these models is the energy export, which has its own closed form.)
        for (int d = lane; d < nv; d += 64) {
          const int b = dof_bodyid[d], k = d - body_dofadr[b];
          const float a = qa ? qa[d] : 0.0f;
          if (k < 3) out[d] = s_p_mass[b] * (a - grav[k]);
          else {
            const float* w = qv + (d - k) + 3; const float* I = s_p_inertia + 3*b;
            const float Iw[3] = {I[0] * w[0], I[1] * w[1], I[2] * w[2]};
            float cr[3]; cross3(cr, w, Iw);
            out[d] = I[k-3] * a + cr[k-3];
          }
        }
        WSYNC();
        return;
      }
      if (lane == 0) { for (int k = 0; k < 6; k++) { s_cvel[k] = 0; s_cacc[k] = (k >= 3) ? -grav[k-3] : 0.0f; s_cfrc[k] = 0; } }
      WSYNC();
      for (int lev = 1; lev <= M.maxlevel; lev++) {
        for (int b = lane; b < nbody; b += 64) {
          if (body_level[b] != lev) continue;
          const int p = body_parentid[b];
          float cv[6], ca[6];
#pragma unroll
          for (int q = 0; q < 6; q++) { cv[q] = s_cvel[6*p+q]; ca[q] = s_cacc[6*p+q]; }
          int bda = body_dofadr[b];
          for (int j = 0; j < body_jntnum[b]; j++) {
            const int jt = jnt_type[body_jntadr[b] + j];
            if (jt == MJH_JNT_FREE) {
              for (int k = 0; k < 3; k++) {
#pragma unroll
                for (int q = 0; q < 6; q++) { s_cdofdot[6*(bda+k)+q] = 0; cv[q] += s_cdof[6*(bda+k)+q] * qv[bda+k]; }
              }
              bda += 3;
            }
            const int nd = (jt == MJH_JNT_FREE || jt == MJH_JNT_BALL) ? 3 : 1;
            float cvn[6];
#pragma unroll
            for (int q = 0; q < 6; q++) cvn[q] = cv[q];
            for (int k = 0; k < nd; k++) {
              float cd[6], cdd[6];
#pragma unroll
              for (int q = 0; q < 6; q++) cd[q] = s_cdof[6*(bda+k)+q];
              cross_motion(cdd, cv, cd);
              const float v = qv[bda+k];
#pragma unroll
              for (int q = 0; q < 6; q++) { s_cdofdot[6*(bda+k)+q] = cdd[q]; cvn[q] += cd[q] * v; }
            }
#pragma unroll
            for (int q = 0; q < 6; q++) cv[q] = cvn[q];
            bda += nd;
          }
          for (int d = body_dofadr[b]; d < body_dofadr[b] + body_dofnum[b]; d++) {
            const float v = qv[d], a = qa ? qa[d] : 0.0f;
#pragma unroll
            for (int q = 0; q < 6; q++) ca[q] += s_cdofdot[6*d+q] * v + s_cdof[6*d+q] * a;
          }
          float ci[10], f[6], t[6], t1[6];
#pragma unroll
          for (int k = 0; k < 10; k++) ci[k] = s_cinert[10*b+k];
          mul_inert_vec(f, ci, ca); mul_inert_vec(t, ci, cv); cross_force(t1, cv, t);
#pragma unroll
          for (int q = 0; q < 6; q++) { s_cvel[6*b+q] = cv[q]; s_cacc[6*b+q] = ca[q]; s_cfrc[6*b+q] = f[q] + t1[q]; }
        }
        WSYNC();
      }
      for (int b = lane; b < nbody; b += 64) {
        float acc[6] = {0, 0, 0, 0, 0, 0};
        if (b > 0) for (int c = b; c < b + body_subtreesize[b]; c++)
#pragma unroll
          for (int q = 0; q < 6; q++) acc[q] += s_cfrc[6*c+q];
#pragma unroll
        for (int q = 0; q < 6; q++) s_cfrcsub[6*b+q] = acc[q];
      }
      WSYNC();
      for (int d = lane; d < nv; d += 64) {
        float v = 0; const float* f = s_cfrcsub + 6*dof_bodyid[d];
#pragma unroll
        for (int q = 0; q < 6; q++) v += s_cdof[6*d+q] * f[q];
        out[d] = v;
      }
      WSYNC();
    };
    auto vel_stage = [&](const float* qv) __attribute__((always_inline)) {
      vel_levels(qv, nullptr, s_bias);
      // mj_passive: springs, dampers, gravity compensation (gravcomp: mj_sim.cpp:301-310)
      for (int d = lane; d < nv; d += 64) {
        float v = 0;
        if (!(M.disableflags & MJH_DSBL_PASSIVE)) {
          const int j = dof_jntid[d], jt = jnt_type[j];
          if ((jt == MJH_JNT_HINGE || jt == MJH_JNT_SLIDE) && jnt_stiffness[j] != 0) v -= jnt_stiffness[j] * (s_qpos[jnt_qposadr[j]] - qpos_spring[jnt_qposadr[j]]);
          v -= dof_damping[d] * qv[d];
          const int bd = dof_bodyid[d];
          if constexpr (DIAGM) {   // free body: the compensating force -m gravcomp g acts at its centre: translational dofs only
            const int k = d - body_dofadr[bd];
            if (k < 3) for (int g = 0; g < M.ngc; g++) if (gc_body[g] == bd) v -= s_p_mass[bd] * body_gravcomp[bd] * grav[k];
          } else
          for (int g = 0; g < M.ngc; g++) {
            const int b = gc_body[g];
            if (b < bd || b >= bd + body_subtreesize[bd]) continue;
            const float* com = s_com + 3*body_rootid[b];
            const float off[3] = {s_xipos[3*b] - com[0], s_xipos[3*b+1] - com[1], s_xipos[3*b+2] - com[2]};
            const float* cd = s_cdof + 6*d;
            float cr[3]; cross3(cr, cd, off);
            const float sc = -s_p_mass[b] * body_gravcomp[b];
            v += sc * ((cd[3] + cr[0]) * grav[0] + (cd[4] + cr[1]) * grav[1] + (cd[5] + cr[2]) * grav[2]);
          }
        }
        s_passive[d] = v;
      }
      // mj_referenceConstraint: aref = -B (J qvel) - K imp (pos - margin), per row of every block
      WSYNC();
      base_dots(qv, s_bv);
      for (int b = lane; b < nblk; b += 64) {
        const int* hd = s_blki_i + b * BLKI_STRIDE;
        float* bf = s_blkf + b * BLKF_STRIDE;
        const int kind = hd[0] & 15, nr = (hd[0] >> 4) & 15, jadr = 4 * b;
        // aref of row r = J_n +- J_k is  ab[0] +- ab[k]  with the per-base values  ab[j] = -Bc (J_j qvel) - (j == 0) KI
        const float KI = bf[2], Bc = bf[3];
        (void)kind; (void)nr;
#pragma unroll
        for (int j = 0; j < 4; j++) bf[BF_AREF + j] = -Bc * s_bv[jadr + j] - (j == 0 ? KI : 0.0f);
      }
      WSYNC();
    };

    if (ph & PH_STEP1) {
      vel_stage(s_qvel);
      PROF(9);
      if ((xflags & XF_FORCE) && S.x_energy) {  // mj_energyPos / mj_energyVel
        float pe = 0, ke = 0;
        for (int b = lane; b < nbody; b += 64) if (b > 0) {
          pe -= s_p_mass[b] * (grav[0]*s_xipos[3*b] + grav[1]*s_xipos[3*b+1] + grav[2]*s_xipos[3*b+2]);
          if constexpr (DIAGM) {   // free bodies: 1/2 m |v|^2 + 1/2 w . I w  (world-frame velocity, body-frame spin; see vel_levels)
            if (body_dofnum[b] == 6) {
              const float* v = s_qvel + body_dofadr[b]; const float* I = s_p_inertia + 3*b;
              ke += 0.5f * (s_p_mass[b] * (v[0]*v[0] + v[1]*v[1] + v[2]*v[2]) + I[0]*v[3]*v[3] + I[1]*v[4]*v[4] + I[2]*v[5]*v[5]);
            }
          } else {
          float t[6]; mul_inert_vec(t, s_cinert + 10*b, s_cvel + 6*b);
          for (int q = 0; q < 6; q++) ke += 0.5f * t[q] * s_cvel[6*b+q];
          }
        }
        pe = wave_sum<4>(pe); ke = wave_sum<4>(ke);
        if (lane == 0) { S.x_energy[2*xrow] = pe; S.x_energy[2*xrow+1] = ke; }
      }
      // ---- controller (MjSim::controller, mj_sim.cpp:1055-1077)
      bool anydd = false, anydq = false;
      for (int d = lane; d < nv; d += 64) {
        float a = (step == 0) ? S.ddq[vrow + d] : 0.0f, v = (step == 0) ? S.dq[vrow + d] : 0.0f;
        if (S.pd_target) {   // in-engine PD law (mjh_set_pd_controller): the command of this step, as mjh_pd_kernel would have written it
          const int j = dof_jntid[d], jt = jnt_type[j];
          if (jt == MJH_JNT_HINGE || jt == MJH_JNT_SLIDE) {
            a = S.pd_kp * (S.pd_target[(size_t)env * nv + d] - s_qpos[jnt_qposadr[j]]) - S.pd_kd * s_qvel[d];
            S.ddq[vrow + d] = a;
          }
        }
        s_tmpv[d] = a; s_tmpv2[d] = v; anydd |= a != 0; anydq |= fabsf(v) > MJ_MINVAL;
        s_applied[d] = 0; s_qvref[d] = s_qvel[d];
      }
      anydd = wave_any(anydd); anydq = wave_any(anydq);
      WSYNC();
      if (anydd) {  // tau = M ddq (mj_mulM, :1057)
        for (int i = lane; i < nv; i += 64) {
          int adr = s_dofMadr_i[i]; float vi = s_tmpv[i], acc = qM_ro[adr] * vi; int k = 1;
          for (int j = s_dofpar_i[i]; j >= 0; j = s_dofpar_i[j]) { float mij = qM_ro[adr + k]; acc += mij * s_tmpv[j]; atomicAdd(&s_applied[j], mij * vi); k++; }
          atomicAdd(&s_applied[i], acc);
        }
        WSYNC();
      }
      for (int d = lane; d < nv; d += 64) {
        if (controlled[d]) s_applied[d] += s_bias[d];            // :1058-1063
        if (fabsf(s_tmpv2[d]) > MJ_MINVAL) s_qvel[d] = s_tmpv2[d];  // :1067-1073 velocity override
        if ((step == 0 || S.pd_target) && (anydd || anydq)) { S.ddq[vrow + d] = 0; S.dq[vrow + d] = 0; }  // :1075-1076 (later steps of a launch: only the PD law has written a command)
      }
      WSYNC();
      if ((ph & PH_INV) && anydq) { vel_stage(s_qvel); for (int d = lane; d < nv; d += 64) s_qvref[d] = s_qvel[d]; WSYNC(); }
    } else if (!post) {
      // split API: re-create the velocity-stage quantities the previous call left behind
      if (ph & PH_INV) { for (int d = lane; d < nv; d += 64) s_qvref[d] = s_qvel[d]; WSYNC(); }
      vel_stage(s_qvref);
    }
    if ((xflags & XF_FORCE) && (ph & (PH_STEP1 | PH_INV))) {
      const size_t e = (size_t)xrow * M.nvp;
      for (int d = lane; d < nv; d += 64) { if (S.x_bias) S.x_bias[e + d] = s_bias[d]; if (S.x_passive) S.x_passive[e + d] = s_passive[d]; }
    }

    // ================================================================ inverse (mj_inverse, mj_hw_interface.cpp:61)
    if (ph & PH_INV) {
      // analytic constraint force at the current qacc (mj_constraintUpdate), qc = J^T f
      base_dots(s_qacc, s_bv);
      forces_from(s_bv, false);
      accum_T(false, s_phi, s_tmpv2);
      if constexpr (DIAGM) {
        vel_levels(s_qvel, s_qacc, s_tmpv);  // RNE with acceleration
        for (int d = lane; d < nv; d += 64) S.qfrc_inverse[vrow + d] = s_tmpv[d] + dof_armature[d] * s_qacc[d] - s_passive[d] - s_tmpv2[d];
      } else {
        // RNE with acceleration = M qacc + qfrc_bias (M without armature); both terms are at hand — qfrc_bias from the velocity stage of
        // this launch, M from the CRBA (armature included) — so the second pass over the tree levels is one mj_mulM instead
        for (int d = lane; d < nv; d += 64) s_tmpv[d] = 0;
        WSYNC();
        for (int i = lane; i < nv; i += 64) {
          int adr = s_dofMadr_i[i]; const float vi = s_qacc[i]; float acc = qM_ro[adr] * vi; int k = 1;
          for (int j = s_dofpar_i[i]; j >= 0; j = s_dofpar_i[j]) { const float mij = qM_ro[adr + k]; acc += mij * s_qacc[j]; atomicAdd(&s_tmpv[j], mij * vi); k++; }
          atomicAdd(&s_tmpv[i], acc);
        }
        WSYNC();
        for (int d = lane; d < nv; d += 64) S.qfrc_inverse[vrow + d] = s_tmpv[d] + s_bias[d] - s_passive[d] - s_tmpv2[d];
      }
    }

    PROF(10);
    // ================================================================ step2 (mj_step2, mj_main.cpp:108)
    if (ph & (PH_STEP2 | PH_NOINT)) {
      if (post) {
        const int* meta = (const int*)(gs + L.g_meta);
        niter = __builtin_amdgcn_readfirstlane(meta[5]);
        for (int d = lane; d < nv; d += 64) { const float qa = gs[L.g_qacc + d]; s_qacc[d] = qa; s_ws[d] = qa; s_smooth[d] = gs[L.g_smooth + d]; s_asmooth[d] = 0; s_tmpv2[d] = 0; }
        WSYNC();
        if (nefc > 0 && (xflags & XF_FORCE)) { phi_from_forces(); accum_T(false, s_phi, s_tmpv2); }
      } else {
      // ---- smooth acceleration (mj_fwdAcceleration)
      for (int d = lane; d < nv; d += 64) { float f = s_passive[d] - s_bias[d] + s_applied[d]; s_smooth[d] = f; s_asmooth[d] = f; }
      WSYNC();
      if (EXTRA && S.xfrc_applied) {
        // mj_xfrcAccumulate: d->xfrc_applied (force, torque at the body's centre of mass, world frame) as a spatial force about
        // the tree root's COM, summed over each body's subtree and projected on the dofs.  Scratch: the velocity-stage
        // spatial vectors (dead since qfrc_bias was formed).
        const float* xf = S.xfrc_applied + (size_t)env * S.xfrc_stride;
        if constexpr (DIAGM) {
          // free bodies: the force acts at the body's centre (translational dofs, world frame), the torque on the rotational dofs in
          // the body frame (R^T tau, R from the body's quaternion: the frames of the position stage are gone by now)
          for (int d = lane; d < nv; d += 64) {
            const int b = dof_bodyid[d], k = d - body_dofadr[b];
            float acc;
            if (k < 3) acc = xf[6*b + k];
            else {
              const float* q7 = s_qpos + jnt_qposadr[dof_jntid[d]];
              float qw = q7[3], qx = q7[4], qy = q7[5], qz = q7[6];
              { const float inv = 1.0f / sqrtf(qw*qw + qx*qx + qy*qy + qz*qz); qw *= inv; qx *= inv; qy *= inv; qz *= inv; }
              const float tx = xf[6*b+3], ty = xf[6*b+4], tz = xf[6*b+5];
              const int c = k - 3;     // column c of R
              const float r0 = c == 0 ? 1 - 2*(qy*qy + qz*qz) : (c == 1 ? 2*(qx*qy - qw*qz) : 2*(qx*qz + qw*qy));
              const float r1 = c == 0 ? 2*(qx*qy + qw*qz) : (c == 1 ? 1 - 2*(qx*qx + qz*qz) : 2*(qy*qz - qw*qx));
              const float r2 = c == 0 ? 2*(qx*qz - qw*qy) : (c == 1 ? 2*(qy*qz + qw*qx) : 1 - 2*(qx*qx + qy*qy));
              acc = r0 * tx + r1 * ty + r2 * tz;
            }
            s_smooth[d] += acc; s_asmooth[d] += acc;
          }
          WSYNC();
        } else {
        for (int b = lane; b < nbody; b += 64) {
          float F[6] = {0, 0, 0, 0, 0, 0};
          if (b > 0) {
            const float f[3] = {xf[6*b], xf[6*b+1], xf[6*b+2]};
            const float* com = s_com + 3*body_rootid[b];
            const float off[3] = {s_xipos[3*b] - com[0], s_xipos[3*b+1] - com[1], s_xipos[3*b+2] - com[2]};
            float cr[3]; cross3(cr, off, f);
            F[0] = xf[6*b+3] + cr[0]; F[1] = xf[6*b+4] + cr[1]; F[2] = xf[6*b+5] + cr[2]; F[3] = f[0]; F[4] = f[1]; F[5] = f[2];
          }
#pragma unroll
          for (int q = 0; q < 6; q++) s_cfrc[6*b+q] = F[q];
        }
        WSYNC();
        for (int d = lane; d < nv; d += 64) {
          const int bd = dof_bodyid[d];
          float acc = 0;
          for (int c = bd; c < bd + body_subtreesize[bd]; c++)
#pragma unroll
            for (int q = 0; q < 6; q++) acc += s_cdof[6*d+q] * s_cfrc[6*c+q];
          s_smooth[d] += acc; s_asmooth[d] += acc;
        }
        WSYNC();
        }   // !DIAGM
      }
      if (DIAGM) { for (int d = lane; d < nv; d += 64) s_asmooth[d] *= s_qLDinv[d]; }
      else {
        if (level_solves) solve_trees_levels(s_asmooth, s_qLD, s_qLDinv, s_anc_i, s_dofMadr_i, nv, M.nM, lane);
        else {
        for (int t = 0; t < M.ntree; t++) if (tree_dofnum[t] >= MJH_WAVE_TREE_MIN) solve_tree_wave(s_asmooth, s_qLD, s_qLDinv, s_anc_i, s_dofMadr_i, tree_dofadr[t], tree_dofnum[t], M.nM, nv, lane);
        for (int t = lane; t < M.ntree; t += 64) if (tree_dofnum[t] < MJH_WAVE_TREE_MIN) solve_tree(s_asmooth, s_qLD, s_qLDinv, s_dofpar_i, s_dofMadr_i, tree_dofadr[t], tree_dofnum[t]);
        }
      }
      WSYNC();
      niter = 0;
      PROF(11);
      if (pre) {   // hand-over to mjh_solve_kernel / the PH_POST launch
        int* meta = (int*)(gs + L.g_meta);
        for (int d = lane; d < nv; d += 64) { gs[L.g_qvel + d] = s_qvel[d]; gs[L.g_smooth + d] = s_smooth[d]; if (nefc == 0) gs[L.g_qacc + d] = s_asmooth[d]; }
        if (DIAGM) for (int i = lane; i < M.nM; i += 64) gs[L.g_qM + i] = s_qM[i];      // (articulated models: written by the CRBA)
        if (lane == 0) { meta[0] = nefc == 0 ? 0 : nblk; meta[1] = nfixblk; meta[2] = nefc; meta[3] = ncon; meta[4] = flags; meta[5] = 0; meta[6] = ngrp; meta[7] = 0; }   // meta[7]: set by mjh_dense_build_kernel when the dense solver takes this env
        if (nefc == 0) return;
      }
      const bool wdefer = wpre && (xflags & XF_DEFER);       // (split API through the window chain: unconstrained envs are handed over as well)
      if (nefc == 0 && !wdefer) {
        for (int d = lane; d < nv; d += 64) { s_qacc[d] = s_asmooth[d]; s_ws[d] = s_asmooth[d]; s_tmpv2[d] = 0; }
        WSYNC();
      } else {
        // ---- warm start (mj_fwdConstraint): forces implied by qacc_warmstart, kept if their dual cost < 0
        const bool warm = !(M.disableflags & MJH_DSBL_WARMSTART);
        bool zero_f = !warm;
        bool patch_ws = false;   // contact-patch sweep: the warm start runs in patch form, after the patches are built (window hand-over: in the window kernel)
        if constexpr (DIAGM && NROW <= 2) patch_ws = M.patch != 0 || wpre;
        if (warm && !patch_ws) {
          base_dots(s_ws, s_bv);
          forces_from(s_bv, true);
          accum_T(true, s_phi, s_tmpv);               // da = M^-1 J^T f
          if (dense_pre) {                            // (the pool holds Y = J L^-1: what came out is Y^T f = L^-T J^T f; finish with D^-1 and L^-1)
            for (int d = lane; d < nv; d += 64) s_tmpv[d] *= s_qLDinv[d];
            WSYNC();
            tree_l_levels(s_tmpv, s_qLD, s_anc_i, s_dofMadr_i, nv, M.nM, lane);
            WSYNC();
          }
          base_dots(s_tmpv, s_bv);                    // J da
          base_dots(s_asmooth, s_phi);                // J a_smooth  (phi is free again)
          float cost = 0;
          for (int b = lane; b < nblk; b += 64) {
            const int* hd = s_blki_i + b * BLKI_STRIDE;
            const float* bf = s_blkf + b * BLKF_STRIDE;
            const int kind = hd[0] & 15, nr = (hd[0] >> 4) & 15, jadr = 4 * b;
            for (int r = 0; r < nr; r++) {
              PYR_KC(r, bf[2], bf[3], k, c);
              const bool pyr = kind != BK_SINGLE;
              const float jda = s_bv[jadr] + (pyr ? c * s_bv[jadr + k] : 0.0f);
              const float bb = s_phi[jadr] + (pyr ? c * s_phi[jadr + k] : 0.0f) - (bf[BF_AREF] + (pyr ? c * bf[BF_AREF + k] : 0.0f));
              const float f = bf[BF_F + r];
              cost += f * (0.5f * (jda + bf[0] * f) + bb);
            }
          }
          cost = wave_sum<4>(cost);
          zero_f = cost > 0;
        }
        if (zero_f && !patch_ws) {
          for (int b = lane; b < nblk; b += 64) { float* bf = s_blkf + b * BLKF_STRIDE; for (int r = 0; r < 6; r++) bf[BF_F + r] = 0; }
          for (int d = lane; d < nv; d += 64) s_tmpv[d] = 0;
        }
        WSYNC();
        bool patched = false;
        if constexpr (DIAGM && NROW <= 2) {
          if (M.patch || wpre) {
            // ---- contact-patch sweep (patch_pgs.h): the blocks are regrouped into patches of up to 16 rows between the same
            //      bodies; the running acceleration lives in LDS in the scaled coordinates a^ = M^1/2 a
            patched = true;
            PatchArgs pa;
            pa.lds = lds; pa.pool = M.pool; pa.pool_floats = M.pool_floats; pa.pdesc = M.pdesc; pa.pslot = M.pslot; pa.zero = L.zero; pa.ahat = L.qacc;
            // M^-1/2 per dof (s_bias is dead since the smooth force was formed), M^1/2 qacc_smooth, M^1/2 qacc_warmstart, da = 0
            for (int d = lane; d < nv; d += 64) {
              const float sq = sqrtf(s_qLDinv[d]);
              s_bias[d] = sq; s_qacc[d] = s_asmooth[d] / sq; s_tmpv2[d] = s_ws[d] / sq; s_tmpv[d] = 0;
            }
            WSYNC();
            pa.blki = s_blki_i; pa.blkf = s_blkf; pa.J = s_J; pa.qLDinv = s_bias; pa.nblk = nblk; pa.nv = nv; pa.maxcon = M.maxcon;
            pa.row_order = M.pgs_row_order;
            if (wpre) {
              // ---- window sweep (window_pgs.h): this launch ends here; mjh_window_kernel (four envs per wavefront) runs the warm start,
              //      the sweeps in constraint-row order, mj_checkAcc and mj_Euler, and stores the state.  Handed over: the rows (J^ dense over
              //      the dofs, aref, R), M^1/2 qacc_smooth, M^1/2 qacc_warmstart, M^-1/2, qvel after the controller, the normalised qpos.
              float* wb = S.wbuf + (size_t)env * (size_t)S.wstride;
              // (the split API's hand-over has no sweep to fall back to either: rows beyond the capacity are dropped with the flag, never the whole set)
              const int nrow = window_emit(wb, M.win_nvt, M.win_maxw, s_blki_i, s_blkf, s_J, s_bias, nblk, lane, WPRE || wdefer || !M.patch);
              if ((WPRE || wdefer || !M.patch) && nefc > nrow) flags |= 2;          // (rows beyond the window kernel's capacity were dropped)
              if (nrow > 0 || wdefer) {
                for (int d = lane; d < nv; d += 64) { wb[WN_AS + d] = s_qacc[d]; wb[WN_AWS + d] = s_tmpv2[d]; wb[WN_SINV + d] = s_bias[d]; wb[WN_QVEL + d] = s_qvel[d]; }
                for (int i = lane; i < nq; i += 64) wb[WN_QPOS + i] = s_qpos[i];
                if (wdefer) {     // what mj_step1 leaves in mjData for the calls between the two halves of the step (and for a later full mjh_step2)
                  for (int i = lane; i < nq; i += 64) S.qpos[qrow + i] = s_qpos[i];
                  for (int d = lane; d < nv; d += 64) { S.qvel[vrow + d] = s_qvel[d]; S.qvel_ref[vrow + d] = s_qvref[d]; S.qfrc_applied[vrow + d] = s_applied[d]; }
                }
                // [4]: the form the window kernel sweeps this env in — 1: 32-row windows for many rows (window_kernel.h: wn_run32), 2: 64-row windows for the most (wn_run64); a function of the env's own row count
                // (split API: the counts of THIS step are what mjh_get_stats / mjh_get_field see between the two halves, as after a plain mj_step1)
                if (lane == 0 && wdefer) { S.stats[4*env] = ncon; S.stats[4*env+1] = nefc; S.stats[4*env+2] = 0; S.stats[4*env+3] |= flags & 0xff; }
                if (lane == 0) { int* wh = (int*)wb; wh[0] = nrow; wh[1] = ncon; wh[2] = nefc; wh[3] = flags; wh[4] = (S.win64 > 0 && M.win_nvt == 24 && nrow > S.win64 && nrow <= (M.win_maxw > 16 ? 64 * (WN64_NW + WN64_NT) : 64 * WN64_NW)) ? 2 : ((S.win32 > 0 && M.win_nvt == 24 && nrow > S.win32 && nrow <= 32 * WN32_NW) ? 1 : 0); wh[5] = (wdefer && nrow == 0) ? 1 : 0; }   // [5]: an env without rows that the window kernel integrates (split API)
                return;
              }
              // (more rows than the window kernel takes: this env finishes the step here, in patch form)
            }
            if constexpr (!WPRE) if (M.patch) {      // (window-only models — more than 64 contacts — never get here: their hand-over clamps)
            int swork = 0, npatch = 0;
            const int nstep = patch_build(pa, lane, flags, swork, npatch);
            if (warm) patch_warmstart(pa, lane, nstep, npatch, L.tmpv2, L.qacc, L.tmpv);
            for (int d = lane; d < nv; d += 64) s_qacc[d] += s_tmpv[d];       // a^ = M^1/2 (qacc_smooth + M^-1 J^T f)
            WSYNC();
            PROF(12);
            niter = patch_sweep(pa, lane, nstep, M.iterations, M.tolerance, 1.0f / (M.meaninertia * (float)(nv > 1 ? nv : 1)));
            cost_hint = min(((niter * swork) >> 1) + 1, 1 << 22);   // (100 sweeps x 5 steps of 40 -> 10000 -> bucket 156 of the launch order's 256)
            WSYNC();
            for (int d = lane; d < nv; d += 64) { const float qa = s_qacc[d] * s_bias[d]; s_qacc[d] = qa; s_ws[d] = qa; }
            PROF(13);
            WSYNC();
            // qfrc_constraint = M (qacc - qacc_smooth): the base rows are gone
            if (xflags & XF_FORCE) for (int d = lane; d < nv; d += 64) s_tmpv2[d] = (s_qacc[d] - s_asmooth[d]) * s_qM[dof_Madr[d]];
            }   // !WPRE
          }
        }
        if constexpr (!WPRE)
        if (!patched) {
        // (an env that the dense solver takes — dense_pgs.h: it forms the full AR on the matrix cores — needs neither the blocks'
        //  own A_c nor their row-space matrices, only the projection intervals)
        const bool dense_env = dense_pre;
        // ---- A_c = J_base M^-1 J_base^T, upper triangle (lanes = (block, base jb): row jb).  Built here, after the last user
        //      of the contact records and the velocity-stage spatial vectors: s_blkq reuses their space.
        if (!dense_env)
        for (int b = lane; b < nblk; b += 64) {     // lanes = blocks: every dof's four base entries are one 16-byte read
          const int4 hd = *(const int4*)(s_blki_i + b * BLKI_STRIDE);
          const int nb = (hd.x >> 8) & 15;
          float* A = s_blkq + b * BLKQ_STRIDE;
          ROW_TREES(hd.z, hd.w);
          const bool quad = BLK_SLOTS(hd.y) == 4; const int jo = BLK_JOFF(hd.x);
          float a00 = 0, a01 = 0, a02 = 0, a03 = 0, a11 = 0, a12 = 0, a13 = 0, a22 = 0, a23 = 0, a33 = 0;
          for (int k = 0; k < n1 + n2; k++) {
            float4 jv = make_float4(0, 0, 0, 0), bv;
            if (quad) jv = *(const float4*)(s_J + jo + 4*k); else jv.x = s_J[jo + k];
            if (DIAGM) { const float mi = s_qLDinv[k < n1 ? a1 + k : a2 + k - n1]; bv = make_float4(jv.x * mi, jv.y * mi, jv.z * mi, jv.w * mi); }
            else { bv = make_float4(0, 0, 0, 0); if (quad) bv = *(const float4*)(s_B + jo + 4*k); else bv.x = s_B[jo + k]; }
            a00 += jv.x * bv.x; a01 += jv.y * bv.x; a02 += jv.z * bv.x; a03 += jv.w * bv.x;
            a11 += jv.y * bv.y; a12 += jv.z * bv.y; a13 += jv.w * bv.y;
            a22 += jv.z * bv.z; a23 += jv.w * bv.z; a33 += jv.w * bv.w;
          }
          // upper triangle over the block's nb bases; unused rows / bases must be inert (dual-block solver)
          *(float4*)(A) = make_float4(nb > 0 ? a00 : 0.0f, nb > 1 ? a01 : 0.0f, nb > 2 ? a02 : 0.0f, nb > 3 ? a03 : 0.0f);
          *(float4*)(A + 4) = make_float4(0.0f, nb > 1 ? a11 : 0.0f, nb > 2 ? a12 : 0.0f, nb > 3 ? a13 : 0.0f);
          *(float4*)(A + 8) = make_float4(0.0f, 0.0f, nb > 2 ? a22 : 0.0f, nb > 3 ? a23 : 0.0f);
          *(float4*)(A + 12) = make_float4(0.0f, 0.0f, 0.0f, nb > 3 ? a33 : 0.0f);
        }
        WSYNC();
        // ---- row-space matrix of every block: AR = E A_c E^T + R I (rows e_r = e_n +- e_k), laid out for pgs_rows().
        //      Done here, after the last user of the velocity-stage spatial vectors: the X extension aliases them.
        for (int b = lane; b < nblk; b += 64) {
          const int* hd = s_blki_i + b * BLKI_STRIDE;
          float* bf = s_blkf + b * BLKF_STRIDE;
          if (dense_env) {
            const int clamp = (hd[0] >> 12) & 3;
            bf[BF_LO] = clamp == 0 ? -3.0e38f : (clamp == 2 ? -bf[1] : 0.0f);
            bf[BF_LO + 1] = clamp == 2 ? bf[1] : 3.0e38f;
            continue;
          }
          float* Q = s_blkq + b * BLKQ_STRIDE;
          const int kind = hd[0] & 15, nr = (hd[0] >> 4) & 15;
          float Ac[4][4];
    #pragma unroll
          for (int i = 0; i < 4; i++)
    #pragma unroll
            for (int j = i; j < 4; j++) { Ac[i][j] = Q[4*i + j]; Ac[j][i] = Ac[i][j]; }
          const float R = bf[0];
          float q[SOLQ_N], x[SOLX_N];
    #pragma unroll
          for (int i = 0; i < SOLQ_N; i++) q[i] = 0;
    #pragma unroll
          for (int i = 0; i < SOLX_N; i++) x[i] = 0;
          if (kind == BK_SINGLE) { const float AR = Ac[0][0] + R; q[0] = 1.0f / AR; q[4] = 0.5f * AR; }
          else {
    #pragma unroll
            for (int r = 0; r < 6; r++) {
              if (r >= nr) continue;
              const int kr = 1 + (r >> 1); const float cr = (r & 1) ? -1.0f : 1.0f;
              const float AR = Ac[0][0] + 2.0f * cr * Ac[0][kr] + Ac[kr][kr] + R;
              const float inv = 1.0f / AR, half = 0.5f * AR;
              if (r < 4) { q[r] = inv; q[4 + r] = half; } else { q[10 + r] = inv; x[r - 4] = half; }
    #pragma unroll
              for (int t = r + 1; t < 6; t++) {
                if (t >= nr) continue;
                const int kt = 1 + (t >> 1); const float ct = (t & 1) ? -1.0f : 1.0f;
                const float v = Ac[0][0] + ct * Ac[0][kt] + cr * Ac[kr][0] + cr * ct * Ac[kr][kt];
                const int sl = ar_off_slot(r, t);
                if (sl < 16) q[sl] = v; else x[sl - 16] = v;
              }
            }
          }
    #pragma unroll
          for (int i = 0; i < SOLQ_N; i++) Q[i] = q[i];
          {   // projection interval of the block's rows: equality (-inf, inf), friction loss [-fl, fl], everything else [0, inf)
            const int clamp = (hd[0] >> 12) & 3;
            bf[BF_LO] = clamp == 0 ? -3.0e38f : (clamp == 2 ? -bf[1] : 0.0f);
            bf[BF_LO + 1] = clamp == 2 ? bf[1] : 3.0e38f;
          }
          if (M.has_dim4) {
            float* X = s_ext + b * SOLX_N;
    #pragma unroll
            for (int i = 0; i < SOLX_N; i++) X[i] = x[i];
          }
        }
        WSYNC();
        PROF(12);
        if (pre) {   // initial acceleration and 1/M_dd for the stand-alone solver; the blocks are in the pools already
          for (int d = lane; d < nv; d += 64) { gs[L.g_a0 + d] = s_asmooth[d] + s_tmpv[d]; gs[L.g_minv + d] = s_qLDinv[d]; }
          if (dense_pre) {
            for (int i = lane; i < M.nM; i += 64) { gs[L.g_qLD + i] = s_qLD[i]; ((int*)(gs + L.g_anc))[i] = s_anc_i[i]; }     // (the dense solve finishes qacc with L^-1)
            // mjh_dense_build_kernel takes J a0 = J qacc_smooth + J da from the warm start's base-row products (pools phi, bv)
            if (!warm) base_dots(s_asmooth, s_phi);
            if (zero_f) for (int i = lane; i < 4 * nblk; i += 64) s_bv[i] = 0;
          }
          return;
        }
        // ---- PGS (mj_solPGS), matrix-free, one block at a time.  lanes = dofs, running acceleration `a` in a
        //      register.  Per block: u = J_base.a (nbase wave reductions, interleaved); its 2(dim-1) pyramid rows
        //      are then updated with uniform scalar math in contact space:
        //        res_r = e_r.u - aref_r + R f_r ;  f_r <- max(0, f_r - res_r/AR_rr) ;  u += A_c e_r delta
        //      and `a` is touched once per block:  a += B_base^T dphi.
        const int d0 = (NROW <= 2) ? (lane & 31) : lane;   // dual mode: both halves carry a copy of the dof vector
        float a = (NROW != 8 && d0 < nv) ? s_asmooth[d0] + s_tmpv[d0] : 0.0f;
        const float minv0 = (NROW != 8 && d0 < nv) ? s_qLDinv[d0] : 0.0f;
        const int r6 = d0 % 6, dbase6 = d0 < nv ? d0 - r6 : -1;   // DIAGM: dof inside its free body / first dof of that body
        const float scale = 1.0f / (M.meaninertia * (float)(nv > 1 ? nv : 1));
        // sweep mode (dual / single-block sweeps below; the many-body sweep keeps its own): main PGS, then noslip
        bool ns = false; int itmax = M.iterations, nmain = 0; float tol = M.tolerance;
        const int nmode = (EXTRA && M.noslip_iterations > 0) ? 2 : 1;
        const float4* blkf4 = (const float4*)s_blkf;
        const float4* blkq4 = (const float4*)s_blkq;
        const int4* blki4 = (const int4*)s_blki_i;
        const bool has_dim4 = M.has_dim4 != 0;   // condim-4 contacts present: blocks carry the X extension
        if constexpr (NROW == 8) {
          // ======== nv > 64: the running acceleration lives in LDS (s_qacc); lanes = the compact dofs of ONE block
          //          (at most 64: rowW), gathered before and scattered after the block's update; the block's operands come
          //          from the global pools and are prefetched one block ahead.
          for (int d = lane; d < nv; d += 64) s_qacc[d] = s_asmooth[d] + s_tmpv[d];
          WSYNC();
          {
            ManyCtx mc;
            mc.J = s_J; mc.B = s_B; mc.blkf = s_blkf; mc.blkq = s_blkq; mc.ext = s_ext; mc.blki = s_blki_i;
            // visiting order and group starts: LDS copies in the (dead) position-stage arrays when they fit
            mc.order = s_order_i; mc.gstart = s_sched_i; mc.ngrp = ngrp; mc.order_packed = false;
            if (nblk + ngrp + 2 <= M.k1_floats) {
              int* ol = (int*)(lds + M.scratch_off); int* gl = ol + nblk;
              for (int i = lane; i < nblk; i += 64) { const int b = s_order_i[i]; const int4 hd = ((const int4*)s_blki_i)[b]; ol[i] = MJH_ORDER_WORD(b, hd); }
              if (nblk > 64) for (int i = lane; i <= ngrp; i += 64) gl[i] = s_sched_i[i];
              WSYNC();
              mc.order = ol; mc.gstart = gl; mc.order_packed = true;
            }
            mc.qacc = s_qacc; mc.qLDinv = s_qLDinv;
            mc.nblk = nblk; mc.nfixblk = __builtin_amdgcn_readfirstlane(nfixblk); mc.rowW = rowW; mc.iterations = M.iterations;
            mc.has_dim4 = has_dim4; mc.scale = scale; mc.tolerance = M.tolerance;
            mc.noslip_iterations = M.noslip_iterations; mc.noslip_tolerance = M.noslip_tolerance;
            mc.nwave = 1; mc.wid = 0; mc.red = nullptr; mc.a4 = nullptr; mc.m4 = nullptr;
            {   // the pools are in the env's global slice here too: same buffer-descriptor fetch as mjh_solve_kernel
              const unsigned long long ga = (unsigned long long)gs;
              const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ga), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ga >> 32));
              const long long nb = S.gstride * 4;
              mc.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)(nb > 0x7ffffff0ll ? 0x7ffffff0ll : nb), 0x00020000);
              mc.oJ = 4 * (-1 - L.J); mc.oB = 4 * (-1 - L.B); mc.oblkf = 4 * (-1 - L.blkf); mc.oblkq = 4 * (-1 - L.blkq); mc.oext = 4 * (-1 - L.ext); mc.oblki = 4 * (-1 - L.blki);
            }
            niter = pgs_many_body<DIAGM, EXTRA, true>(mc, lane);
          }
          WSYNC();
          for (int d = lane; d < nv; d += 64) s_ws[d] = s_qacc[d];
        } else if constexpr (NROW <= 2) {
          // ======== dual-block sweep: half 0 solves block p, half 1 its independent partner q of the schedule
          const int hh = lane >> 5, lq = lane & 3;
          // software pipeline over the (cyclic) schedule, every stage consuming LDS data requested one step earlier:
          //   S: pair of step t+3  ->  H: block header of step t+2  ->  L: operands of step t+1  ->  solve step t
          struct DHd { int4 hd; int b; float act; };
          struct DOp { int hx, b, kind; float act, R; float4 J, B, r0, r1, r2, A0, A1, A2, A3, X0, X1, X2; };
          int gS = 0;
          auto nextS = [&]() __attribute__((always_inline)) {
            const int2 pq = *(const int2*)(s_sched_i + 2*gS);
            gS = gS + 1 == ngrp ? 0 : gS + 1;
            return pq;
          };
          auto hdrOf = [&](const int2 pq) __attribute__((always_inline)) {
            DHd h;
            const int bsel = hh ? pq.y : pq.x;
            const bool act = bsel >= 0;
            h.b = act ? bsel : pq.x; h.act = act ? 1.0f : 0.0f;
            h.hd = blki4[h.b];
            return h;
          };
          auto loadOp = [&](const DHd& h) __attribute__((always_inline)) {
            DOp op;
            KEEP4(h.hd);
            ROW_TREES(h.hd.z, h.hd.w);
            int o; bool on;                                    // lanes outside the block read the zero slot
            if (DIAGM) { const bool on2 = dbase6 == a2; o = r6 + (on2 ? 6 : 0); on = (dbase6 == a1 || on2) && h.act > 0.0f; }   // free bodies: 6 dofs each
            else { o = row_off(d0, a1, n1, a2, n2); on = o >= 0 && h.act > 0.0f; }
            const int jo = BLK_JOFF(h.hd.x), b = h.b;
            const bool quad = DIAGM || BLK_SLOTS(h.hd.y) == 4; // DIAGM models have contact blocks only (engine.hip)
            if (quad) op.J = *(const float4*)(on ? s_J + jo + 4*o : s_zero);
            else op.J = make_float4(on ? s_J[jo + o] : 0.0f, 0, 0, 0);
            if (!DIAGM) {
              if (quad) op.B = *(const float4*)(on ? s_B + jo + 4*o : s_zero);
              else op.B = make_float4(on ? s_B[jo + o] : 0.0f, 0, 0, 0);
            }
            op.R = s_blkf[BLKF_STRIDE * b]; op.r0 = blkf4[4*b+1]; op.r1 = blkf4[4*b+2]; op.r2 = blkf4[4*b+3];
            op.A0 = blkq4[4*b]; op.A1 = blkq4[4*b+1]; op.A2 = blkq4[4*b+2]; op.A3 = blkq4[4*b+3];
            op.hx = h.act > 0.0f ? h.hd.x : 0; op.b = b; op.act = h.act;
            // the unrolled row count follows the larger of the two blocks; the smaller one's extra rows are inert
            const int k0 = __builtin_amdgcn_readlane(op.hx & 15, 0), k1 = __builtin_amdgcn_readlane(op.hx & 15, 32);
            op.kind = k0 > k1 ? k0 : k1;
            if (has_dim4 && op.kind == BK_PYR4) { const float4* x4 = (const float4*)(s_ext + b * SOLX_N); op.X0 = x4[0]; op.X1 = x4[1]; op.X2 = x4[2]; }
            return op;
          };
          ImpQ iq = imp_quantum(scale, tol);     // (fixed-point cost decrease: the total must not depend on which half carried which block)
          auto processD = [&](DOp& op, int& improvement) __attribute__((always_inline)) {
            const int kind = op.kind;
            KEEP4(op.J); KEEP4(op.r0); KEEP4(op.r1); KEEP4(op.r2); KEEP4(op.A0); KEEP4(op.A1); KEEP4(op.A2); KEEP4(op.A3);
            if (!DIAGM) KEEP4(op.B);
            if (has_dim4 && kind == BK_PYR4) { KEEP4(op.X0); KEEP4(op.X1); KEEP4(op.X2); }
            float f[6] = {op.r1.x, op.r1.y, op.r1.z, op.r1.w, op.r2.x, op.r2.y};
            const float lo = op.r2.z, hi = op.r2.w;
            const float ab[4] = {op.r0.x, op.r0.y, op.r0.z, op.r0.w};
            const float Q[16] = {op.A0.x, op.A0.y, op.A0.z, op.A0.w, op.A1.x, op.A1.y, op.A1.z, op.A1.w,
                                 op.A2.x, op.A2.y, op.A2.z, op.A2.w, op.A3.x, op.A3.y, op.A3.z, op.A3.w};
            const float X[12] = {op.X0.x, op.X0.y, op.X0.z, op.X0.w, op.X1.x, op.X1.y, op.X1.z, op.X1.w, op.X2.x, op.X2.y, op.X2.z, op.X2.w};
            const float Jd[4] = {op.J.x, op.J.y, op.J.z, op.J.w}, Bd[4] = {op.B.x, op.B.y, op.B.z, op.B.w};
            const float* Bp = DIAGM ? Jd : Bd;                 // diagonal M: B = J / M_dd, applied as one scale of da
            const float bs = DIAGM ? minv0 : 1.0f;
            float imp;
            if (kind == BK_PYR4) imp = pgs_dual<4, 6, EXTRA>(ns, op.R, lo, hi, ab, f, Q, X, Jd, Bp, bs, lq, a);
            else if (kind == BK_PYR3) imp = pgs_dual<3, 4, EXTRA>(ns, op.R, lo, hi, ab, f, Q, X, Jd, Bp, bs, lq, a);
            else imp = pgs_dual<1, 1, EXTRA>(ns, op.R, lo, hi, ab, f, Q, X, Jd, Bp, bs, lq, a);
            improvement += imp_fixed(op.act * imp, iq.qs);
            if (d0 == 0 && op.act > 0.0f) {
              float* bf = s_blkf + op.b * BLKF_STRIDE + BF_F;
              *(float4*)(bf) = make_float4(f[0], f[1], f[2], f[3]);
              *(float2*)(bf + 4) = make_float2(f[4], f[5]);
            }
          };
          for (int mode = 0; mode < nmode; mode++) {   // main sweeps, then (EXTRA) the noslip sweeps over the same schedule
          if (mode == 1) { ns = true; nmain = niter; niter = 0; itmax = M.noslip_iterations; tol = M.noslip_tolerance; gS = 0; iq = imp_quantum(scale, tol); WSYNC(); }
          if (ngrp == 1) {
            // a single group: its operands (the forces) change under the prefetch, so no pipeline
            for (int it = 0; it < itmax; it++) {
              int impl = 0;
              DOp op = loadOp(hdrOf(*(const int2*)s_sched_i));
              processD(op, impl);
              niter = it + 1;
              if (__builtin_amdgcn_readlane(impl, 0) + __builtin_amdgcn_readlane(impl, 32) < iq.thr) break;
            }
          } else {
            int2 pqN = nextS();                      // pair of step 0
            DHd hN = hdrOf(pqN); pqN = nextS();      // header of step 0, pair of step 1
            DOp opA = loadOp(hN), opB;               // operands of step 0
            hN = hdrOf(pqN); pqN = nextS();          // header of step 1, pair of step 2
            for (int it = 0; it < itmax; it++) {
              int impl = 0;
              for (int g = 0; g < ngrp; g += 2) {
                opB = loadOp(hN); hN = hdrOf(pqN); pqN = nextS();
                processD(opA, impl);
                if (g + 1 < ngrp) {
                  opA = loadOp(hN); hN = hdrOf(pqN); pqN = nextS();
                  processD(opB, impl);
                } else opA = opB;                    // odd group count: step 0 of the next sweep was loaded into B
              }
              niter = it + 1;
              if (__builtin_amdgcn_readlane(impl, 0) + __builtin_amdgcn_readlane(impl, 32) < iq.thr) break;
            }
          }
          }   // sweep mode
          niter += nmain;
        } else {
        // operands of one block: 1 header + 1 (2) Jacobian + 8 parameter ds_read_b128 per lane
        struct BlkOp { int hx; float4 J, B, P, r0, r1, r2, A0, A1, A2, A3, X0, X1, X2; };
        auto fetch = [&](int b) __attribute__((always_inline)) {
          BlkOp op;
          const int4 hd = blki4[b];
          ROW_TREES(hd.z, hd.w);
          int o; bool on;                                    // lanes outside the block read the zero slot
          if (DIAGM) { const bool on2 = dbase6 == a2; o = r6 + (on2 ? 6 : 0); on = dbase6 == a1 || on2; }   // free bodies: 6 dofs each
          else { o = row_off(d0, a1, n1, a2, n2); on = o >= 0; }
          const int jo = BLK_JOFF(hd.x);
          const bool quad = DIAGM || BLK_SLOTS(hd.y) == 4;   // DIAGM models have contact blocks only (engine.hip)
          if (quad) op.J = *(const float4*)(on ? s_J + jo + 4*o : s_zero);
          else op.J = make_float4(on ? s_J[jo + o] : 0.0f, 0, 0, 0);
          if (!DIAGM) {
            if (quad) op.B = *(const float4*)(on ? s_B + jo + 4*o : s_zero);
            else op.B = make_float4(on ? s_B[jo + o] : 0.0f, 0, 0, 0);
          }
          op.P = blkf4[4*b]; op.r0 = blkf4[4*b+1]; op.r1 = blkf4[4*b+2]; op.r2 = blkf4[4*b+3];
          op.A0 = blkq4[4*b]; op.A1 = blkq4[4*b+1]; op.A2 = blkq4[4*b+2]; op.A3 = blkq4[4*b+3];
          if (has_dim4) { const float4* x4 = (const float4*)(s_ext + b * SOLX_N); op.X0 = x4[0]; op.X1 = x4[1]; op.X2 = x4[2]; }
          op.hx = hd.x;
          return op;
        };
        auto process = [&](BlkOp& op, int b, float& improvement) __attribute__((always_inline)) {
          const int kind = __builtin_amdgcn_readfirstlane(op.hx & 15);
          float f[6] = {op.r1.x, op.r1.y, op.r1.z, op.r1.w, op.r2.x, op.r2.y};
          const float aref[4] = {op.r0.x, op.r0.y, op.r0.z, op.r0.w};
          const float Q[16] = {op.A0.x, op.A0.y, op.A0.z, op.A0.w, op.A1.x, op.A1.y, op.A1.z, op.A1.w,
                               op.A2.x, op.A2.y, op.A2.z, op.A2.w, op.A3.x, op.A3.y, op.A3.z, op.A3.w};
          const float X[12] = {op.X0.x, op.X0.y, op.X0.z, op.X0.w, op.X1.x, op.X1.y, op.X1.z, op.X1.w, op.X2.x, op.X2.y, op.X2.z, op.X2.w};
          const float Jd[4] = {op.J.x, op.J.y, op.J.z, op.J.w}, Bd[4] = {op.B.x, op.B.y, op.B.z, op.B.w};
          const float* Bp = DIAGM ? Jd : Bd;
          const float bs = DIAGM ? minv0 : 1.0f;
          const float R = op.P.x, lo = op.r2.z, hi = op.r2.w;
          if (kind == BK_PYR4) pgs_block<4, 6, NROW, EXTRA>(ns, R, lo, hi, aref, f, Q, X, Jd, Bp, bs, a, improvement);
          else if (kind == BK_PYR3) pgs_block<3, 4, NROW, EXTRA>(ns, R, lo, hi, aref, f, Q, X, Jd, Bp, bs, a, improvement);
          else pgs_block<1, 1, NROW, EXTRA>(ns, R, lo, hi, aref, f, Q, X, Jd, Bp, bs, a, improvement);
          if (lane == 0) {
            float* bf = s_blkf + b * BLKF_STRIDE + BF_F;
            *(float4*)(bf) = make_float4(f[0], f[1], f[2], f[3]);
            *(float2*)(bf + 4) = make_float2(f[4], f[5]);
          }
        };
        for (int mode = 0; mode < nmode; mode++) {   // main sweeps, then (EXTRA) the noslip sweeps in the same visiting order
        if (mode == 1) { ns = true; nmain = niter; niter = 0; itmax = M.noslip_iterations; tol = M.noslip_tolerance; WSYNC(); }
        for (int it = 0; it < itmax; it++) {
          float improvement = 0;
          // two operand buffers in ping-pong: the next block's LDS reads are in flight while this one is solved
          BlkOp opA = fetch(s_order_i[0]), opB;
          for (int k = 0; k < nblk; k += 2) {
            const int b0 = s_order_i[k], b1 = s_order_i[k + 1 < nblk ? k + 1 : k], b2 = s_order_i[k + 2 < nblk ? k + 2 : k];
            opB = fetch(b1);
            process(opA, b0, improvement);
            if (k + 1 < nblk) {
              opA = fetch(b2);
              process(opB, b1, improvement);
            }
          }
          niter = it + 1;
          if (improvement * scale < tol) break;
          WSYNC();   // the next sweep re-reads the forces from LDS
        }
        }   // sweep mode
        niter += nmain;
        }
        WSYNC();
        if (NROW != 8 && d0 < nv && lane < 64 / (NROW <= 2 ? 2 : 1)) { s_qacc[d0] = a; s_ws[d0] = a; }
        PROF(13);
        WSYNC();
        // qfrc_constraint = J^T f (only needed for export: the implicit-damping integrator works from qacc)
        if (xflags & XF_FORCE) { phi_from_forces(); accum_T(false, s_phi, s_tmpv2); }
        }   // !patched
      }
      }   // !post
      if (xflags & XF_FORCE) {
        const size_t e = (size_t)xrow * M.nvp;
        for (int d = lane; d < nv; d += 64) { if (S.x_smooth) S.x_smooth[e + d] = s_asmooth[d]; if (S.x_constraint) S.x_constraint[e + d] = s_tmpv2[d]; }
      }
      if (EXTRA && M.nsensor > 0 && S.sensordata && !post) {
        // ---- mj_sensorAcc -> mj_rnePostConstraint (force / torque sensors; consumer: mj_ros.cpp:1933-1966).  External spatial
        //      force per body about its tree root's COM (xfrc_applied, contact forces, connect / weld forces), then the RNE
        //      forward pass with the SOLVED qacc, cfrc_int = cinert cacc + cvel x* (cinert cvel) - cfrc_ext summed over each
        //      body's subtree, read at the site in the site frame.  Lanes = bodies gather their own forces (deterministic).
        WSYNC();
        const float* xf = S.xfrc_applied ? S.xfrc_applied + (size_t)env * S.xfrc_stride : nullptr;
        for (int b = lane; b < nbody; b += 64) {
          float F[6] = {0, 0, 0, 0, 0, 0};
          if (b > 0) {
            const float* com = s_com + 3*body_rootid[b];
            auto add = [&](const float* point, const float* force, const float* torque, const float sg) __attribute__((always_inline)) {
              const float off[3] = {point[0] - com[0], point[1] - com[1], point[2] - com[2]};
              float cr[3]; cross3(cr, off, force);
              F[0] += sg * (torque[0] + cr[0]); F[1] += sg * (torque[1] + cr[1]); F[2] += sg * (torque[2] + cr[2]);
              F[3] += sg * force[0]; F[4] += sg * force[1]; F[5] += sg * force[2];
            };
            if (xf) { const float f[3] = {xf[6*b], xf[6*b+1], xf[6*b+2]}, t[3] = {xf[6*b+3], xf[6*b+4], xf[6*b+5]}; add(s_xipos + 3*b, f, t, 1.0f); }
            for (int k = 0; k < nblk; k++) {
              const int* hd = s_blki_i + k * BLKI_STRIDE;
              const int id = hd[1] & 0xffffff, rtype = (hd[1] >> 24) & 15, kind = hd[0] & 15;
              const float* bf = s_blkf + k * BLKF_STRIDE;
              if (rtype == RT_CONTACT) {
                const float* c = s_con + id * CON_STRIDE;
                const int g1 = CON_G1(c), g2 = CON_G2(c);
                const int b1 = geom_bodyid[g1], b2 = geom_bodyid[g2];
                if (b1 != b && b2 != b) continue;
                // contact-frame force on geom2's body: bases carry their friction coefficient, so the signed row sums of a
                // base ARE its force component (phi_from_forces); torsional base 3 = torque about the normal
                float ph[4] = {0, 0, 0, 0};
                const int nr = (hd[0] >> 4) & 15;
                for (int r = 0; r < nr; r++) {
                  const float f = bf[BF_F + r];
                  ph[0] += f;
                  if (kind != BK_SINGLE) { const int kk = 1 + (r >> 1); ph[kk] += (r & 1) ? -f : f; }
                }
                const float mu1 = fmaxf(geom_friction[3*g1], geom_friction[3*g2]), mu3 = fmaxf(geom_friction[3*g1+1], geom_friction[3*g2+1]);
                const float cf[3] = {ph[0], mu1 * ph[1], mu1 * ph[2]}, tors = mu3 * ph[3];
                float Fw[3], Tw[3];
#pragma unroll
                for (int q = 0; q < 3; q++) { Fw[q] = c[4+q] * cf[0] + c[7+q] * cf[1] + c[10+q] * cf[2]; Tw[q] = c[4+q] * tors; }
                if (b2 == b) add(c + 1, Fw, Tw, 1.0f);
                if (b1 == b) add(c + 1, Fw, Tw, -1.0f);
              } else if (rtype == RT_WELD) {
                const int e = id & 0xffff, r = id >> 16;
                const int b1 = eq_obj1id[e], b2 = eq_obj2id[e];
                if (b1 != b && b2 != b) continue;
                const bool weld = M.I[M.o_eq_type + e] == MJH_EQ_WELD;
                const float* dat = eq_data + 11*e;
                const float f = bf[BF_F];
                const float z3[3] = {0, 0, 0};
                const bool first = b1 == b;
                float pw[3]; rotvec(pw, s_xmat + 9*b, weld ? (first ? dat + 3 : dat) : (first ? dat : dat + 3));
                const float pt[3] = {s_xpos[3*b] + pw[0], s_xpos[3*b+1] + pw[1], s_xpos[3*b+2] + pw[2]};
                if (r < 3) { float fv[3] = {0, 0, 0}; fv[r] = f; add(pt, fv, z3, first ? 1.0f : -1.0f); }
                else {
                  const float q1i[4] = {s_xquat[4*b1], -s_xquat[4*b1+1], -s_xquat[4*b1+2], -s_xquat[4*b1+3]}, rel[4] = {dat[6], dat[7], dat[8], dat[9]};
                  float q2r[4]; mulquat(q2r, s_xquat + 4*b2, rel);
                  float T[3];
#pragma unroll
                  for (int a = 0; a < 3; a++) {
                    const float w[4] = {0.0f, a == 0 ? 1.0f : 0.0f, a == 1 ? 1.0f : 0.0f, a == 2 ? 1.0f : 0.0f};
                    float u[4], vv[4]; mulquat(u, q1i, w); mulquat(vv, u, q2r);
                    T[a] = 0.5f * dat[10] * vv[1 + (r - 3)] * f;
                  }
                  add(pt, z3, T, first ? -1.0f : 1.0f);
                }
              }
            }
          }
#pragma unroll
          for (int q = 0; q < 6; q++) s_fext[6*b+q] = F[q];
        }
        WSYNC();
        // forward pass (as vel_levels with flg_acc), then cfrc_int and its subtree sums
        if (lane == 0) { for (int k = 0; k < 6; k++) { s_cvel[k] = 0; s_cacc[k] = (k >= 3) ? -grav[k-3] : 0.0f; s_cfrc[k] = 0; } }
        WSYNC();
        for (int lev = 1; lev <= M.maxlevel; lev++) {
          for (int b = lane; b < nbody; b += 64) {
            if (body_level[b] != lev) continue;
            const int p = body_parentid[b];
            float cv[6], ca[6];
#pragma unroll
            for (int q = 0; q < 6; q++) { cv[q] = s_cvel[6*p+q]; ca[q] = s_cacc[6*p+q]; }
            int bda = body_dofadr[b];
            for (int j = 0; j < body_jntnum[b]; j++) {
              const int jt = jnt_type[body_jntadr[b] + j];
              if (jt == MJH_JNT_FREE) {
                for (int k = 0; k < 3; k++)
#pragma unroll
                  for (int q = 0; q < 6; q++) { s_cdofdot[6*(bda+k)+q] = 0; cv[q] += s_cdof[6*(bda+k)+q] * s_qvel[bda+k]; }
                bda += 3;
              }
              const int nd = (jt == MJH_JNT_FREE || jt == MJH_JNT_BALL) ? 3 : 1;
              float cvn[6];
#pragma unroll
              for (int q = 0; q < 6; q++) cvn[q] = cv[q];
              for (int k = 0; k < nd; k++) {
                float cd[6], cdd[6];
#pragma unroll
                for (int q = 0; q < 6; q++) cd[q] = s_cdof[6*(bda+k)+q];
                cross_motion(cdd, cv, cd);
                const float v = s_qvel[bda+k];
#pragma unroll
                for (int q = 0; q < 6; q++) { s_cdofdot[6*(bda+k)+q] = cdd[q]; cvn[q] += cd[q] * v; }
              }
#pragma unroll
              for (int q = 0; q < 6; q++) cv[q] = cvn[q];
              bda += nd;
            }
            for (int d = body_dofadr[b]; d < body_dofadr[b] + body_dofnum[b]; d++) {
              const float v = s_qvel[d], a = s_qacc[d];
#pragma unroll
              for (int q = 0; q < 6; q++) ca[q] += s_cdofdot[6*d+q] * v + s_cdof[6*d+q] * a;
            }
            float ci[10], f[6], t[6], t1[6];
#pragma unroll
            for (int k = 0; k < 10; k++) ci[k] = s_cinert[10*b+k];
            mul_inert_vec(f, ci, ca); mul_inert_vec(t, ci, cv); cross_force(t1, cv, t);
#pragma unroll
            for (int q = 0; q < 6; q++) { s_cvel[6*b+q] = cv[q]; s_cacc[6*b+q] = ca[q]; s_cfrc[6*b+q] = f[q] + t1[q] - s_fext[6*b+q]; }
          }
          WSYNC();
        }
        for (int si = lane; si < M.nsensor; si += 64) {
          const int site = M.I[M.o_sensor_objid + si], b = M.I[M.o_site_bodyid + site];
          float F[6] = {0, 0, 0, 0, 0, 0};
          for (int c = b; c < b + body_subtreesize[b]; c++)
#pragma unroll
            for (int q = 0; q < 6; q++) F[q] += s_cfrc[6*c+q];
          const float* com = s_com + 3*body_rootid[b];
          const float* sp = s_site + 12*site;
          const float off[3] = {sp[0] - com[0], sp[1] - com[1], sp[2] - com[2]};
          float cr[3]; cross3(cr, off, F + 3);
          const float tq[3] = {F[0] - cr[0], F[1] - cr[1], F[2] - cr[2]};
          const float* src = M.I[M.o_sensor_type + si] == MJH_SENS_FORCE ? F + 3 : tq;
          float* out = S.sensordata + ((size_t)env * M.nsensor + si) * 3;
#pragma unroll
          for (int k = 0; k < 3; k++) out[k] = sp[3+k] * src[0] + sp[6+k] * src[1] + sp[9+k] * src[2];     // R_site^T v
        }
        WSYNC();
      }
      if (ph & PH_STEP2) {
        // ---- mj_checkAcc
        bool badv = false;
        for (int i = lane; i < nv; i += 64) { float y = s_qacc[i]; badv |= !(y == y) || fabsf(y) > MJ_MAXVAL; }
        if (wave_any(badv)) {
          for (int i = lane; i < nq; i += 64) s_qpos[i] = S.initial_qpos[qrow + i];
          for (int i = lane; i < nv; i += 64) { s_qvel[i] = 0; s_qacc[i] = 0; s_ws[i] = 0; s_applied[i] = 0; s_tmpv2[i] = 0; s_smooth[i] = 0; }
          flags |= 4;
          WSYNC();
        }
        // ---- semi-implicit Euler with implicit joint damping (mj_Euler)
        float* qint = s_qacc;
        if (M.has_damping && !(M.disableflags & MJH_DSBL_EULERDAMP)) {
          // (M + h D) qacc' = qfrc_smooth + qfrc_constraint = M qacc   <=>   qacc' = qacc - h (M + h D)^-1 D qacc: the same update
          // without forming the constraint force (J^T f over all blocks), and the solve only carries the small correction
          for (int i = lane; i < M.nM; i += 64) s_qLD[i] = qM_ro[i];
          for (int d = lane; d < nv; d += 64) s_tmpv[d] = dof_damping[d] * s_qacc[d];
          WSYNC();
          for (int d = lane; d < nv; d += 64) s_qLD[s_dofMadr_i[d]] += h * dof_damping[d];
          WSYNC();
          for (int t = 0; t < M.ntree; t++) if (tree_dofnum[t] >= MJH_WAVE_TREE_MIN) {
            factor_tree_wave(s_qLD, s_qLDinv, s_anc_i, s_dofMadr_i, tree_dofadr[t], tree_dofnum[t], M.nM, nv, lane);
            if (!level_solves) solve_tree_wave(s_tmpv, s_qLD, s_qLDinv, s_anc_i, s_dofMadr_i, tree_dofadr[t], tree_dofnum[t], M.nM, nv, lane);
          }
          if (short_rows) factor_trees_short(s_qLD, s_qLDinv, s_anc_i, s_dofMadr_i, tree_dofadr, tree_dofnum, M.ntree, M.nM, nv, lane);
          else for (int t = lane; t < M.ntree; t += 64) if (tree_dofnum[t] < MJH_WAVE_TREE_MIN) {
            factor_tree(s_qLD, s_qLDinv, s_dofpar_i, s_dofMadr_i, tree_dofadr[t], tree_dofnum[t]);
            if (!level_solves) solve_tree(s_tmpv, s_qLD, s_qLDinv, s_dofpar_i, s_dofMadr_i, tree_dofadr[t], tree_dofnum[t]);
          }
          WSYNC();
          if (level_solves) solve_trees_levels(s_tmpv, s_qLD, s_qLDinv, s_anc_i, s_dofMadr_i, nv, M.nM, lane);
          WSYNC();
          for (int d = lane; d < nv; d += 64) s_tmpv[d] = s_qacc[d] - h * s_tmpv[d];
          WSYNC();
          qint = s_tmpv;
        }
        for (int d = lane; d < nv; d += 64) {
          const int bd = dof_bodyid[d];
          const unsigned rb = (unsigned)(bd - sbase);
          const bool parked = rb < 32u && ((slotmask >> rb) & 1u);      // inactive slot: frozen in place
          s_qvel[d] = parked ? 0.0f : s_qvel[d] + h * qint[d];
          if (parked) { s_qacc[d] = 0; s_ws[d] = 0; }
        }
        WSYNC();
        for (int j = lane; j < njnt; j += 64) {
          const int qa = jnt_qposadr[j], da = jnt_dofadr[j], jt = jnt_type[j];
          if (jt == MJH_JNT_FREE) {
            s_qpos[qa] += h * s_qvel[da]; s_qpos[qa+1] += h * s_qvel[da+1]; s_qpos[qa+2] += h * s_qvel[da+2];
            float q[4] = {s_qpos[qa+3], s_qpos[qa+4], s_qpos[qa+5], s_qpos[qa+6]}, w[3] = {s_qvel[da+3], s_qvel[da+4], s_qvel[da+5]};
            quat_integrate(q, w, h);
            s_qpos[qa+3] = q[0]; s_qpos[qa+4] = q[1]; s_qpos[qa+5] = q[2]; s_qpos[qa+6] = q[3];
          } else if (jt == MJH_JNT_BALL) {
            float q[4] = {s_qpos[qa], s_qpos[qa+1], s_qpos[qa+2], s_qpos[qa+3]}, w[3] = {s_qvel[da], s_qvel[da+1], s_qvel[da+2]};
            quat_integrate(q, w, h);
            s_qpos[qa] = q[0]; s_qpos[qa+1] = q[1]; s_qpos[qa+2] = q[2]; s_qpos[qa+3] = q[3];
          } else s_qpos[qa] += h * s_qvel[da];
        }
        time += M.timestep_d;
        WSYNC();
        // ---- odom velocities (MjSim::set_odom_vels, mj_sim.cpp:1079-1153)
        if (lane == 0 && odom[9]) {
          const float* v = S.odom_vel + (size_t)env * 6;
          const float ax = odom[6] >= 0 ? s_qpos[odom[6]] : 0.0f, ay = odom[7] >= 0 ? s_qpos[odom[7]] : 0.0f, az = odom[8] >= 0 ? s_qpos[odom[8]] : 0.0f;
          const float sx = sinf(ax), cx = cosf(ax), sy = sinf(ay), cy = cosf(ay), sz = sinf(az), cz = cosf(az);
          if (odom[0] >= 0) s_qvel[odom[0]] = v[0]*cy*cz + v[1]*(sx*sy*cz - cx*sz) + v[2]*(cx*sy*cz + sx*sz);
          if (odom[1] >= 0) s_qvel[odom[1]] = v[0]*cy*sz + v[1]*(sx*sy*sz + cx*cz) + v[2]*(cx*sy*sz - sx*cz);
          if (odom[2] >= 0) s_qvel[odom[2]] = -v[0]*sy + v[1]*sx*cy + v[2]*cx*cy;
          for (int k = 0; k < 3; k++) if (odom[3+k] >= 0) s_qvel[odom[3+k]] = v[3+k];
        }
        WSYNC();
      }
    }
  }  // one step
  if constexpr (LOOP) { if (++step < nsteps) goto step_again; }

  PROF(14);
  if ((ph & (PH_FKONLY | PH_MULM)) || (xflags & XF_NOSTORE)) return;
  // ------------------------------------------------------------------ store state
  for (int i = lane; i < nq; i += 64) S.qpos[qrow + i] = s_qpos[i];
  for (int i = lane; i < nv; i += 64) {
    // qacc and qacc_warmstart are ONE array in HBM (S.qacc aliases S.qacc_ws): every path that solves for qacc also stores
    // it as the next warm start (mj_advance: qacc_warmstart = qacc), and no path changes one without the other
    S.qvel[vrow + i] = s_qvel[i]; S.qacc_ws[vrow + i] = s_ws[i];
    if (!((ph & PH_STEP1) && (ph & PH_STEP2))) { S.qvel_ref[vrow + i] = s_qvref[i]; S.qfrc_applied[vrow + i] = s_applied[i]; }
  }
  if (lane == 0) {
    S.time[env] = time;
    S.stats[4*env] = ncon; S.stats[4*env+1] = nefc; S.stats[4*env+2] = niter; S.stats[4*env+3] = ((S.stats[4*env+3] | flags) & 0xff) | (cost_hint << 8);
  }
  PROF(15);
  if ((xflags & XF_PROF) && lane == 0) S.x_prof[(size_t)blockIdx.x * PROF_STRIDE + 17] = (long long)__builtin_amdgcn_s_memrealtime();
}

#ifndef MJH_WINDOW_TU
// pack time + qpos + qvel per env into one contiguous fp32 buffer (feeds the RCCL all-gather)
__global__ void mjh_export_kernel(const DState S, float* out, int env0, int nenv, int nq, int nv, int nqp, int nvp) {
  const int stride = 1 + nq + nv;
  out += (size_t)env0 * stride;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)nenv * stride; i += (size_t)gridDim.x * blockDim.x) {
    const int e = env0 + (int)(i / stride), k = (int)(i % stride);
    out[i] = (k == 0) ? (float)S.time[e] : (k <= nq ? S.qpos[(size_t)e * nqp + k - 1] : S.qvel[(size_t)e * nvp + k - 1 - nq]);
  }
}

// Longest-processing-time-first order of the environments for the next launch: counting sort (descending) of the
// previous step's cost estimate (solver sweeps x constraint rows) in one 1024-thread workgroup.
__global__ __launch_bounds__(1024) void mjh_order_kernel(const int* __restrict__ stats, int* __restrict__ order, int env0, int nenv, int* __restrict__ dense_sel, int dense_min_iter) {
  stats += 4 * (size_t)env0; order += env0;   // this cohort's slice; order[] holds absolute env ids
  if (dense_sel) {   // dense row-space solver (dense_pgs.h) for this cohort's next steps iff one of its envs swept long in the last one:
    __shared__ int mx;   // one word in host-mapped memory; the host reads it, unsynchronised, when it queues the cohort's next steps
    if (threadIdx.x == 0) mx = 0;
    __syncthreads();
    int m = 0;
    // (dense_min_iter < 0 — window models: the word receives the cohort's largest constraint-row count instead; engine.hip: LDS tier)
    for (int e = threadIdx.x; e < nenv; e += 1024) m = max(m, stats[4*e + (dense_min_iter < 0 ? 1 : 2)]);
    atomicMax(&mx, m);
    __syncthreads();
    if (threadIdx.x == 0) *dense_sel = dense_min_iter < 0 ? mx : (mx >= dense_min_iter ? 1 : 0);
  }
  __shared__ int hist[256], base[256];
  const int t = threadIdx.x;
  if (t < 256) hist[t] = 0;
  __syncthreads();
  // (kernels that know their solver work better leave a hint in bits 8.. of the flag word: the patch sweep's sweeps x step cost)
  auto bucket = [&](int e) { const int hint = stats[4*e + 3] >> 8; const int cost = hint ? hint : stats[4*e + 2] * (stats[4*e + 1] + 24); int b = cost >> 6; return b > 255 ? 255 : (b < 0 ? 0 : b); };   // 100 it x 232 rows -> 362 -> clamp
  for (int e = t; e < nenv; e += 1024) atomicAdd(&hist[255 - bucket(e)], 1);
  __syncthreads();
  if (t == 0) { int acc = 0; for (int b = 0; b < 256; b++) { base[b] = acc; acc += hist[b]; } }
  __syncthreads();
  for (int e = t; e < nenv; e += 1024) order[atomicAdd(&base[255 - bucket(e)], 1)] = env0 + e;
}

#endif   // MJH_WINDOW_TU

// Stand-alone solver of the many-body layout's three-launch step: everything it needs is in the env's scratch slice
// (pools + hand-over vectors), its LDS footprint is two dof vectors, so many environments are resident per CU while the
// fused kernel holds ~70 KB per env for the stages around the sweeps.
template <bool DIAGM, bool EXTRA>
DEV void mjh_solve_body(const DConst* __restrict__ C, const DState& S, int env0) {
  const DModel& M = C->M;
  const Lay& L = C->L;
  extern __shared__ float lds[];
  // 64 threads (one wave per environment) or 64 x nwave (wide groups: the waves share the chunks of a group)
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, nv = M.nv;
  const int env = S.env_order ? S.env_order[env0 + blockIdx.x] : env0 + (int)blockIdx.x;
  float* const gs = S.gscratch + (size_t)env * (size_t)S.gstride;
  int* meta = (int*)(gs + L.g_meta);
  const int nblk = __builtin_amdgcn_readfirstlane(meta[0]);
  if (nblk == 0) return;                                  // unconstrained env: the assemble launch wrote qacc itself
  if (__builtin_amdgcn_readfirstlane(meta[7]) != 0) return;   // solved by the dense kernels (dense_pgs.h)
  const int nv4 = ((nv + 3) / 4) * 4, ngrp = __builtin_amdgcn_readfirstlane(meta[6]);
  float* s_qacc = lds; float* s_minv = lds + nv4;
  float* s_red = lds + 2 * nv4;                                               // per-wave partial sums (8 floats)
  int* s_ord = (int*)(s_red + 8); int* s_gst = s_ord + nblk;                  // visiting order, group starts (many-block models)
  const int* g_ord = (const int*)(gs + (-1 - L.order)); const int* g_gst = (const int*)(gs + (-1 - L.sched));
  for (int d = tid; d < nv; d += nthr) { s_qacc[d] = gs[L.g_a0 + d]; s_minv[d] = gs[L.g_minv + d]; }
  {
    const int4* g_hd = (const int4*)(gs + (-1 - L.blki));
    for (int i = tid; i < nblk; i += nthr) { const int b = g_ord[i]; const int4 hd = g_hd[b]; s_ord[i] = MJH_ORDER_WORD(b, hd); }
  }
  if (nblk > 64) for (int i = tid; i <= ngrp; i += nthr) s_gst[i] = g_gst[i];
  // quad sweep (free-body piles with groups of up to 16 blocks, one wave per env): padded copies, four floats per 3 dofs
  const bool quad16 = DIAGM && M.group_max == 16 && M.rowW <= 12 && nthr == 64 && nblk > 64;
  const int nt4 = 4 * ((nv + 2) / 3);
  float* s_a4 = (float*)(((size_t)(s_gst + M.maxblk + 2) + 15) & ~(size_t)15); float* s_m4 = s_a4 + nt4;
  if (quad16) for (int d = tid; d < nv; d += nthr) { const int t = d / 3, r = d - 3 * t; s_a4[4*t + r] = gs[L.g_a0 + d]; s_m4[4*t + r] = gs[L.g_minv + d]; }
  __syncthreads();
  ManyCtx mc;
  mc.J = gs + (-1 - L.J); mc.B = gs + (-1 - L.B); mc.blkf = gs + (-1 - L.blkf); mc.blkq = gs + (-1 - L.blkq); mc.ext = gs + (-1 - L.ext);
  mc.blki = (const int*)(gs + (-1 - L.blki)); mc.order = s_ord; mc.gstart = s_gst; mc.ngrp = ngrp; mc.order_packed = true;
  mc.qacc = s_qacc; mc.qLDinv = s_minv;
  mc.nblk = nblk; mc.nfixblk = __builtin_amdgcn_readfirstlane(meta[1]); mc.rowW = M.rowW; mc.iterations = M.iterations;
  mc.has_dim4 = M.has_dim4 != 0; mc.scale = 1.0f / (M.meaninertia * (float)(nv > 1 ? nv : 1)); mc.tolerance = M.tolerance;
  mc.noslip_iterations = M.noslip_iterations; mc.noslip_tolerance = M.noslip_tolerance;
  mc.nwave = nthr >> 6; mc.wid = tid >> 6; mc.red = s_red;
  mc.a4 = quad16 ? s_a4 : nullptr; mc.m4 = quad16 ? s_m4 : nullptr;
  {
    // (debug probe: the operand stream of a few envs' slices — resident in L2 — instead of every env's own: what the sweeps cost without the
    //  stream from MALL / HBM; the results are garbage and only timed)
    const unsigned long long ga = (unsigned long long)(S.probe_slices > 0 ? S.gscratch + (size_t)(env0 + (int)(blockIdx.x % (unsigned)S.probe_slices)) * (size_t)S.gstride : gs);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ga), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ga >> 32));
    const long long nb = S.gstride * 4;
    mc.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)(nb > 0x7ffffff0ll ? 0x7ffffff0ll : nb), 0x00020000);
    mc.oJ = 4 * (-1 - L.J); mc.oB = 4 * (-1 - L.B); mc.oblkf = 4 * (-1 - L.blkf); mc.oblkq = 4 * (-1 - L.blkq); mc.oext = 4 * (-1 - L.ext); mc.oblki = 4 * (-1 - L.blki);
  }
  const int niter = pgs_many_body<DIAGM, EXTRA, true>(mc, lane);
  __syncthreads();
  if (quad16 && nblk > 64 && M.rowW <= 16 && ngrp >= 2) { for (int d = tid; d < nv; d += nthr) { const int t = d / 3, r = d - 3 * t; gs[L.g_qacc + d] = s_a4[4*t + r]; } }
  else
  for (int d = tid; d < nv; d += nthr) gs[L.g_qacc + d] = s_qacc[d];
  if (tid == 0) meta[5] = niter;
}
template <bool DIAGM, bool EXTRA>
#ifndef MJH_SOLVE_EU_WAVES
#define MJH_SOLVE_EU_WAVES 1
#endif
__global__ __launch_bounds__(256, MJH_SOLVE_EU_WAVES) void mjh_solve_kernel(const DConst* __restrict__ C, const DState S, int env0) { mjh_solve_body<DIAGM, EXTRA>(C, S, env0); }
