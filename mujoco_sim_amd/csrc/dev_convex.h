// dev_convex.h — generic convex - convex narrow phase in fp32, one candidate pair per lane: Minkowski portal
// refinement over support mappings (sphere, capsule, cylinder, box, ellipsoid, convex mesh), one contact per pair.
// Follows the test oracle's restatement (oracle/mjh_oracle.c mpr_penetration) step for step; thresholds that are
// absolute in fp64 there are loosened to fp32 resolution here.  Everything is statically indexed (no scratch).
#pragma once
#include "dev_math.h"

#define MPR_TOL 1e-6f
#define MPR_ITER 50
#define MPR_EPS_LEN2 1e-10f
#define MPR_EPS_VOL 1e-7f

struct CvxGeom { int type; float pos[3], mat[9], size[3], pad; const float* vert; int nvert; };
struct MprPt { float v[3], s[3]; };   // v = a - b (Minkowski difference), s = a + b (twice the midpoint of the witnesses)

DEV void mpr_tri_normal(float* n, const MprPt& p1, const MprPt& p2, const MprPt& p3) {
  const float e1[3] = {p2.v[0]-p1.v[0], p2.v[1]-p1.v[1], p2.v[2]-p1.v[2]}, e2[3] = {p3.v[0]-p1.v[0], p3.v[1]-p1.v[1], p3.v[2]-p1.v[2]};
  cross3(n, e1, e2); normalize3(n);
}
DEV bool mpr_converged(const MprPt& p1, const MprPt& p2, const MprPt& p3, const MprPt& p4, const float* n) {
  const float d4 = dot3(p4.v, n);
  const float m = fminf(d4 - dot3(p1.v, n), fminf(d4 - dot3(p2.v, n), d4 - dot3(p3.v, n)));
  return m <= MPR_TOL;
}
// q <- (take ? p : q), component-wise selects (a conditional struct assignment becomes a pointer select + scratch)
DEV void mpr_take(MprPt& q, const MprPt& p, bool take) {
#pragma unroll
  for (int k = 0; k < 3; k++) { q.v[k] = take ? p.v[k] : q.v[k]; q.s[k] = take ? p.s[k] : q.s[k]; }
}
DEV void mpr_expand(const MprPt& p0, MprPt& p1, MprPt& p2, MprPt& p3, const MprPt& p4) {
  float c[3];
  cross3(c, p4.v, p0.v);
  const bool a = dot3(p1.v, c) > 0, b = dot3(p2.v, c) > 0, d = dot3(p3.v, c) > 0;
  const bool r1 = a ? b : !d, r2 = !a && d, r3 = a && !b;
  mpr_take(p1, p4, r1); mpr_take(p2, p4, r2); mpr_take(p3, p4, r3);
}
DEV void tri_closest_to_origin(const float* a, const float* b, const float* c, float* out) {
  const float ab[3] = {b[0]-a[0], b[1]-a[1], b[2]-a[2]}, ac[3] = {c[0]-a[0], c[1]-a[1], c[2]-a[2]};
  const float d1 = -dot3(ab, a), d2 = -dot3(ac, a);
  if (d1 <= 0 && d2 <= 0) { out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; return; }
  const float d3 = -dot3(ab, b), d4 = -dot3(ac, b);
  if (d3 >= 0 && d4 <= d3) { out[0] = b[0]; out[1] = b[1]; out[2] = b[2]; return; }
  const float vc = d1*d4 - d3*d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { const float t = d1 / (d1 - d3); out[0] = a[0] + t*ab[0]; out[1] = a[1] + t*ab[1]; out[2] = a[2] + t*ab[2]; return; }
  const float d5 = -dot3(ab, c), d6 = -dot3(ac, c);
  if (d6 >= 0 && d5 <= d6) { out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; return; }
  const float vb = d5*d2 - d1*d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { const float t = d2 / (d2 - d6); out[0] = a[0] + t*ac[0]; out[1] = a[1] + t*ac[1]; out[2] = a[2] + t*ac[2]; return; }
  const float va = d3*d6 - d5*d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    const float t = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    out[0] = b[0] + t*(c[0]-b[0]); out[1] = b[1] + t*(c[1]-b[1]); out[2] = b[2] + t*(c[2]-b[2]); return;
  }
  const float den = 1.0f / (va + vb + vc), v = vb * den, w = vc * den;
  out[0] = a[0] + v*ab[0] + w*ac[0]; out[1] = a[1] + v*ab[1] + w*ac[1]; out[2] = a[2] + v*ab[2] + w*ac[2];
}

// ---- wave-cooperative form.  A lane owns a candidate pair and runs the portal algorithm on it; what it cannot do well alone is
// the support mapping of a convex MESH: a scan over the hull's vertices, which on 30 lanes with 30 different hulls is 30-way
// uncoalesced — every load instruction of the scan costs what 64 separate cache-line requests cost (PR2's 37 hulls: 230 k of the
// 360 k clocks in front of the solver).  So the wave stays converged over the trips of the phase loop, and the mesh scans of a
// trip are SERVED by the whole wave, one request after the other (four in flight): lanes = vertices, one coalesced load of the
// hull, a wave-wide arg-max with the sequential scan's tie rule (lowest vertex index), the winner handed to the requesting lane.
DEV void cvx_support_local(const CvxGeom& g, const float* dl, float* pl) {       // analytic shapes: the lane on its own
  const float* s = g.size;
  pl[0] = pl[1] = pl[2] = 0;
  if (g.type == MJH_GEOM_SPHERE) { pl[0] = s[0]*dl[0]; pl[1] = s[0]*dl[1]; pl[2] = s[0]*dl[2]; }
  else if (g.type == MJH_GEOM_CAPSULE) { pl[0] = s[0]*dl[0]; pl[1] = s[0]*dl[1]; pl[2] = s[0]*dl[2] + (dl[2] >= 0 ? s[1] : -s[1]); }
  else if (g.type == MJH_GEOM_CYLINDER) {
    const float r2 = dl[0]*dl[0] + dl[1]*dl[1];
    if (r2 > MPR_EPS_LEN2) { const float sc = s[0] * rsqrtf(r2); pl[0] = dl[0]*sc; pl[1] = dl[1]*sc; }
    pl[2] = dl[2] >= 0 ? s[1] : -s[1];
  } else if (g.type == MJH_GEOM_BOX) { pl[0] = dl[0] >= 0 ? s[0] : -s[0]; pl[1] = dl[1] >= 0 ? s[1] : -s[1]; pl[2] = dl[2] >= 0 ? s[2] : -s[2]; }
  else if (g.type == MJH_GEOM_ELLIPSOID) {
    const float w0 = s[0]*s[0]*dl[0], w1 = s[1]*s[1]*dl[1], w2 = s[2]*s[2]*dl[2];
    const float den = sqrtf(w0*dl[0] + w1*dl[1] + w2*dl[2]);
    if (den > MJ_MINVAL) { const float inv = 1.0f / den; pl[0] = w0*inv; pl[1] = w1*inv; pl[2] = w2*inv; }
  }
}
// farthest vertex of the hull of every lane with `want` along its local direction dl -> pl (other lanes: pl untouched)
DEV void mesh_support_wave(const bool want, const float* vert, const int nvert, const float* dl, float* pl, const int lane) {
  unsigned long long req = __ballot(want);
  const unsigned long long va = (unsigned long long)vert;
  const int valo = (int)(unsigned)va, vahi = (int)(unsigned)(va >> 32);
  while (req) {
    constexpr int NB = 4;                      // requests in flight
    int L[NB], nv[NB]; const float* vp[NB]; float d0[NB], d1[NB], d2[NB]; bool on[NB];
#pragma unroll
    for (int u = 0; u < NB; u++) {
      on[u] = req != 0;
      L[u] = on[u] ? __ffsll((long long)req) - 1 : 0;
      if (on[u]) req &= req - 1;
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane(valo, L[u]), hi = (unsigned)__builtin_amdgcn_readlane(vahi, L[u]);
      vp[u] = (const float*)(((unsigned long long)hi << 32) | lo);
      nv[u] = on[u] ? __builtin_amdgcn_readlane(nvert, L[u]) : 0;
      d0[u] = readlane_f(dl[0], L[u]); d1[u] = readlane_f(dl[1], L[u]); d2[u] = readlane_f(dl[2], L[u]);
    }
    float bx[NB], by[NB], bz[NB], bd[NB]; int bi[NB];
    // first 64 vertices of every request: the loads go out together
#pragma unroll
    for (int u = 0; u < NB; u++) {
      bd[u] = -3.0e38f; bi[u] = 0x7fffffff; bx[u] = by[u] = bz[u] = 0;
      if (on[u]) { const int i = min(lane, nv[u] - 1); bx[u] = vp[u][3*i]; by[u] = vp[u][3*i+1]; bz[u] = vp[u][3*i+2]; }
    }
#pragma unroll
    for (int u = 0; u < NB; u++) {
      if (!on[u]) continue;                                       // (uniform)
      if (lane < nv[u]) { bd[u] = bx[u]*d0[u] + by[u]*d1[u] + bz[u]*d2[u]; bi[u] = lane; }
      for (int i = lane + 64; i < nv[u]; i += 64) {               // hulls with more than 64 vertices
        const float x = vp[u][3*i], y = vp[u][3*i+1], z = vp[u][3*i+2], dp = x*d0[u] + y*d1[u] + z*d2[u];
        if (dp > bd[u]) { bd[u] = dp; bi[u] = i; bx[u] = x; by[u] = y; bz[u] = z; }
      }
      const float gmax = wave_max_f(bd[u]);
      const bool cand = bd[u] == gmax && bi[u] != 0x7fffffff;
      unsigned long long m = __ballot(cand);
      if (__popcll(m) > 1) { const int wi = wave_min_i(cand ? bi[u] : 0x7fffffff); m = __ballot(cand && bi[u] == wi); }   // ties: the lowest index, as a sequential scan keeps it
      const int wl = m ? __ffsll((long long)m) - 1 : 0;
      const float wx = readlane_f(bx[u], wl), wy = readlane_f(by[u], wl), wz = readlane_f(bz[u], wl);
      if (lane == L[u]) { pl[0] = wx; pl[1] = wy; pl[2] = wz; }
    }
  }
}
// support points of both geoms of every active lane along +-dir; mesh scans are served by the wave (converged call)
DEV void mpr_support_wave(const CvxGeom& g1, const CvxGeom& g2, const float* dir, MprPt& p, const bool act, const int lane) {
  const float nd[3] = {-dir[0], -dir[1], -dir[2]};
  float dl1[3], dl2[3], pl1[3], pl2[3], a[3], b[3];
  rotvecT(dl1, g1.mat, dir); rotvecT(dl2, g2.mat, nd);
  cvx_support_local(g1, dl1, pl1); cvx_support_local(g2, dl2, pl2);
  mesh_support_wave(act && g1.type == MJH_GEOM_MESH, g1.vert, g1.nvert, dl1, pl1, lane);
  mesh_support_wave(act && g2.type == MJH_GEOM_MESH, g2.vert, g2.nvert, dl2, pl2, lane);
  rotvec(a, g1.mat, pl1); rotvec(b, g2.mat, pl2);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    a[k] += g1.pos[k] + g1.pad*dir[k]; b[k] += g2.pos[k] + g2.pad*nd[k];
    p.v[k] = a[k] - b[k]; p.s[k] = a[k] + b[k];
  }
}

// one raw contact {dist, pos, normal from geom1 to geom2} or nothing, for every lane with `act` (the wave calls it converged).
// The five places of the algorithm that ask for a support point (first and second portal vertex, portal discovery, refinement,
// push to the surface) share ONE support evaluation per trip of a phase loop: lanes that are in different phases still evaluate
// their support mappings together, and the wave leaves the loop when its last pair is decided.
DEV int c_convex_wave(CvxGeom& g1, CvxGeom& g2, float margin, float* st, const bool act, const int lane) {
  g1.pad = g2.pad = 0.5f * margin;
  MprPt p0, p1, p2, p3, p4;
  float n[3], c[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { p0.v[k] = g1.pos[k] - g2.pos[k]; p0.s[k] = g1.pos[k] + g2.pos[k]; p1.v[k] = p2.v[k] = p3.v[k] = 0; p1.s[k] = p2.s[k] = p3.s[k] = 0; }
  if (dot3(p0.v, p0.v) < MPR_EPS_LEN2) p0.v[0] += 1e-4f;
  n[0] = -p0.v[0]; n[1] = -p0.v[1]; n[2] = -p0.v[2];
  normalize3(n);
  enum { PH_FIRST, PH_SECOND, PH_DISCOVER, PH_REFINE, PH_PUSH };
  int phase = PH_FIRST, it = 0;
  bool live = act;          // still iterating
  int res = 0;              // 0: no contact, 1: contact emitted on the ray p0 -> p1, 2: contact from the final portal
  while (__ballot(live) != 0) {
    mpr_support_wave(g1, g2, n, p4, live, lane);
    if (live) {
    const float d4 = dot3(p4.v, n);
    if (phase == PH_FIRST) {
      if (d4 <= 0) live = false;
      else {
      p1 = p4;
      cross3(n, p0.v, p1.v);
      if (dot3(n, n) < 1e-12f * dot3(p0.v, p0.v) * dot3(p1.v, p1.v)) {   // origin on the ray p0 -> p1
        const float depth = norm3(p1.v);
        float dir[3] = {p1.v[0], p1.v[1], p1.v[2]}; normalize3(dir);
        const float pos[3] = {0.5f * p1.s[0], 0.5f * p1.s[1], 0.5f * p1.s[2]};
        raw_emit(st, 0, margin - depth, pos, dir);
        res = 1; live = false;
      } else {
      normalize3(n);
      phase = PH_SECOND;
      } }
    } else if (phase == PH_SECOND) {
      if (d4 <= 0) live = false;
      else {
      p2 = p4;
      const float e1[3] = {p1.v[0]-p0.v[0], p1.v[1]-p0.v[1], p1.v[2]-p0.v[2]}, e2[3] = {p2.v[0]-p0.v[0], p2.v[1]-p0.v[1], p2.v[2]-p0.v[2]};
      cross3(n, e1, e2); normalize3(n);
      const bool sw = dot3(n, p0.v) > 0;
      const MprPt t = p1;
      mpr_take(p1, p2, sw); mpr_take(p2, t, sw);
      n[0] = sw ? -n[0] : n[0]; n[1] = sw ? -n[1] : n[1]; n[2] = sw ? -n[2] : n[2];
      phase = PH_DISCOVER; it = 0;
      }
    } else if (phase == PH_DISCOVER) {
      if (d4 <= 0 || it > MPR_ITER) live = false;
      else {
      it++;
      cross3(c, p1.v, p4.v);
      const bool t2 = dot3(c, p0.v) < -MPR_EPS_VOL;
      cross3(c, p4.v, p2.v);
      const bool t1 = !t2 && dot3(c, p0.v) < -MPR_EPS_VOL;
      if (t1 || t2) {
        mpr_take(p2, p4, t2); mpr_take(p1, p4, t1);
        const float e1[3] = {p1.v[0]-p0.v[0], p1.v[1]-p0.v[1], p1.v[2]-p0.v[2]}, e2[3] = {p2.v[0]-p0.v[0], p2.v[1]-p0.v[1], p2.v[2]-p0.v[2]};
        cross3(n, e1, e2); normalize3(n);
      } else {
        p3 = p4;
        mpr_tri_normal(n, p1, p2, p3);
        phase = dot3(n, p1.v) >= -MPR_EPS_VOL ? PH_PUSH : PH_REFINE; it = 0;
      }
      }
    } else if (phase == PH_REFINE) {
      if (d4 < -MPR_EPS_VOL || mpr_converged(p1, p2, p3, p4, n) || it > MPR_ITER) live = false;
      else {
      it++;
      mpr_expand(p0, p1, p2, p3, p4);
      mpr_tri_normal(n, p1, p2, p3);
      if (dot3(n, p1.v) >= -MPR_EPS_VOL) { phase = PH_PUSH; it = 0; }
      }
    } else {
      if (mpr_converged(p1, p2, p3, p4, n) || it > MPR_ITER) { res = 2; live = false; }
      else {
      it++;
      mpr_expand(p0, p1, p2, p3, p4);
      mpr_tri_normal(n, p1, p2, p3);
      }
    }
    }
  }
  if (res != 2) return res;
  float depth, dir[3], pos[3];
  tri_closest_to_origin(p1.v, p2.v, p3.v, c);
  depth = norm3(c);
  if (depth < 1e-7f) { dir[0] = n[0]; dir[1] = n[1]; dir[2] = n[2]; } else { const float inv = 1.0f / depth; dir[0] = c[0]*inv; dir[1] = c[1]*inv; dir[2] = c[2]*inv; }
  float b0, b1, b2, b3, x[3];
  cross3(x, p1.v, p2.v); b0 = dot3(x, p3.v);
  cross3(x, p3.v, p2.v); b1 = dot3(x, p0.v);
  cross3(x, p0.v, p1.v); b2 = dot3(x, p3.v);
  cross3(x, p2.v, p1.v); b3 = dot3(x, p0.v);
  float sum = b0 + b1 + b2 + b3;
  if (sum <= 0) {
    b0 = 0;
    cross3(x, p2.v, p3.v); b1 = dot3(x, n);
    cross3(x, p3.v, p1.v); b2 = dot3(x, n);
    cross3(x, p1.v, p2.v); b3 = dot3(x, n);
    sum = b1 + b2 + b3;
  }
  const float inv = 0.5f / sum;
#pragma unroll
  for (int k = 0; k < 3; k++) pos[k] = (b0*p0.s[k] + b1*p1.s[k] + b2*p2.s[k] + b3*p3.s[k]) * inv;
  raw_emit(st, 0, margin - depth, pos, dir);
  return 1;
}

// pairs without an analytic routine (types ordered t1 <= t2); same table as the oracle's pair_is_convex
DEV bool pair_is_convex(int t1, int t2) {
  if (t1 == MJH_GEOM_PLANE || t1 == MJH_GEOM_HFIELD || t2 == MJH_GEOM_HFIELD) return false;
  if (t1 == MJH_GEOM_ELLIPSOID || t2 == MJH_GEOM_ELLIPSOID || t1 == MJH_GEOM_CYLINDER || t2 == MJH_GEOM_CYLINDER) return true;
  if (t2 == MJH_GEOM_MESH) return true;
  return t1 == MJH_GEOM_CAPSULE && t2 == MJH_GEOM_BOX;
}
