// dev_collide.h — fp32 narrow-phase primitives, one candidate geom pair per lane.
// Each routine writes up to its pair's capacity of raw contacts {dist, pos[3], normal[3]}
// (normal from geom1 to geom2) into the lane's LDS staging slots and returns the count.
// Definitions follow DESIGN.md §2 (A6) / HISTORY.md §9; plane-* / sphere-* / capsule-* follow MuJoCo's
// published primitives, box-box is this project's own SAT + face-manifold definition.
#pragma once
#include "dev_math.h"

#define RAW_STRIDE 7

DEV void raw_emit(float* st, int k, float dist, const float* pos, const float* n) {
  float* o = st + k * RAW_STRIDE;
  o[0] = dist; o[1] = pos[0]; o[2] = pos[1]; o[3] = pos[2]; o[4] = n[0]; o[5] = n[1]; o[6] = n[2];
}

DEV int c_plane_sphere(const float* pp, const float* pm, const float* c, float r, float margin, float* st, int k) {
  float n[3] = {pm[2], pm[5], pm[8]}, t[3] = {c[0]-pp[0], c[1]-pp[1], c[2]-pp[2]};
  float dist = dot3(t, n) - r;
  if (dist > margin) return 0;
  float pos[3] = {c[0] - n[0]*(r + 0.5f*dist), c[1] - n[1]*(r + 0.5f*dist), c[2] - n[2]*(r + 0.5f*dist)};
  raw_emit(st, k, dist, pos, n);
  return 1;
}

DEV int c_plane_capsule(const float* pp, const float* pm, const float* c, const float* cm, const float* size, float margin, float* st) {
  float ax[3] = {cm[2]*size[1], cm[5]*size[1], cm[8]*size[1]};
  float e1[3] = {c[0]+ax[0], c[1]+ax[1], c[2]+ax[2]}, e2[3] = {c[0]-ax[0], c[1]-ax[1], c[2]-ax[2]};
  int n = c_plane_sphere(pp, pm, e1, size[0], margin, st, 0);
  n += c_plane_sphere(pp, pm, e2, size[0], margin, st, n);
  return n;
}

// plane - cylinder: deepest rim point of the cap facing the plane, the same rim direction on the other cap, and two more
// points of the near cap at +-120 degrees (a triangle under a standing cylinder); at most 4
DEV int c_plane_cylinder(const float* pp, const float* pm, const float* c, const float* cm, const float* size, float margin, float* st) {
  const float n[3] = {pm[2], pm[5], pm[8]}, t[3] = {c[0]-pp[0], c[1]-pp[1], c[2]-pp[2]};
  float ax[3] = {cm[2], cm[5], cm[8]};
  const float r = size[0], h = size[1];
  float prjaxis = dot3(n, ax);
  if (prjaxis > 0) { ax[0] = -ax[0]; ax[1] = -ax[1]; ax[2] = -ax[2]; prjaxis = -prjaxis; }   // axis points towards the plane
  const float dist0 = dot3(t, n);
  float vec[3] = {ax[0]*prjaxis - n[0], ax[1]*prjaxis - n[1], ax[2]*prjaxis - n[2]};          // -normal without its axial part
  const float len2 = dot3(vec, vec);
  if (len2 >= 1e-10f) { const float sc = r * rsqrtf(len2); vec[0] *= sc; vec[1] *= sc; vec[2] *= sc; }   // (threshold shared with the oracle)
  else { vec[0] = cm[0] * r; vec[1] = cm[3] * r; vec[2] = cm[6] * r; }                        // cap parallel to the plane: cylinder x axis
  const float prjvec = dot3(vec, n);
  ax[0] *= h; ax[1] *= h; ax[2] *= h; prjaxis *= h;
  const float d1 = dist0 + prjaxis + prjvec;
  if (d1 > margin) return 0;
  int cnt = 0;
  { const float pos[3] = {c[0] + vec[0] + ax[0] - n[0]*0.5f*d1, c[1] + vec[1] + ax[1] - n[1]*0.5f*d1, c[2] + vec[2] + ax[2] - n[2]*0.5f*d1};
    raw_emit(st, cnt, d1, pos, n); cnt++; }
  const float d2 = dist0 - prjaxis + prjvec;
  if (d2 <= margin) {
    const float pos[3] = {c[0] + vec[0] - ax[0] - n[0]*0.5f*d2, c[1] + vec[1] - ax[1] - n[1]*0.5f*d2, c[2] + vec[2] - ax[2] - n[2]*0.5f*d2};
    raw_emit(st, cnt, d2, pos, n); cnt++;
  }
  const float d3 = dist0 + prjaxis - 0.5f * prjvec;
  if (d3 <= margin) {
    float v1[3]; cross3(v1, vec, ax);
    const float l2 = dot3(v1, v1);
    if (l2 > MJ_MINVAL * MJ_MINVAL) {
      const float sc = r * 0.8660254037844386f * rsqrtf(l2);
      v1[0] *= sc; v1[1] *= sc; v1[2] *= sc;
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const float sg = q ? -1.0f : 1.0f;
        const float pos[3] = {c[0] + sg*v1[0] + ax[0] - 0.5f*vec[0] - n[0]*0.5f*d3, c[1] + sg*v1[1] + ax[1] - 0.5f*vec[1] - n[1]*0.5f*d3,
                              c[2] + sg*v1[2] + ax[2] - 0.5f*vec[2] - n[2]*0.5f*d3};
        raw_emit(st, cnt, d3, pos, n); cnt++;
      }
    }
  }
  return cnt;
}

// plane - ellipsoid: the support point of the ellipsoid against the plane normal
DEV int c_plane_ellipsoid(const float* pp, const float* pm, const float* c, const float* em, const float* size, float margin, float* st) {
  const float n[3] = {pm[2], pm[5], pm[8]}, nn[3] = {-n[0], -n[1], -n[2]};
  float dl[3], pl[3], pw[3];
  rotvecT(dl, em, nn);
  const float w[3] = {size[0]*size[0]*dl[0], size[1]*size[1]*dl[1], size[2]*size[2]*dl[2]};
  const float inv = rsqrtf(fmaxf(w[0]*dl[0] + w[1]*dl[1] + w[2]*dl[2], 1e-30f));
  pl[0] = w[0]*inv; pl[1] = w[1]*inv; pl[2] = w[2]*inv;
  rotvec(pw, em, pl);
  pw[0] += c[0]; pw[1] += c[1]; pw[2] += c[2];
  const float t[3] = {pw[0]-pp[0], pw[1]-pp[1], pw[2]-pp[2]};
  const float dist = dot3(t, n);
  if (dist > margin) return 0;
  const float pos[3] = {pw[0] - n[0]*0.5f*dist, pw[1] - n[1]*0.5f*dist, pw[2] - n[2]*0.5f*dist};
  raw_emit(st, 0, dist, pos, n);
  return 1;
}

// plane - convex mesh: vertices below the margin; the deepest one, the one farthest from it, the one farthest from the
// line through those two and the one farthest on the other side of that line; at most 4 (same definition as the oracle)
#define PLANE_MESH_EPS2 1e-8f
DEV int c_plane_mesh(const float* pp, const float* pm, const float* c, const float* mm, const float* vert, int nvert, float margin, float* st) {
  const float n[3] = {pm[2], pm[5], pm[8]}, t[3] = {c[0]-pp[0], c[1]-pp[1], c[2]-pp[2]};
  float nl[3];
  rotvecT(nl, mm, n);
  const float d0 = dot3(t, n);
  float best = 3.0e38f; int i1 = -1, i2 = -1, ipos = -1, ineg = -1;
  mesh_scan(vert, nvert, [&](const int i, const float x, const float y, const float z) __attribute__((always_inline)) {
    const float di = d0 + x*nl[0] + y*nl[1] + z*nl[2]; if (di < best) { best = di; i1 = i; } });
  if (i1 < 0 || best > margin) return 0;
  const float v1[3] = {vert[3*i1], vert[3*i1+1], vert[3*i1+2]};
  best = PLANE_MESH_EPS2;
  mesh_scan(vert, nvert, [&](const int i, const float x, const float y, const float z) __attribute__((always_inline)) {
    if (d0 + x*nl[0] + y*nl[1] + z*nl[2] > margin) return;
    const float e[3] = {x - v1[0], y - v1[1], z - v1[2]}, l2 = dot3(e, e);
    if (l2 > best) { best = l2; i2 = i; }
  });
  bool posfirst = true;
  if (i2 >= 0) {
    const float e12[3] = {vert[3*i2] - v1[0], vert[3*i2+1] - v1[1], vert[3*i2+2] - v1[2]};
    float side[3];
    cross3(side, e12, nl);
    float bpos = sqrtf(PLANE_MESH_EPS2 * dot3(e12, e12)), bneg = bpos;
    mesh_scan(vert, nvert, [&](const int i, const float x, const float y, const float z) __attribute__((always_inline)) {
      if (d0 + x*nl[0] + y*nl[1] + z*nl[2] > margin) return;
      const float sd = (x - v1[0])*side[0] + (y - v1[1])*side[1] + (z - v1[2])*side[2];
      if (sd > bpos) { bpos = sd; ipos = i; }
      if (-sd > bneg) { bneg = -sd; ineg = i; }
    });
    posfirst = bpos >= bneg;
  }
  const int third = (ipos >= 0 && ineg >= 0) ? (posfirst ? ipos : ineg) : (ipos >= 0 ? ipos : ineg);
  const int fourth = (ipos >= 0 && ineg >= 0) ? (posfirst ? ineg : ipos) : -1;
  int cnt = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int iv = q == 0 ? i1 : q == 1 ? i2 : q == 2 ? third : fourth;
    if (iv < 0) continue;
    const float v[3] = {vert[3*iv], vert[3*iv+1], vert[3*iv+2]};
    float w[3];
    rotvec(w, mm, v);
    const float di = d0 + dot3(v, nl);
    const float pos[3] = {c[0] + w[0] - n[0]*0.5f*di, c[1] + w[1] - n[1]*0.5f*di, c[2] + w[2] - n[2]*0.5f*di};
    raw_emit(st, cnt, di, pos, n); cnt++;
  }
  return cnt;
}

DEV int c_plane_box(const float* pp, const float* pm, const float* c, const float* bm, const float* size, float margin, float* st) {
  float n[3] = {pm[2], pm[5], pm[8]}, t[3] = {c[0]-pp[0], c[1]-pp[1], c[2]-pp[2]};
  float dist = dot3(t, n);
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    float v[3] = {(i & 1) ? size[0] : -size[0], (i & 2) ? size[1] : -size[1], (i & 4) ? size[2] : -size[2]}, corner[3];
    rotvec(corner, bm, v);
    float ldist = dot3(n, corner);
    if (dist + ldist > margin || ldist > 0 || cnt >= 4) continue;
    float dd = dist + ldist;
    float pos[3] = {corner[0] + c[0] - n[0]*0.5f*dd, corner[1] + c[1] - n[1]*0.5f*dd, corner[2] + c[2] - n[2]*0.5f*dd};
    raw_emit(st, cnt, dd, pos, n);
    cnt++;
  }
  return cnt;
}

DEV int c_sphere_sphere(const float* c1, float r1, const float* c2, float r2, float margin, float* st) {
  float t[3] = {c2[0]-c1[0], c2[1]-c1[1], c2[2]-c1[2]};
  float len = norm3(t), dist = len - r1 - r2;
  if (dist > margin) return 0;
  if (len < MJ_MINVAL) { t[0] = 1; t[1] = 0; t[2] = 0; } else { float s = 1.0f/len; t[0] *= s; t[1] *= s; t[2] *= s; }
  float pos[3] = {c1[0] + t[0]*(r1 + 0.5f*dist), c1[1] + t[1]*(r1 + 0.5f*dist), c1[2] + t[2]*(r1 + 0.5f*dist)};
  raw_emit(st, 0, dist, pos, t);
  return 1;
}

DEV int c_sphere_capsule(const float* c1, float r1, const float* c2, const float* m2, const float* s2, float margin, float* st) {
  float ax[3] = {m2[2], m2[5], m2[8]}, t[3] = {c1[0]-c2[0], c1[1]-c2[1], c1[2]-c2[2]};
  float x = fminf(s2[1], fmaxf(-s2[1], dot3(ax, t)));
  float p[3] = {c2[0] + ax[0]*x, c2[1] + ax[1]*x, c2[2] + ax[2]*x};
  return c_sphere_sphere(c1, r1, p, s2[0], margin, st);
}

DEV int c_capsule_capsule(const float* c1, const float* m1, const float* s1, const float* c2, const float* m2, const float* s2, float margin, float* st) {
  float a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]}, dif[3] = {c1[0]-c2[0], c1[1]-c2[1], c1[2]-c2[2]};
  float ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif);
  float det = ma * mc - mb * mb, x1, x2;
  if (fabsf(det) >= 1e-12f) {
    x1 = (mc * u - mb * v) / det; x2 = (ma * v - mb * u) / det;
    if (x1 > s1[1]) { x1 = s1[1]; x2 = (v - mb * x1) / mc; } else if (x1 < -s1[1]) { x1 = -s1[1]; x2 = (v - mb * x1) / mc; }
    if (x2 > s2[1]) { x2 = s2[1]; x1 = fminf(s1[1], fmaxf(-s1[1], (u - mb * x2) / ma)); }
    else if (x2 < -s2[1]) { x2 = -s2[1]; x1 = fminf(s1[1], fmaxf(-s1[1], (u - mb * x2) / ma)); }
  } else {
    x2 = fminf(s2[1], fmaxf(-s2[1], v / mc));
    x1 = fminf(s1[1], fmaxf(-s1[1], (u - mb * x2) / ma));
  }
  float p1[3] = {c1[0] + a1[0]*x1, c1[1] + a1[1]*x1, c1[2] + a1[2]*x1}, p2[3] = {c2[0] + a2[0]*x2, c2[1] + a2[1]*x2, c2[2] + a2[2]*x2};
  return c_sphere_sphere(p1, s1[0], p2, s2[0], margin, st);
}

DEV int c_sphere_box(const float* c1, float r1, const float* c2, const float* m2, const float* s2, float margin, float* st) {
  float t[3] = {c1[0]-c2[0], c1[1]-c2[1], c1[2]-c2[2]}, loc[3], cl[3];
  rotvecT(loc, m2, t);
  bool inside = true;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    cl[k] = loc[k];
    if (cl[k] > s2[k]) { cl[k] = s2[k]; inside = false; } else if (cl[k] < -s2[k]) { cl[k] = -s2[k]; inside = false; }
  }
  float nloc[3], dist, ploc[3];
  if (!inside) {
    float dv[3] = {loc[0]-cl[0], loc[1]-cl[1], loc[2]-cl[2]};
    float len = norm3(dv);
    dist = len - r1;
    if (dist > margin) return 0;
    float inv = 1.0f / len;
#pragma unroll
    for (int k = 0; k < 3; k++) { nloc[k] = -dv[k] * inv; ploc[k] = cl[k] - nloc[k] * 0.5f * dist; }
  } else {
    float d0 = s2[0] - fabsf(loc[0]), d1 = s2[1] - fabsf(loc[1]), d2 = s2[2] - fabsf(loc[2]);
    int best = 0; float bd = d0;
    if (d1 < bd) { bd = d1; best = 1; }
    if (d2 < bd) { bd = d2; best = 2; }
    dist = -bd - r1;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float sg = loc[k] >= 0 ? 1.0f : -1.0f;
      nloc[k] = (k == best) ? -sg : 0.0f;
      ploc[k] = (k == best) ? (sg * s2[k] - nloc[k] * 0.5f * dist) : loc[k];
    }
  }
  float n[3], pw[3];
  rotvec(n, m2, nloc); rotvec(pw, m2, ploc);
  pw[0] += c2[0]; pw[1] += c2[1]; pw[2] += c2[2];
  raw_emit(st, 0, dist, pw, n);
  return 1;
}

// column c of a row-major 3x3
DEV void mcol(float* r, const float* m, int c) {
  r[0] = sel3(m[0], m[1], m[2], c); r[1] = sel3(m[3], m[4], m[5], c); r[2] = sel3(m[6], m[7], m[8], c);
}

// Box-box: 15-axis SAT, then either one edge-edge point or a face manifold of <= 8 points made of
// (a) incident-face vertices inside the reference rectangle, (b) reference corners projected on
// the incident face, (c) incident edges crossing the rectangle sides.  Same definition as the
// test oracle (oracle/mjh_oracle.c orc_box_box); all indexing is select-based (no scratch).
DEV int c_box_box(const float* p1, const float* m1, const float* s1, const float* p2, const float* m2, const float* s2,
                  float margin, float* st) {
  float t[3] = {p2[0]-p1[0], p2[1]-p1[1], p2[2]-p1[2]};
  float C[9], AC[9], ta[3], tb[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    float ai[3] = {m1[i], m1[3+i], m1[6+i]};
    ta[i] = dot3(t, ai);
#pragma unroll
    for (int j = 0; j < 3; j++) { float bj[3] = {m2[j], m2[3+j], m2[6+j]}; C[3*i+j] = dot3(ai, bj); AC[3*i+j] = fabsf(C[3*i+j]) + 1e-9f; }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) { float bj[3] = {m2[j], m2[3+j], m2[6+j]}; tb[j] = dot3(t, bj); }
  float sface = -1e30f; int cface = -1; bool sep = false;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    float s = fabsf(ta[i]) - (s1[i] + s2[0]*AC[3*i] + s2[1]*AC[3*i+1] + s2[2]*AC[3*i+2]);
    sep |= s > margin;
    if (s > sface) { sface = s; cface = i; }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    float s = fabsf(tb[j]) - (s2[j] + s1[0]*AC[j] + s1[1]*AC[3+j] + s1[2]*AC[6+j]);
    sep |= s > margin;
    if (s > sface) { sface = s; cface = 3 + j; }
  }
  float sedge = -1e30f; int cedge = -1;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float l2 = 1 - C[3*i+j]*C[3*i+j];
      if (l2 < 1e-6f) continue;
      float l = sqrtf(l2);
      const int i1 = (i+1)%3, i2 = (i+2)%3, j1 = (j+1)%3, j2 = (j+2)%3;
      float tl = ta[i2]*C[3*i1+j] - ta[i1]*C[3*i2+j];
      float ra = s1[i1]*AC[3*i2+j] + s1[i2]*AC[3*i1+j], rb = s2[j1]*AC[3*i+j2] + s2[j2]*AC[3*i+j1];
      float s = (fabsf(tl) - (ra + rb)) / l;
      sep |= s > margin;
      if (s > sedge) { sedge = s; cedge = 3*i + j; }
    }
  if (sep) return 0;
  if (cedge >= 0 && sedge > sface + 0.05f * fabsf(sface) + 1e-9f) {
    int i = cedge / 3, j = cedge - 3 * i;
    float ai[3], bj[3], L[3];
    mcol(ai, m1, i); mcol(bj, m2, j);
    cross3(L, ai, bj); normalize3(L);
    if (dot3(L, t) < 0) { L[0] = -L[0]; L[1] = -L[1]; L[2] = -L[2]; }
    float P1[3] = {p1[0], p1[1], p1[2]}, P2[3] = {p2[0], p2[1], p2[2]};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float ak[3] = {m1[k], m1[3+k], m1[6+k]}, bk[3] = {m2[k], m2[3+k], m2[6+k]};
      float sa = (k != i) ? ((dot3(L, ak) >= 0 ? 1.0f : -1.0f) * s1[k]) : 0.0f;
      float sb = (k != j) ? ((dot3(L, bk) >= 0 ? -1.0f : 1.0f) * s2[k]) : 0.0f;
#pragma unroll
      for (int q = 0; q < 3; q++) { P1[q] += sa * ak[q]; P2[q] += sb * bk[q]; }
    }
    float dd[3] = {P2[0]-P1[0], P2[1]-P1[1], P2[2]-P1[2]};
    float c = sel3(sel3(C[0], C[1], C[2], j), sel3(C[3], C[4], C[5], j), sel3(C[6], C[7], C[8], j), i);
    float da = dot3(dd, ai), db = dot3(dd, bj), den = 1 - c*c;
    float ha = sel3(s1[0], s1[1], s1[2], i), hb = sel3(s2[0], s2[1], s2[2], j);
    float sa = fminf(ha, fmaxf(-ha, (da - c*db) / den)), sb = fminf(hb, fmaxf(-hb, (c*da - db) / den));
    float pos[3];
#pragma unroll
    for (int q = 0; q < 3; q++) pos[q] = 0.5f * ((P1[q] + sa*ai[q]) + (P2[q] + sb*bj[q]));
    raw_emit(st, 0, sedge, pos, L);
    return 1;
  }
  // face case
  const bool flip = cface >= 3;
  const int k = flip ? cface - 3 : cface;
  const float* pR = flip ? p2 : p1; const float* mR = flip ? m2 : m1; const float* sR = flip ? s2 : s1;
  const float* pI = flip ? p1 : p2; const float* mI = flip ? m1 : m2; const float* sI = flip ? s1 : s2;
  float dpw[3] = {pI[0]-pR[0], pI[1]-pR[1], pI[2]-pR[2]}, p[3], Mm[9];
  rotvecT(p, mR, dpw);
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) Mm[3*r+c] = mR[r]*mI[c] + mR[3+r]*mI[3+c] + mR[6+r]*mI[6+c];
  const int u = (k + 1) % 3, v = (k + 2) % 3;
  // permute R-frame coordinates to (u, v, k) so that everything below is statically indexed
  float pp[3] = {sel3(p[0], p[1], p[2], u), sel3(p[0], p[1], p[2], v), sel3(p[0], p[1], p[2], k)};
  float hR[3] = {sel3(sR[0], sR[1], sR[2], u), sel3(sR[0], sR[1], sR[2], v), sel3(sR[0], sR[1], sR[2], k)};
  float Mp[9];  // rows permuted (u,v,k)
#pragma unroll
  for (int c = 0; c < 3; c++) {
    Mp[c] = sel3(Mm[c], Mm[3+c], Mm[6+c], u); Mp[3+c] = sel3(Mm[c], Mm[3+c], Mm[6+c], v); Mp[6+c] = sel3(Mm[c], Mm[3+c], Mm[6+c], k);
  }
  float sg = pp[2] >= 0 ? 1.0f : -1.0f;
  int js = 0; float best = fabsf(Mp[6]);
  if (fabsf(Mp[7]) > best) { best = fabsf(Mp[7]); js = 1; }
  if (fabsf(Mp[8]) > best) { best = fabsf(Mp[8]); js = 2; }
  const int j1 = (js + 1) % 3, j2 = (js + 2) % 3;
  float mn[3], e1[3], e2[3];
  mcol(mn, Mp, js); mcol(e1, Mp, j1); mcol(e2, Mp, j2);
  float hIs = sel3(sI[0], sI[1], sI[2], js), hI1 = sel3(sI[0], sI[1], sI[2], j1), hI2 = sel3(sI[0], sI[1], sI[2], j2);
  float tau = (sg * mn[2] > 0) ? -1.0f : 1.0f;
  float cI[3] = {pp[0] + tau*hIs*mn[0], pp[1] + tau*hIs*mn[1], pp[2] + tau*hIs*mn[2]};
  float vq[4][3];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    float a = (q == 0 || q == 3) ? hI1 : -hI1, b = (q < 2) ? hI2 : -hI2;
#pragma unroll
    for (int r = 0; r < 3; r++) vq[q][r] = cI[r] + a * e1[r] + b * e2[r];
  }
  // reference axis columns in world frame, permuted
  float cu[3], cv[3], ck[3];
  mcol(cu, mR, u); mcol(cv, mR, v); mcol(ck, mR, k);
  int cnt = 0;
  auto emit = [&](const float* cand) {
    float delta = sg * cand[2] - hR[2];
    if (delta < margin && cnt < 8) {
      float mk = cand[2] - sg * 0.5f * delta;
      float pos[3] = {pR[0] + cu[0]*cand[0] + cv[0]*cand[1] + ck[0]*mk, pR[1] + cu[1]*cand[0] + cv[1]*cand[1] + ck[1]*mk,
                      pR[2] + cu[2]*cand[0] + cv[2]*cand[1] + ck[2]*mk};
      float* o = st + cnt * RAW_STRIDE;
      o[0] = delta; o[1] = pos[0]; o[2] = pos[1]; o[3] = pos[2];
      cnt++;
    }
  };
#pragma unroll
  for (int q = 0; q < 4; q++) if (fabsf(vq[q][0]) <= hR[0] && fabsf(vq[q][1]) <= hR[1]) emit(vq[q]);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    float ru = ((q == 0 || q == 3) ? hR[0] : -hR[0]), rv = ((q < 2) ? hR[1] : -hR[1]);
    float xk = cI[2] - ((ru - cI[0]) * mn[0] + (rv - cI[1]) * mn[1]) / mn[2];
    float x[3] = {ru, rv, xk};
    float dv[3] = {x[0]-cI[0], x[1]-cI[1], x[2]-cI[2]};
    if (fabsf(dot3(dv, e1)) <= hI1 && fabsf(dot3(dv, e2)) <= hI2) emit(x);
  }
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const float* P = vq[e]; const float* Q = vq[(e + 1) % 4];
#pragma unroll
    for (int side = 0; side < 4; side++) {
      const int ax = (side < 2) ? 0 : 1, ox = 1 - ax;
      float hh = ((side & 1) ? -1.0f : 1.0f) * hR[ax];
      float fp = P[ax] - hh, fq = Q[ax] - hh;
      if (fp * fq >= 0) continue;
      float s = fp / (fp - fq);
      float cand[3] = {P[0] + s*(Q[0]-P[0]), P[1] + s*(Q[1]-P[1]), P[2] + s*(Q[2]-P[2])};
      if (fabsf(cand[ox]) < hR[ox]) emit(cand);
    }
  }
  float nsg = flip ? -sg : sg;
  for (int q = 0; q < cnt; q++) { float* o = st + q * RAW_STRIDE; o[4] = nsg*ck[0]; o[5] = nsg*ck[1]; o[6] = nsg*ck[2]; }
  return cnt;
}

// contact frame from the normal (same rule as the oracle's make_frame)
DEV void make_frame(float* f) {
  normalize3(f);
  float t[3] = {0, 0, 0};
  if (f[1] < 0.5f && f[1] > -0.5f) t[1] = 1; else t[2] = 1;
  float dp = dot3(f, t);
  t[0] -= dp*f[0]; t[1] -= dp*f[1]; t[2] -= dp*f[2];
  normalize3(t);
  f[3] = t[0]; f[4] = t[1]; f[5] = t[2];
  cross3(f + 6, f, t);
}
