// model_builder.cpp — host-side model compiler for the mjhip engine.
//
// Replaces, for programmatic scenes, the model-ingest boundary of the reference
// (mj_loadXML at include/mujoco_sim/mj_util.h:190, called from
// MjSim::load_tmp_model, src/mujoco_sim/mj_sim.cpp:804-845).  It produces the
// flat, topology-sorted mjh_model the device engine and the test oracle both read.
// Derived constants (inertia from geoms at density 1000, qpos0, invweight0,
// meaninertia, bounding radii, the static candidate pair list) are computed here
// in fp64 with a Jacobian-based mass matrix — deliberately a different formulation
// from the CRBA used on the device, so the two cross-check each other in tests.
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mjhip.h"
#include "hmath.h"

namespace {

struct BBody {
  std::string name; int parent; double pos[3], quat[4]; double gravcomp;
  bool explicit_inertial = false; double mass = 0, ipos[3] = {0,0,0}, iquat[4] = {1,0,0,0}, inertia[3] = {0,0,0};
};
struct BJoint {
  std::string name; int body, type; double pos[3], axis[3]; bool limited; double range[2];
  double damping, stiffness, armature, frictionloss, ref;
};
struct BGeom {
  std::string name; int body, type; double size[3], pos[3], quat[4], friction[3];
  int condim, contype, conaffinity; double density; int mesh = -1;
};
// convex mesh asset in its own frame: origin = centre of mass, axes = principal axes of inertia
struct BMesh {
  std::vector<double> vert;        // support-relevant vertices, own frame
  double pos[3], quat[4];          // own frame in the mesh file's frame
  double volume, inertia[3];       // unit density
  double rbound;
};
struct BEq { int type = MJH_EQ_JOINT; int j1 = -1, j2 = -1; double poly[5] = {0, 0, 0, 0, 0};        // joint coupling: joints j1, j2
              int b1 = 0, b2 = 0; double anchor[3] = {0, 0, 0}; double torquescale = 1; };       // connect / weld: bodies b1, b2
struct BSite { std::string name; int body; double pos[3], quat[4]; };
struct BSensor { std::string name; int type, site; };

}  // namespace

struct mjh_builder {
  mjh_option opt;
  int maxcon = 0, maxefc = 0;
  double boundmass = 0, boundinertia = 0;   // <compiler boundmass boundinertia> (the reference forces 1e-6, mj_sim.cpp:584-590)
  bool balanceinertia = false;              // <compiler balanceinertia> (mujoco_compile.cpp:157-160 sets it for URDF-derived models)
  std::vector<BBody> bodies;
  std::vector<BJoint> joints;
  std::vector<BGeom> geoms;
  std::vector<BMesh> meshes;
  std::vector<std::pair<int,int>> excludes;
  std::vector<BEq> eqs;
  std::vector<BSite> sites;
  std::vector<BSensor> sensors;
  std::vector<int> mocap;       // builder body ids flagged <body mocap="true">
};

static thread_local std::string g_err;
extern "C" const char* mjh_last_error(void) { return g_err.c_str(); }
void mjh_set_error(const std::string& s) { g_err = s; }

static void default_option(mjh_option* o) {
  o->timestep = 0.002; o->gravity[0] = 0; o->gravity[1] = 0; o->gravity[2] = -9.81;
  o->iterations = 100; o->tolerance = 1e-8; o->impratio = 1; o->noslip_iterations = 0; o->noslip_tolerance = 1e-6; o->disableflags = 0;
}

extern "C" mjh_builder* mjh_builder_create(void) {
  mjh_builder* b = new mjh_builder();
  default_option(&b->opt);
  BBody w; w.name = "world"; w.parent = -1; w.pos[0] = w.pos[1] = w.pos[2] = 0;
  w.quat[0] = 1; w.quat[1] = w.quat[2] = w.quat[3] = 0; w.gravcomp = 0; w.explicit_inertial = true;
  b->bodies.push_back(w);
  return b;
}
extern "C" void mjh_builder_destroy(mjh_builder* b) { delete b; }
extern "C" void mjh_builder_set_option(mjh_builder* b, const mjh_option* o) { b->opt = *o; }
extern "C" void mjh_builder_get_option(const mjh_builder* b, mjh_option* o) { *o = b->opt; }
extern "C" void mjh_builder_set_capacity(mjh_builder* b, int maxcon, int maxefc) { b->maxcon = maxcon; b->maxefc = maxefc; }
extern "C" void mjh_builder_set_bounds(mjh_builder* b, double boundmass, double boundinertia) { b->boundmass = boundmass; b->boundinertia = boundinertia; }
extern "C" void mjh_builder_set_balanceinertia(mjh_builder* b, int on) { b->balanceinertia = on != 0; }

extern "C" int mjh_builder_add_body(mjh_builder* b, const char* name, int parent, const double pos[3],
                                    const double quat[4], double gravcomp) {
  if (parent < 0 || parent >= (int)b->bodies.size()) { g_err = "add_body: bad parent"; return MJH_ERR_ARG; }
  BBody x; x.name = name ? name : ""; x.parent = parent;
  for (int i = 0; i < 3; i++) x.pos[i] = pos ? pos[i] : 0;
  if (quat) { for (int i = 0; i < 4; i++) x.quat[i] = quat[i]; hm::normalize4(x.quat); }
  else { x.quat[0] = 1; x.quat[1] = x.quat[2] = x.quat[3] = 0; }
  x.gravcomp = gravcomp;
  b->bodies.push_back(x);
  return (int)b->bodies.size() - 1;
}

extern "C" int mjh_builder_set_inertial(mjh_builder* b, int body, double mass, const double ipos[3],
                                        const double iquat[4], const double diaginertia[3]) {
  if (body <= 0 || body >= (int)b->bodies.size()) { g_err = "set_inertial: bad body"; return MJH_ERR_ARG; }
  BBody& x = b->bodies[body];
  x.explicit_inertial = true; x.mass = mass;
  for (int i = 0; i < 3; i++) { x.ipos[i] = ipos ? ipos[i] : 0; x.inertia[i] = diaginertia[i]; }
  if (iquat) { for (int i = 0; i < 4; i++) x.iquat[i] = iquat[i]; hm::normalize4(x.iquat); }
  return MJH_OK;
}

extern "C" int mjh_builder_add_joint(mjh_builder* b, const char* name, int body, int type, const double pos[3],
                                     const double axis[3], const double range[2], double damping,
                                     double stiffness, double armature, double frictionloss, double ref) {
  if (body <= 0 || body >= (int)b->bodies.size()) { g_err = "add_joint: bad body"; return MJH_ERR_ARG; }
  if (type < 0 || type > 3) { g_err = "add_joint: bad type"; return MJH_ERR_ARG; }
  BJoint j; j.name = name ? name : ""; j.body = body; j.type = type;
  for (int i = 0; i < 3; i++) { j.pos[i] = pos ? pos[i] : 0; j.axis[i] = axis ? axis[i] : (i == 2 ? 1 : 0); }
  hm::normalize3(j.axis);
  j.limited = range != nullptr; j.range[0] = range ? range[0] : 0; j.range[1] = range ? range[1] : 0;
  j.damping = damping; j.stiffness = stiffness; j.armature = armature; j.frictionloss = frictionloss; j.ref = ref;
  b->joints.push_back(j);
  return (int)b->joints.size() - 1;
}

extern "C" int mjh_builder_add_geom(mjh_builder* b, const char* name, int body, int type, const double size[3],
                                    const double pos[3], const double quat[4], const double friction[3],
                                    int condim, int contype, int conaffinity, double density) {
  if (body < 0 || body >= (int)b->bodies.size()) { g_err = "add_geom: bad body"; return MJH_ERR_ARG; }
  BGeom g; g.name = name ? name : ""; g.body = body; g.type = type;
  for (int i = 0; i < 3; i++) { g.size[i] = size ? size[i] : 0; g.pos[i] = pos ? pos[i] : 0; }
  if (quat) { for (int i = 0; i < 4; i++) g.quat[i] = quat[i]; hm::normalize4(g.quat); }
  else { g.quat[0] = 1; g.quat[1] = g.quat[2] = g.quat[3] = 0; }
  // MuJoCo defaults: friction 1 0.005 0.0001, condim 3, contype/conaffinity 1, density 1000
  g.friction[0] = friction ? friction[0] : 1.0; g.friction[1] = friction ? friction[1] : 0.005;
  g.friction[2] = friction ? friction[2] : 0.0001;
  g.condim = condim > 0 ? condim : 3; g.contype = contype >= 0 ? contype : 1;
  g.conaffinity = conaffinity >= 0 ? conaffinity : 1; g.density = density >= 0 ? density : 1000.0;   // negative = unset (MuJoCo default 1000); an explicit 0 is a massless geom
  b->geoms.push_back(g);
  return (int)b->geoms.size() - 1;
}

// ---- convex mesh assets
// Volume, centre of mass and inertia of a closed triangle mesh by signed tetrahedra against the origin
// [UPSTREAM: mj_loadXML derives mesh inertia from the faces]; thin or open meshes fall back to the bounding box.
static int add_mesh_impl(mjh_builder* b, std::vector<double> v, const int* face, int nface, const double scale[3]) {
  int nv = (int)v.size() / 3;
  if (nv < 4) { g_err = "add_mesh: fewer than 4 vertices"; return MJH_ERR_ARG; }
  if (scale) for (int i = 0; i < nv; i++) for (int k = 0; k < 3; k++) v[3*i+k] *= scale[k];
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int i = 0; i < nv; i++) for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], v[3*i+k]); hi[k] = std::max(hi[k], v[3*i+k]); }
  const double ext[3] = {hi[0]-lo[0], hi[1]-lo[1], hi[2]-lo[2]}, boxvol = ext[0]*ext[1]*ext[2];
  double vol = 0, com[3] = {0, 0, 0}, C[9] = {0};   // C = integral of x x^T
  for (int f = 0; f < nface; f++) {
    const int ia = face[3*f], ib = face[3*f+1], ic = face[3*f+2];
    if (ia < 0 || ib < 0 || ic < 0 || ia >= nv || ib >= nv || ic >= nv) { g_err = "add_mesh: face index out of range"; return MJH_ERR_ARG; }
    const double *A = &v[3*ia], *Bv = &v[3*ib], *Cv = &v[3*ic];
    double bc[3]; hm::cross(bc, Bv, Cv);
    const double tv = hm::dot3(A, bc) / 6.0, sm[3] = {A[0]+Bv[0]+Cv[0], A[1]+Bv[1]+Cv[1], A[2]+Bv[2]+Cv[2]};
    vol += tv;
    for (int k = 0; k < 3; k++) com[k] += tv * sm[k] / 4.0;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
      C[3*r+c] += tv / 20.0 * (A[r]*A[c] + Bv[r]*Bv[c] + Cv[r]*Cv[c] + sm[r]*sm[c]);
  }
  if (vol < 0) { vol = -vol; for (int k = 0; k < 3; k++) com[k] = -com[k]; for (int k = 0; k < 9; k++) C[k] = -C[k]; }   // inward-facing triangles
  BMesh M;
  double It[9];
  if (nface > 0 && boxvol > 0 && vol > 1e-4 * boxvol) {
    for (int k = 0; k < 3; k++) com[k] /= vol;
    double Cc[9];   // about the centre of mass
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Cc[3*r+c] = C[3*r+c] - vol * com[r] * com[c];
    const double tr = Cc[0] + Cc[4] + Cc[8];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) It[3*r+c] = (r == c ? tr : 0.0) - Cc[3*r+c];
    M.volume = vol;
  } else {                                            // no usable faces: the bounding box stands in
    for (int k = 0; k < 3; k++) com[k] = 0.5 * (lo[k] + hi[k]);
    M.volume = std::max(boxvol, 1e-12);
    for (int k = 0; k < 9; k++) It[k] = 0;
    It[0] = M.volume / 12.0 * (ext[1]*ext[1] + ext[2]*ext[2]); It[4] = M.volume / 12.0 * (ext[0]*ext[0] + ext[2]*ext[2]);
    It[8] = M.volume / 12.0 * (ext[0]*ext[0] + ext[1]*ext[1]);
  }
  double R[9] = {1,0,0, 0,1,0, 0,0,1};
  if (std::fabs(It[1]) + std::fabs(It[2]) + std::fabs(It[5]) < 1e-12 * (It[0] + It[4] + It[8])) {
    M.inertia[0] = It[0]; M.inertia[1] = It[4]; M.inertia[2] = It[8];
  } else {
    double e[3], V[9]; hm::eig3(It, e, V);
    double c0[3] = {V[0], V[3], V[6]}, c1[3] = {V[1], V[4], V[7]}, c2[3];
    hm::cross(c2, c0, c1);
    const double Rm[9] = {c0[0], c1[0], c2[0], c0[1], c1[1], c2[1], c0[2], c1[2], c2[2]};
    std::memcpy(R, Rm, sizeof R);
    for (int k = 0; k < 3; k++) M.inertia[k] = e[k];
  }
  hm::mat2quat(M.quat, R);
  for (int k = 0; k < 3; k++) M.pos[k] = com[k];
  // own-frame coordinates
  for (int i = 0; i < nv; i++) {
    double d[3] = {v[3*i]-com[0], v[3*i+1]-com[1], v[3*i+2]-com[2]}, r[3];
    hm::rotvecT(r, R, d);
    v[3*i] = r[0]; v[3*i+1] = r[1]; v[3*i+2] = r[2];
  }
  // The mesh collides as its convex hull, and the hull's support mapping only ever returns hull vertices: keep the
  // vertices that are extreme along a dense direction set (Fibonacci sphere + the 26 axis/diagonal directions).  An
  // inner approximation of the hull: a hull vertex that is never selected is within a fraction of a degree of one.
  {   // STL repeats every vertex once per facet: scan each distinct point once
    std::vector<int> idx(nv);
    for (int i = 0; i < nv; i++) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int a, int c) { return std::lexicographical_compare(&v[3*a], &v[3*a] + 3, &v[3*c], &v[3*c] + 3); });
    std::vector<double> u; u.reserve(v.size());
    for (int k = 0; k < nv; k++) {
      const int i = idx[k];
      if (k > 0 && v[3*i] == v[3*idx[k-1]] && v[3*i+1] == v[3*idx[k-1]+1] && v[3*i+2] == v[3*idx[k-1]+2]) continue;
      u.push_back(v[3*i]); u.push_back(v[3*i+1]); u.push_back(v[3*i+2]);
    }
    v.swap(u); nv = (int)v.size() / 3;
  }
  std::vector<char> keep(nv, 0);
  auto support = [&](const double* dir, bool kept_only, int* arg) {
    double best = -1e300; int bi = -1;
    for (int i = 0; i < nv; i++) {
      if (kept_only && !keep[i]) continue;
      const double dp = v[3*i]*dir[0] + v[3*i+1]*dir[1] + v[3*i+2]*dir[2];
      if (dp > best) { best = dp; bi = i; }
    }
    if (arg) *arg = bi;
    return best;
  };
  auto take = [&](const double* dir) { int bi; support(dir, false, &bi); if (bi >= 0) keep[bi] = 1; };
  for (int x = -1; x <= 1; x++) for (int y = -1; y <= 1; y++) for (int z = -1; z <= 1; z++) if (x || y || z) { const double d[3] = {(double)x, (double)y, (double)z}; take(d); }
  auto fib = [](int i, int n, double phase, double* d) {
    const double z = 1.0 - 2.0 * (i + 0.5) / n, r = std::sqrt(std::max(0.0, 1.0 - z*z)), ph = i * 2.399963229728653 + phase;   // golden angle
    d[0] = r * std::cos(ph); d[1] = r * std::sin(ph); d[2] = z;
  };
  const int NDIR = 1500;
  for (int i = 0; i < NDIR; i++) { double d[3]; fib(i, NDIR, 0.0, d); take(d); }
  // ... then refined against a second, denser direction set until the kept set's support function is within 2e-4 of the
  // mesh size of the full one everywhere on it (tests/test_oracle_pinning.py checks this against the raw STL files: the
  // first set alone left 8e-3 on PR2's head_pan_L)
  {
    const double size = std::sqrt(ext[0]*ext[0] + ext[1]*ext[1] + ext[2]*ext[2]), tol = 2e-4 * size;
    const int NREF = 6000;
    std::vector<double> kv;     // kept vertices, contiguous (the inner loop of the refinement)
    for (int pass = 0; pass < 4; pass++) {
      kv.clear();
      for (int i = 0; i < nv; i++) if (keep[i]) { kv.push_back(v[3*i]); kv.push_back(v[3*i+1]); kv.push_back(v[3*i+2]); }
      int added = 0;
      for (int i = 0; i < NREF; i++) {
        double d[3]; fib(i, NREF, 1.0 + pass, d);
        double hk = -1e300;
        for (size_t k = 0; k < kv.size(); k += 3) hk = std::max(hk, kv[k]*d[0] + kv[k+1]*d[1] + kv[k+2]*d[2]);
        int bi; const double hf = support(d, false, &bi);
        if (hf - hk > tol && !keep[bi]) { keep[bi] = 1; added++; kv.push_back(v[3*bi]); kv.push_back(v[3*bi+1]); kv.push_back(v[3*bi+2]); }
      }
      if (!added) break;
    }
  }
  M.rbound = 0;
  for (int i = 0; i < nv; i++) if (keep[i]) {
    M.vert.push_back(v[3*i]); M.vert.push_back(v[3*i+1]); M.vert.push_back(v[3*i+2]);
    M.rbound = std::max(M.rbound, std::sqrt(v[3*i]*v[3*i] + v[3*i+1]*v[3*i+1] + v[3*i+2]*v[3*i+2]));
  }
  b->meshes.push_back(M);
  return (int)b->meshes.size() - 1;
}
extern "C" int mjh_builder_add_mesh(mjh_builder* b, const double* vert, int nvert, const int* face, int nface, const double scale[3]) {
  if (!b || !vert || nvert < 4 || (nface > 0 && !face)) { g_err = "add_mesh: bad arguments"; return MJH_ERR_ARG; }
  return add_mesh_impl(b, std::vector<double>(vert, vert + 3 * (size_t)nvert), face, nface, scale);
}
// text formats: ASCII STL ("vertex x y z", three per facet) and Wavefront OBJ ("v x y z", "f a b c ..." with a/b/c and
// negative indices; polygons are fanned into triangles)
static int add_mesh_text(mjh_builder* b, const std::string& text, bool obj, const char* path, const double scale[3]) {
  std::vector<double> v; std::vector<int> face;
  size_t pos = 0;
  while (pos < text.size()) {
    size_t eol = text.find('\n', pos); if (eol == std::string::npos) eol = text.size();
    std::string line = text.substr(pos, eol - pos); pos = eol + 1;
    size_t s0 = line.find_first_not_of(" \t\r");
    if (s0 == std::string::npos) continue;
    if (!obj) {
      if (line.compare(s0, 6, "vertex") == 0) {
        double x, y, z;
        if (std::sscanf(line.c_str() + s0 + 6, "%lf %lf %lf", &x, &y, &z) == 3) { v.push_back(x); v.push_back(y); v.push_back(z); }
      }
    } else if (line.compare(s0, 2, "v ") == 0) {
      double x, y, z;
      if (std::sscanf(line.c_str() + s0 + 2, "%lf %lf %lf", &x, &y, &z) == 3) { v.push_back(x); v.push_back(y); v.push_back(z); }
    } else if (line.compare(s0, 2, "f ") == 0) {
      std::vector<int> idx; const char* p = line.c_str() + s0 + 2;
      while (*p) {
        while (*p == ' ' || *p == '\t' || *p == '\r') p++;
        if (!*p) break;
        char* end; long k = std::strtol(p, &end, 10);
        if (end == p) break;
        const int nvert = (int)v.size() / 3;
        idx.push_back(k > 0 ? (int)k - 1 : nvert + (int)k);
        p = end; while (*p && *p != ' ' && *p != '\t') p++;      // skip /vt/vn
      }
      for (size_t k = 2; k < idx.size(); k++) { face.push_back(idx[0]); face.push_back(idx[k-1]); face.push_back(idx[k]); }
    }
  }
  if (!obj) for (int t = 0; t + 2 < (int)v.size() / 3; t += 3) { face.push_back(t); face.push_back(t + 1); face.push_back(t + 2); }
  if (v.size() < 12) { g_err = std::string("add_mesh: no vertices found in ") + path; return MJH_ERR_ARG; }
  const int nvert = (int)v.size() / 3;
  for (int i : face) if (i < 0 || i >= nvert) { g_err = std::string("add_mesh: face index out of range in ") + path; return MJH_ERR_ARG; }
  return add_mesh_impl(b, std::move(v), face.data(), (int)face.size() / 3, scale);
}
// mesh file: binary STL (80-byte header, uint32 triangle count, 50 bytes per triangle: normal, 3 vertices, attribute),
// ASCII STL, or Wavefront OBJ (by extension)
extern "C" int mjh_builder_add_mesh_stl(mjh_builder* b, const char* path, const double scale[3]) {
  if (!b || !path) { g_err = "add_mesh_stl: bad arguments"; return MJH_ERR_ARG; }
  FILE* f = std::fopen(path, "rb");
  if (!f) { g_err = std::string("add_mesh_stl: cannot open ") + path; return MJH_ERR_ARG; }
  {
    const std::string ps = path;
    const bool obj = ps.size() > 4 && (ps.compare(ps.size() - 4, 4, ".obj") == 0 || ps.compare(ps.size() - 4, 4, ".OBJ") == 0);
    char head5[6] = {0}; const size_t got5 = std::fread(head5, 1, 5, f);
    bool ascii = false;
    if (!obj && got5 == 5 && std::strncmp(head5, "solid", 5) == 0) {      // "solid" + a facet keyword further on = ASCII STL
      std::vector<char> probe(1024); const size_t n = std::fread(probe.data(), 1, probe.size(), f);
      ascii = std::string(probe.data(), n).find("facet") != std::string::npos;
    }
    if (obj || ascii) {
      std::fseek(f, 0, SEEK_END); const long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
      std::string text((size_t)std::max(0L, sz), '\0');
      const size_t got = std::fread(&text[0], 1, text.size(), f); std::fclose(f);
      text.resize(got);
      return add_mesh_text(b, text, obj, path, scale);
    }
    std::fseek(f, 0, SEEK_SET);
  }
  unsigned char head[84];
  unsigned ntri = 0;
  if (std::fread(head, 1, 84, f) != 84) { std::fclose(f); g_err = std::string("add_mesh_stl: short file ") + path; return MJH_ERR_ARG; }
  std::memcpy(&ntri, head + 80, 4);
  if (ntri == 0 || ntri > 5000000u) { std::fclose(f); g_err = std::string("add_mesh_stl: not a binary STL: ") + path; return MJH_ERR_ARG; }
  std::vector<unsigned char> buf((size_t)ntri * 50);
  const size_t got = std::fread(buf.data(), 1, buf.size(), f);
  std::fclose(f);
  if (got != buf.size()) { g_err = std::string("add_mesh_stl: truncated (or ASCII) STL: ") + path; return MJH_ERR_ARG; }
  std::vector<double> v((size_t)ntri * 9);
  std::vector<int> face((size_t)ntri * 3);
  for (unsigned t = 0; t < ntri; t++) {
    float xyz[9]; std::memcpy(xyz, buf.data() + (size_t)t * 50 + 12, 36);
    for (int k = 0; k < 9; k++) v[(size_t)t * 9 + k] = xyz[k];
    for (int k = 0; k < 3; k++) face[(size_t)t * 3 + k] = (int)t * 3 + k;
  }
  return add_mesh_impl(b, std::move(v), face.data(), (int)ntri, scale);
}
extern "C" int mjh_builder_add_mesh_geom(mjh_builder* b, const char* name, int body, int mesh, const double pos[3], const double quat[4],
                                         const double friction[3], int condim, int contype, int conaffinity, double density) {
  if (!b || mesh < 0 || mesh >= (int)b->meshes.size()) { g_err = "add_mesh_geom: bad mesh id"; return MJH_ERR_ARG; }
  const BMesh& M = b->meshes[mesh];
  // geom frame = (file frame in the body) o (own frame in the file frame)
  double q[4] = {1, 0, 0, 0}, p[3] = {0, 0, 0}, R[9], off[3], gq[4];
  if (quat) { for (int k = 0; k < 4; k++) q[k] = quat[k]; hm::normalize4(q); }
  if (pos) for (int k = 0; k < 3; k++) p[k] = pos[k];
  hm::quat2mat(R, q); hm::rotvec(off, R, M.pos);
  for (int k = 0; k < 3; k++) p[k] += off[k];
  hm::mulquat(gq, q, M.quat);
  const double size[3] = {M.rbound, 0, 0};
  const int g = mjh_builder_add_geom(b, name, body, MJH_GEOM_MESH, size, p, gq, friction, condim, contype, conaffinity, density);
  if (g >= 0) b->geoms[g].mesh = mesh;
  return g;
}

extern "C" int mjh_builder_add_exclude(mjh_builder* b, int b1, int b2) { b->excludes.push_back({b1, b2}); return MJH_OK; }
extern "C" int mjh_builder_add_eq_joint(mjh_builder* b, int j1, int j2, const double poly[5]) {
  BEq e; e.j1 = j1; e.j2 = j2; for (int i = 0; i < 5; i++) e.poly[i] = poly[i];
  b->eqs.push_back(e); return (int)b->eqs.size() - 1;
}

extern "C" int mjh_builder_add_eq_connect(mjh_builder* b, int body1, int body2, const double anchor[3]) {
  if (body1 <= 0 || body1 >= (int)b->bodies.size() || body2 < 0 || body2 >= (int)b->bodies.size() || body1 == body2) { g_err = "add_eq_connect: bad body"; return MJH_ERR_ARG; }
  BEq e; e.type = MJH_EQ_CONNECT; e.b1 = body1; e.b2 = body2;
  for (int i = 0; i < 3; i++) e.anchor[i] = anchor ? anchor[i] : 0;
  b->eqs.push_back(e); return (int)b->eqs.size() - 1;
}
extern "C" int mjh_builder_add_eq_weld(mjh_builder* b, int body1, int body2, const double anchor[3], double torquescale) {
  if (body1 <= 0 || body1 >= (int)b->bodies.size() || body2 < 0 || body2 >= (int)b->bodies.size() || body1 == body2) { g_err = "add_eq_weld: bad body"; return MJH_ERR_ARG; }
  BEq e; e.type = MJH_EQ_WELD; e.b1 = body1; e.b2 = body2; e.torquescale = torquescale;
  for (int i = 0; i < 3; i++) e.anchor[i] = anchor ? anchor[i] : 0;
  b->eqs.push_back(e); return (int)b->eqs.size() - 1;
}
extern "C" int mjh_builder_set_mocap(mjh_builder* b, int body) {
  if (body <= 0 || body >= (int)b->bodies.size() || b->bodies[body].parent != 0) { g_err = "set_mocap: a mocap body must be a child of the world"; return MJH_ERR_ARG; }
  for (const BJoint& j : b->joints) if (j.body == body) { g_err = "set_mocap: a mocap body cannot have joints"; return MJH_ERR_ARG; }
  for (int x : b->mocap) if (x == body) return MJH_OK;
  b->mocap.push_back(body);
  return MJH_OK;
}
extern "C" int mjh_builder_add_site(mjh_builder* b, const char* name, int body, const double pos[3], const double quat[4]) {
  if (body < 0 || body >= (int)b->bodies.size()) { g_err = "add_site: bad body"; return MJH_ERR_ARG; }
  BSite x; x.name = name ? name : ""; x.body = body;
  for (int i = 0; i < 3; i++) x.pos[i] = pos ? pos[i] : 0;
  if (quat) { for (int i = 0; i < 4; i++) x.quat[i] = quat[i]; hm::normalize4(x.quat); } else { x.quat[0] = 1; x.quat[1] = x.quat[2] = x.quat[3] = 0; }
  b->sites.push_back(x); return (int)b->sites.size() - 1;
}
extern "C" int mjh_builder_add_sensor(mjh_builder* b, const char* name, int type, int site) {
  if (site < 0 || site >= (int)b->sites.size()) { g_err = "add_sensor: bad site"; return MJH_ERR_ARG; }
  if (type != MJH_SENS_FORCE && type != MJH_SENS_TORQUE) { g_err = "add_sensor: only force and torque sensors are implemented (the types MjSim::init_sensors accepts)"; return MJH_ERR_UNSUPPORTED; }
  BSensor x; x.name = name ? name : ""; x.type = type; x.site = site;
  b->sensors.push_back(x); return (int)b->sensors.size() - 1;
}

// ---- geom mass properties (density * volume; inertia about geom centre, geom frame)
static bool geom_massprops(const mjh_builder* B, const BGeom& g, double* mass, double I[3]) {
  const double pi = 3.14159265358979323846;
  const double* s = g.size;
  switch (g.type) {
    case MJH_GEOM_SPHERE: {
      double m = g.density * 4.0 / 3.0 * pi * s[0]*s[0]*s[0];
      *mass = m; I[0] = I[1] = I[2] = 0.4 * m * s[0]*s[0]; return true; }
    case MJH_GEOM_BOX: {
      double m = g.density * 8 * s[0]*s[1]*s[2];
      *mass = m;
      I[0] = m / 3.0 * (s[1]*s[1] + s[2]*s[2]); I[1] = m / 3.0 * (s[0]*s[0] + s[2]*s[2]);
      I[2] = m / 3.0 * (s[0]*s[0] + s[1]*s[1]); return true; }
    case MJH_GEOM_CYLINDER: {
      double r = s[0], h = s[1];  // half-height
      double m = g.density * pi * r*r * 2*h;
      *mass = m; I[0] = I[1] = m * (3*r*r + 4*h*h) / 12.0; I[2] = 0.5 * m * r*r; return true; }
    case MJH_GEOM_CAPSULE: {
      double r = s[0], h = s[1];
      double mc = g.density * pi * r*r * 2*h;            // cylinder part
      double ms = g.density * 4.0 / 3.0 * pi * r*r*r;     // two hemispheres = one sphere
      *mass = mc + ms;
      double Izz = 0.5 * mc * r*r + 0.4 * ms * r*r;
      double Ixx = mc * (3*r*r + 4*h*h) / 12.0 + ms * (0.4*r*r + h*h + 0.75*r*h);
      I[0] = I[1] = Ixx; I[2] = Izz; return true; }
    case MJH_GEOM_MESH: {
      if (g.mesh < 0) return false;
      const BMesh& M = B->meshes[g.mesh];
      *mass = g.density * M.volume;
      for (int k = 0; k < 3; k++) I[k] = g.density * M.inertia[k];
      return true; }
    case MJH_GEOM_ELLIPSOID: {
      double m = g.density * 4.0 / 3.0 * pi * s[0]*s[1]*s[2];
      *mass = m;
      I[0] = 0.2 * m * (s[1]*s[1] + s[2]*s[2]); I[1] = 0.2 * m * (s[0]*s[0] + s[2]*s[2]); I[2] = 0.2 * m * (s[0]*s[0] + s[1]*s[1]); return true; }
    default: return false;  // plane etc: no mass
  }
}

static double geom_rbound(int type, const double* s) {
  switch (type) {
    case MJH_GEOM_SPHERE: return s[0];
    case MJH_GEOM_CAPSULE: return s[0] + s[1];
    case MJH_GEOM_CYLINDER: return std::sqrt(s[0]*s[0] + s[1]*s[1]);
    case MJH_GEOM_BOX: return std::sqrt(s[0]*s[0] + s[1]*s[1] + s[2]*s[2]);
    case MJH_GEOM_ELLIPSOID: return std::max(s[0], std::max(s[1], s[2]));
    case MJH_GEOM_MESH: return s[0];   // size[0] of a mesh geom = bounding radius of its vertices about the centre of mass
    default: return 0;  // plane: unbounded, handled by the pair routine
  }
}
extern "C" double mjh_geom_rbound(int type, const double* size) { return geom_rbound(type, size); }

// which geom-type pairs have a narrow-phase routine (types ordered t1<=t2)
static bool pair_supported(int t1, int t2) {
  if (t1 > t2) std::swap(t1, t2);
  auto is = [](int t, int a) { return t == a; };
  if (is(t1, MJH_GEOM_HFIELD) || is(t2, MJH_GEOM_HFIELD)) return false;
  if (is(t1, MJH_GEOM_PLANE)) return t2 != MJH_GEOM_PLANE;
  return true;   // analytic routine, or the generic convex narrow phase (cylinder-x, capsule-box, ellipsoid-x, mesh-x)
}
static int pair_maxcon(int t1, int t2) {
  if (t1 > t2) std::swap(t1, t2);
  if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_BOX) return 4;
  if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_CAPSULE) return 2;
  if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_CYLINDER) return 4;
  if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_MESH) return 4;
  if (t1 == MJH_GEOM_BOX && t2 == MJH_GEOM_BOX) return 8;
  return 1;
}

template <class T> static T* dup(const std::vector<T>& v) {
  T* p = (T*)std::calloc(v.size() + 1, sizeof(T));
  std::copy(v.begin(), v.end(), p);
  return p;
}
static char** dupnames(const std::vector<std::string>& v) {
  char** p = (char**)std::calloc(v.size() + 1, sizeof(char*));
  for (size_t i = 0; i < v.size(); i++) {
    p[i] = (char*)std::malloc(v[i].size() + 1);
    std::memcpy(p[i], v[i].c_str(), v[i].size() + 1);
  }
  return p;
}

extern "C" mjh_model* mjh_builder_compile(mjh_builder* B) {
  const int nb0 = (int)B->bodies.size();
  // ---- depth-first body order (children of a body in insertion order): trees get contiguous dofs
  std::vector<std::vector<int>> children(nb0);
  for (int i = 1; i < nb0; i++) children[B->bodies[i].parent].push_back(i);
  std::vector<int> order, stack;
  stack.push_back(0);
  while (!stack.empty()) {
    int b = stack.back(); stack.pop_back(); order.push_back(b);
    for (int k = (int)children[b].size() - 1; k >= 0; k--) stack.push_back(children[b][k]);
  }
  std::vector<int> newid(nb0);
  for (int i = 0; i < nb0; i++) newid[order[i]] = i;
  const int nbody = nb0;

  std::vector<int> body_parentid(nbody), body_rootid(nbody), body_weldid(nbody), body_jntadr(nbody), body_jntnum(nbody),
      body_dofadr(nbody), body_dofnum(nbody), body_treeid(nbody), body_level(nbody), body_geomadr(nbody), body_geomnum(nbody);
  std::vector<double> body_pos(3*nbody), body_quat(4*nbody), body_ipos(3*nbody), body_iquat(4*nbody), body_mass(nbody),
      body_inertia(3*nbody), body_gravcomp(nbody), body_invweight0(2*nbody);
  std::vector<std::string> body_names(nbody);

  // joints / geoms grouped by (new) body id, insertion order preserved
  std::vector<int> jorder, gorder;
  for (int nbid = 0; nbid < nbody; nbid++) {
    int ob = order[nbid];
    for (int j = 0; j < (int)B->joints.size(); j++) if (B->joints[j].body == ob) jorder.push_back(j);
    for (int g = 0; g < (int)B->geoms.size(); g++) if (B->geoms[g].body == ob) gorder.push_back(g);
  }
  const int njnt = (int)jorder.size(), ngeom = (int)gorder.size();
  std::vector<int> jnewid(B->joints.size());
  for (int j = 0; j < njnt; j++) jnewid[jorder[j]] = j;

  std::vector<int> jnt_type(njnt), jnt_qposadr(njnt), jnt_dofadr(njnt), jnt_bodyid(njnt), jnt_limited(njnt);
  std::vector<double> jnt_pos(3*njnt), jnt_axis(3*njnt), jnt_stiffness(njnt), jnt_range(2*njnt), jnt_margin(njnt),
      jnt_solref(2*njnt), jnt_solimp(5*njnt);
  std::vector<std::string> jnt_names(njnt);
  int nq = 0, nv = 0;
  static const int QN[4] = {7, 4, 1, 1}, VN[4] = {6, 3, 1, 1};
  for (int j = 0; j < njnt; j++) {
    const BJoint& J = B->joints[jorder[j]];
    jnt_type[j] = J.type; jnt_qposadr[j] = nq; jnt_dofadr[j] = nv; jnt_bodyid[j] = newid[J.body];
    jnt_limited[j] = J.limited ? 1 : 0;
    for (int k = 0; k < 3; k++) { jnt_pos[3*j+k] = J.pos[k]; jnt_axis[3*j+k] = J.axis[k]; }
    jnt_stiffness[j] = J.stiffness; jnt_range[2*j] = J.range[0]; jnt_range[2*j+1] = J.range[1]; jnt_margin[j] = 0;
    jnt_solref[2*j] = 0.02; jnt_solref[2*j+1] = 1;
    const double si[5] = {0.9, 0.95, 0.001, 0.5, 2};
    for (int k = 0; k < 5; k++) jnt_solimp[5*j+k] = si[k];
    jnt_names[j] = J.name;
    nq += QN[J.type]; nv += VN[J.type];
  }

  std::vector<double> qpos0(nq), qpos_spring(nq);
  std::vector<int> dof_bodyid(nv), dof_jntid(nv), dof_parentid(nv), dof_Madr(nv), dof_treeid(nv);
  std::vector<double> dof_armature(nv), dof_damping(nv), dof_frictionloss(nv), dof_invweight0(nv), dof_solref(2*nv), dof_solimp(5*nv);

  // bodies
  {
    int jcur = 0, gcur = 0;
    for (int i = 0; i < nbody; i++) {
      const BBody& X = B->bodies[order[i]];
      body_names[i] = X.name;
      body_parentid[i] = i == 0 ? 0 : newid[X.parent];
      body_level[i] = i == 0 ? 0 : body_level[body_parentid[i]] + 1;
      for (int k = 0; k < 3; k++) body_pos[3*i+k] = X.pos[k];
      for (int k = 0; k < 4; k++) body_quat[4*i+k] = X.quat[k];
      body_gravcomp[i] = X.gravcomp;
      body_jntadr[i] = jcur; int cnt = 0;
      while (jcur < njnt && jnt_bodyid[jcur] == i) { jcur++; cnt++; }
      body_jntnum[i] = cnt; if (!cnt) body_jntadr[i] = -1;
      body_dofadr[i] = cnt ? jnt_dofadr[body_jntadr[i]] : -1;
      int dn = 0; for (int j = 0; j < cnt; j++) dn += VN[jnt_type[body_jntadr[i]+j]];
      body_dofnum[i] = dn;
      body_geomadr[i] = gcur; int gc = 0;
      while (gcur < ngeom && newid[B->geoms[gorder[gcur]].body] == i) { gcur++; gc++; }
      body_geomnum[i] = gc; if (!gc) body_geomadr[i] = -1;
      body_weldid[i] = (i == 0) ? 0 : (cnt ? i : body_weldid[body_parentid[i]]);
      body_rootid[i] = (i == 0) ? 0 : (body_parentid[i] == 0 ? i : body_rootid[body_parentid[i]]);
    }
  }
  // free joint sanity: must be the only joint of a child of the world
  for (int j = 0; j < njnt; j++) if (jnt_type[j] == MJH_JNT_FREE) {
    int b = jnt_bodyid[j];
    if (body_parentid[b] != 0 || body_jntnum[b] != 1) { g_err = "free joint must be the only joint of a top-level body"; return nullptr; }
  }

  // geoms
  std::vector<int> geom_type(ngeom), geom_bodyid(ngeom), geom_condim(ngeom), geom_contype(ngeom), geom_conaffinity(ngeom), geom_priority(ngeom);
  std::vector<double> geom_pos(3*ngeom), geom_quat(4*ngeom), geom_size(3*ngeom), geom_rb(ngeom), geom_friction(3*ngeom),
      geom_solmix(ngeom), geom_solref(2*ngeom), geom_solimp(5*ngeom), geom_margin(ngeom), geom_gap(ngeom);
  std::vector<std::string> geom_names(ngeom);
  std::vector<int> geom_dataid(ngeom, -1), mesh_vertadr, mesh_vertnum;
  std::vector<double> mesh_vert;
  for (const BMesh& Mh : B->meshes) {
    mesh_vertadr.push_back((int)mesh_vert.size() / 3); mesh_vertnum.push_back((int)Mh.vert.size() / 3);
    mesh_vert.insert(mesh_vert.end(), Mh.vert.begin(), Mh.vert.end());
  }
  for (int g = 0; g < ngeom; g++) {
    const BGeom& G = B->geoms[gorder[g]];
    if (G.type == MJH_GEOM_MESH && G.mesh < 0) { g_err = "mesh geom without a mesh (use mjh_builder_add_mesh_geom)"; return nullptr; }
    geom_names[g] = G.name; geom_type[g] = G.type; geom_bodyid[g] = newid[G.body];
    geom_condim[g] = G.condim; geom_contype[g] = G.contype; geom_conaffinity[g] = G.conaffinity; geom_priority[g] = 0;
    for (int k = 0; k < 3; k++) { geom_pos[3*g+k] = G.pos[k]; geom_size[3*g+k] = G.size[k]; geom_friction[3*g+k] = G.friction[k]; }
    for (int k = 0; k < 4; k++) geom_quat[4*g+k] = G.quat[k];
    geom_rb[g] = geom_rbound(G.type, G.size);
    geom_dataid[g] = G.type == MJH_GEOM_MESH ? G.mesh : -1;
    geom_solmix[g] = 1; geom_solref[2*g] = 0.02; geom_solref[2*g+1] = 1;
    const double si[5] = {0.9, 0.95, 0.001, 0.5, 2};
    for (int k = 0; k < 5; k++) geom_solimp[5*g+k] = si[k];
    geom_margin[g] = 0; geom_gap[g] = 0;
  }

  // ---- inertial properties from geoms where not given explicitly
  for (int i = 1; i < nbody; i++) {
    const BBody& X = B->bodies[order[i]];
    if (X.explicit_inertial) {
      body_mass[i] = X.mass;
      for (int k = 0; k < 3; k++) { body_ipos[3*i+k] = X.ipos[k]; body_inertia[3*i+k] = X.inertia[k]; }
      for (int k = 0; k < 4; k++) body_iquat[4*i+k] = X.iquat[k];
      continue;
    }
    double M = 0, com[3] = {0,0,0};
    for (int g = 0; g < ngeom; g++) if (geom_bodyid[g] == i) {
      double m, I[3]; if (!geom_massprops(B, B->geoms[gorder[g]], &m, I)) continue;
      M += m; for (int k = 0; k < 3; k++) com[k] += m * geom_pos[3*g+k];
    }
    if (M <= 0) {  // massless (e.g. a frame body): leave zero; dynamics will need armature
      body_mass[i] = 0; body_iquat[4*i] = 1; continue;
    }
    for (int k = 0; k < 3; k++) com[k] /= M;
    double It[9] = {0};
    for (int g = 0; g < ngeom; g++) if (geom_bodyid[g] == i) {
      double m, I[3]; if (!geom_massprops(B, B->geoms[gorder[g]], &m, I)) continue;
      double R[9]; hm::quat2mat(R, &geom_quat[4*g]);
      // R diag(I) R^T + m (|d|^2 1 - d d^T)
      double d[3] = {geom_pos[3*g] - com[0], geom_pos[3*g+1] - com[1], geom_pos[3*g+2] - com[2]};
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
        double v = 0; for (int k = 0; k < 3; k++) v += R[3*r+k] * I[k] * R[3*c+k];
        v += m * ((r == c ? hm::dot3(d, d) : 0) - d[r]*d[c]);
        It[3*r+c] += v;
      }
    }
    body_mass[i] = M;
    for (int k = 0; k < 3; k++) body_ipos[3*i+k] = com[k];
    bool diag = std::fabs(It[1]) + std::fabs(It[2]) + std::fabs(It[5]) < 1e-14 * (It[0] + It[4] + It[8]);
    if (diag) {
      body_inertia[3*i] = It[0]; body_inertia[3*i+1] = It[4]; body_inertia[3*i+2] = It[8];
      body_iquat[4*i] = 1; body_iquat[4*i+1] = body_iquat[4*i+2] = body_iquat[4*i+3] = 0;
    } else {
      double e[3], V[9]; hm::eig3(It, e, V);
      // make V a proper rotation
      double c0[3] = {V[0], V[3], V[6]}, c1[3] = {V[1], V[4], V[7]}, c2[3];
      hm::cross(c2, c0, c1);
      double Rm[9] = {c0[0], c1[0], c2[0], c0[1], c1[1], c2[1], c0[2], c1[2], c2[2]};
      hm::mat2quat(&body_iquat[4*i], Rm);
      for (int k = 0; k < 3; k++) body_inertia[3*i+k] = e[k];
    }
  }
  body_iquat[0] = 1;

  // <compiler balanceinertia>: a diagonal inertia that violates the triangle inequality A + B >= C (URDF exports do) is
  // replaced by its mean on all three axes [UPSTREAM mjCompiler]
  if (B->balanceinertia) for (int i = 1; i < nbody; i++) {
    double* I = &body_inertia[3*i];
    if (I[0] + I[1] < I[2] || I[0] + I[2] < I[1] || I[1] + I[2] < I[0]) I[0] = I[1] = I[2] = (I[0] + I[1] + I[2]) / 3.0;
  }
  // lower bounds on mass / inertia of every body except the world (mjCompiler boundmass / boundinertia)
  for (int i = 1; i < nbody; i++) {
    if (B->boundmass > 0 && body_mass[i] < B->boundmass) body_mass[i] = B->boundmass;
    if (B->boundinertia > 0) for (int k = 0; k < 3; k++) if (body_inertia[3*i+k] < B->boundinertia) body_inertia[3*i+k] = B->boundinertia;
  }
  // ---- dofs
  for (int j = 0; j < njnt; j++) {
    const BJoint& J = B->joints[jorder[j]];
    int b = jnt_bodyid[j];
    for (int k = 0; k < VN[J.type]; k++) {
      int d = jnt_dofadr[j] + k;
      dof_bodyid[d] = b; dof_jntid[d] = j;
      dof_armature[d] = J.armature; dof_damping[d] = J.damping; dof_frictionloss[d] = J.frictionloss;
      dof_solref[2*d] = 0.02; dof_solref[2*d+1] = 1;
      const double si[5] = {0.9, 0.95, 0.001, 0.5, 2};
      for (int q = 0; q < 5; q++) dof_solimp[5*d+q] = si[q];
      if (d > body_dofadr[b]) dof_parentid[d] = d - 1;
      else {
        int p = body_parentid[b];
        while (p > 0 && body_dofnum[p] == 0) p = body_parentid[p];
        dof_parentid[d] = (p > 0) ? body_dofadr[p] + body_dofnum[p] - 1 : -1;
      }
    }
    // qpos0
    int qa = jnt_qposadr[j];
    if (J.type == MJH_JNT_FREE) {
      for (int k = 0; k < 3; k++) qpos0[qa+k] = body_pos[3*b+k];
      for (int k = 0; k < 4; k++) qpos0[qa+3+k] = body_quat[4*b+k];
    } else if (J.type == MJH_JNT_BALL) {
      qpos0[qa] = 1; qpos0[qa+1] = qpos0[qa+2] = qpos0[qa+3] = 0;
    } else qpos0[qa] = J.ref;
    for (int k = 0; k < QN[J.type]; k++) qpos_spring[qa+k] = qpos0[qa+k];
  }
  int nM = 0;
  for (int d = 0; d < nv; d++) {
    dof_Madr[d] = nM;
    int p = d; while (p >= 0) { nM++; p = dof_parentid[p]; }
  }
  // trees
  std::vector<int> tree_dofadr, tree_dofnum, tree_bodyid;
  for (int i = 0; i < nbody; i++) body_treeid[i] = -1;
  for (int i = 1; i < nbody; i++) {
    if (body_parentid[i] == 0) {
      // count dofs in subtree (contiguous thanks to DFS order)
      int first = -1, cnt = 0, last = i;
      for (int k = i; k < nbody; k++) {
        if (k > i && body_rootid[k] != i) break;
        last = k;
        if (body_dofnum[k]) { if (first < 0) first = body_dofadr[k]; cnt += body_dofnum[k]; }
      }
      if (cnt) {
        int t = (int)tree_dofadr.size();
        tree_dofadr.push_back(first); tree_dofnum.push_back(cnt); tree_bodyid.push_back(i);
        for (int k = i; k <= last; k++) body_treeid[k] = t;
      }
    }
  }
  for (int d = 0; d < nv; d++) dof_treeid[d] = body_treeid[dof_bodyid[d]];
  const int ntree = (int)tree_dofadr.size();

  // ---- reference configuration kinematics + Jacobian-based mass matrix (fp64, dense)
  std::vector<double> xpos(3*nbody), xquat(4*nbody), xmat(9*nbody), xipos(3*nbody), ximat(9*nbody);
  xquat[0] = 1; hm::quat2mat(&xmat[0], &xquat[0]); hm::quat2mat(&ximat[0], &xquat[0]);
  for (int i = 1; i < nbody; i++) {
    int p = body_parentid[i];
    double t[3]; hm::rotvec(t, &xmat[9*p], &body_pos[3*i]);
    for (int k = 0; k < 3; k++) xpos[3*i+k] = xpos[3*p+k] + t[k];
    hm::mulquat(&xquat[4*i], &xquat[4*p], &body_quat[4*i]);
    hm::normalize4(&xquat[4*i]);
    hm::quat2mat(&xmat[9*i], &xquat[4*i]);
    hm::rotvec(t, &xmat[9*i], &body_ipos[3*i]);
    for (int k = 0; k < 3; k++) xipos[3*i+k] = xpos[3*i+k] + t[k];
    double qi[4]; hm::mulquat(qi, &xquat[4*i], &body_iquat[4*i]); hm::quat2mat(&ximat[9*i], qi);
  }
  // world-frame Jacobian columns of a point attached to body b: jp[3*nv], jr[3*nv] (row-major 3 x nv)
  auto jac = [&](int b, const double* point, std::vector<double>& jp, std::vector<double>& jr) {
    std::fill(jp.begin(), jp.end(), 0.0); std::fill(jr.begin(), jr.end(), 0.0);
    while (b > 0) {
      for (int j = body_jntadr[b]; j >= 0 && j < body_jntadr[b] + body_jntnum[b]; j++) {
        int da = jnt_dofadr[j];
        double anchor[3], t[3];
        hm::rotvec(t, &xmat[9*b], &jnt_pos[3*j]);
        for (int k = 0; k < 3; k++) anchor[k] = xpos[3*b+k] + t[k];
        double r[3] = {point[0] - anchor[0], point[1] - anchor[1], point[2] - anchor[2]};
        if (jnt_type[j] == MJH_JNT_FREE) {
          for (int k = 0; k < 3; k++) jp[k*nv + da + k] = 1;
          da += 3;
        }
        if (jnt_type[j] == MJH_JNT_FREE || jnt_type[j] == MJH_JNT_BALL) {
          for (int a = 0; a < 3; a++) {
            double ax[3] = {xmat[9*b + a], xmat[9*b + 3 + a], xmat[9*b + 6 + a]};
            double c[3]; hm::cross(c, ax, r);
            for (int k = 0; k < 3; k++) { jr[k*nv + da + a] = ax[k]; jp[k*nv + da + a] = c[k]; }
          }
        } else {
          double ax[3]; hm::rotvec(ax, &xmat[9*b], &jnt_axis[3*j]);
          if (jnt_type[j] == MJH_JNT_HINGE) {
            double c[3]; hm::cross(c, ax, r);
            for (int k = 0; k < 3; k++) { jr[k*nv + da] = ax[k]; jp[k*nv + da] = c[k]; }
          } else for (int k = 0; k < 3; k++) jp[k*nv + da] = ax[k];
        }
      }
      b = body_parentid[b];
    }
  };
  std::vector<double> Mq((size_t)nv * nv, 0.0), jp(3*(size_t)nv), jr(3*(size_t)nv);
  for (int b = 1; b < nbody; b++) {
    if (body_mass[b] <= 0 && body_inertia[3*b] <= 0) continue;
    jac(b, &xipos[3*b], jp, jr);
    // world inertia
    double Iw[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
      double v = 0; for (int k = 0; k < 3; k++) v += ximat[9*b + 3*r + k] * body_inertia[3*b + k] * ximat[9*b + 3*c + k];
      Iw[3*r+c] = v;
    }
    // support = dofs of this body's chain: only nonzero columns matter
    std::vector<int> sup;
    for (int d = 0; d < nv; d++) {
      bool nz = false; for (int k = 0; k < 3; k++) if (jp[k*nv+d] != 0 || jr[k*nv+d] != 0) nz = true;
      if (nz) sup.push_back(d);
    }
    for (int a : sup) for (int c : sup) {
      double v = 0;
      for (int k = 0; k < 3; k++) v += body_mass[b] * jp[k*nv+a] * jp[k*nv+c];
      for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) v += jr[r*nv+a] * Iw[3*r+k] * jr[k*nv+c];
      Mq[(size_t)a*nv + c] += v;
    }
  }
  for (int d = 0; d < nv; d++) Mq[(size_t)d*nv + d] += dof_armature[d];
  double meaninertia = 0;
  for (int d = 0; d < nv; d++) meaninertia += Mq[(size_t)d*nv + d];
  meaninertia = nv ? meaninertia / nv : 1.0;
  if (meaninertia < 1e-15) meaninertia = 1e-15;
  // dense Cholesky M = L L^T (trees are independent blocks but dense is fine on the host)
  std::vector<double> L(Mq);
  bool spd = true;
  for (int c = 0; c < nv && spd; c++) {
    double s = L[(size_t)c*nv + c];
    for (int k = 0; k < c; k++) s -= L[(size_t)c*nv + k] * L[(size_t)c*nv + k];
    if (s <= 1e-300) { spd = false; break; }
    double dd = std::sqrt(s); L[(size_t)c*nv + c] = dd;
    for (int r = c + 1; r < nv; r++) {
      double t = L[(size_t)r*nv + c];
      if (t == 0) { bool any = false; for (int k = 0; k < c; k++) if (L[(size_t)r*nv+k] != 0 && L[(size_t)c*nv+k] != 0) { any = true; break; } if (!any) continue; }
      for (int k = 0; k < c; k++) t -= L[(size_t)r*nv + k] * L[(size_t)c*nv + k];
      L[(size_t)r*nv + c] = t / dd;
    }
  }
  if (!spd) { g_err = "mass matrix at qpos0 is not positive definite (massless body without armature?)"; return nullptr; }
  auto solveM = [&](std::vector<double>& x) {  // in-place M^{-1} x
    for (int r = 0; r < nv; r++) { double s = x[r]; for (int k = 0; k < r; k++) s -= L[(size_t)r*nv+k] * x[k]; x[r] = s / L[(size_t)r*nv+r]; }
    for (int r = nv - 1; r >= 0; r--) { double s = x[r]; for (int k = r + 1; k < nv; k++) s -= L[(size_t)k*nv+r] * x[k]; x[r] = s / L[(size_t)r*nv+r]; }
  };
  // dof_invweight0: diag(M^-1), averaged within ball / free-translational / free-rotational triples
  {
    std::vector<double> dinv(nv), e(nv);
    for (int d = 0; d < nv; d++) { std::fill(e.begin(), e.end(), 0.0); e[d] = 1; solveM(e); dinv[d] = e[d]; }
    for (int j = 0; j < njnt; j++) {
      int da = jnt_dofadr[j];
      if (jnt_type[j] == MJH_JNT_FREE) {
        double a = (dinv[da] + dinv[da+1] + dinv[da+2]) / 3, r = (dinv[da+3] + dinv[da+4] + dinv[da+5]) / 3;
        for (int k = 0; k < 3; k++) { dof_invweight0[da+k] = a; dof_invweight0[da+3+k] = r; }
      } else if (jnt_type[j] == MJH_JNT_BALL) {
        double a = (dinv[da] + dinv[da+1] + dinv[da+2]) / 3;
        for (int k = 0; k < 3; k++) dof_invweight0[da+k] = a;
      } else dof_invweight0[da] = dinv[da];
    }
  }
  // body_invweight0: (1/3) trace(Jp M^-1 Jp^T), (1/3) trace(Jr M^-1 Jr^T) at the body COM
  for (int b = 1; b < nbody; b++) {
    if (body_weldid[b] == 0) { body_invweight0[2*b] = body_invweight0[2*b+1] = 0; continue; }
    jac(b, &xipos[3*b], jp, jr);
    double tr = 0, rr = 0;
    std::vector<double> x(nv);
    for (int k = 0; k < 3; k++) {
      for (int d = 0; d < nv; d++) x[d] = jp[k*nv+d];
      solveM(x); for (int d = 0; d < nv; d++) tr += jp[k*nv+d] * x[d];
      for (int d = 0; d < nv; d++) x[d] = jr[k*nv+d];
      solveM(x); for (int d = 0; d < nv; d++) rr += jr[k*nv+d] * x[d];
    }
    body_invweight0[2*b] = tr / 3; body_invweight0[2*b+1] = rr / 3;
  }

  // ---- static candidate pair list
  struct P { int b1, b2, g1, g2; };
  std::vector<P> pairs;
  auto excluded = [&](int b1, int b2) {
    for (auto& e : B->excludes) { int x = newid[e.first], y = newid[e.second]; if ((x == b1 && y == b2) || (x == b2 && y == b1)) return true; }
    return false;
  };
  const bool filterparent = !(B->opt.disableflags & MJH_DSBL_FILTERPARENT);
  for (int g1 = 0; g1 < ngeom; g1++) for (int g2 = g1 + 1; g2 < ngeom; g2++) {
    int b1 = geom_bodyid[g1], b2 = geom_bodyid[g2];
    int w1 = body_weldid[b1], w2 = body_weldid[b2];
    if (w1 == w2) continue;  // same body / welded together / both static
    int wp1 = body_weldid[body_parentid[w1]], wp2 = body_weldid[body_parentid[w2]];
    if (filterparent && w1 != 0 && w2 != 0 && (w1 == wp2 || w2 == wp1)) continue;
    if (!((geom_contype[g1] & geom_conaffinity[g2]) || (geom_contype[g2] & geom_conaffinity[g1]))) continue;
    if (excluded(b1, b2)) continue;
    if (!pair_supported(geom_type[g1], geom_type[g2])) continue;
    P p; p.b1 = std::min(b1, b2); p.b2 = std::max(b1, b2);
    if (geom_type[g1] <= geom_type[g2]) { p.g1 = g1; p.g2 = g2; } else { p.g1 = g2; p.g2 = g1; }
    pairs.push_back(p);
  }
  std::stable_sort(pairs.begin(), pairs.end(), [](const P& a, const P& b) {
    if (a.b1 != b.b1) return a.b1 < b.b1; if (a.b2 != b.b2) return a.b2 < b.b2; return false; });
  const int npair = (int)pairs.size();
  std::vector<int> pair_geom1(npair), pair_geom2(npair);
  int capcon = 0, caprow = 0;
  for (int i = 0; i < npair; i++) {
    pair_geom1[i] = pairs[i].g1; pair_geom2[i] = pairs[i].g2;
    int mc = pair_maxcon(geom_type[pairs[i].g1], geom_type[pairs[i].g2]);
    int dim = std::max(geom_condim[pairs[i].g1], geom_condim[pairs[i].g2]);
    capcon += mc; caprow += mc * (dim == 1 ? 1 : 2 * (dim - 1));
  }
  // equality
  const int neq = (int)B->eqs.size();
  std::vector<int> eq_type(neq), eq_obj1id(neq), eq_obj2id(neq), eq_active(neq);
  std::vector<double> eq_data(11*neq), eq_solref(2*neq), eq_solimp(5*neq);
  for (int e = 0; e < neq; e++) {
    const BEq& E = B->eqs[e];
    eq_type[e] = E.type; eq_active[e] = 1;
    if (E.type == MJH_EQ_JOINT) {
      eq_obj1id[e] = jnewid[E.j1]; eq_obj2id[e] = E.j2 >= 0 ? jnewid[E.j2] : -1;
      for (int k = 0; k < 5; k++) eq_data[11*e+k] = E.poly[k];
    } else {
      // connect / weld between bodies: the second anchor and the relative orientation are those of the reference configuration
      const int b1 = newid[E.b1], b2 = newid[E.b2];
      eq_obj1id[e] = b1; eq_obj2id[e] = b2;
      const int ba = E.type == MJH_EQ_CONNECT ? b1 : b2, bo = E.type == MJH_EQ_CONNECT ? b2 : b1;   // body the anchor is given in / the other one
      double w[3], t[3], o[3];
      hm::rotvec(t, &xmat[9*ba], E.anchor);
      for (int k = 0; k < 3; k++) w[k] = xpos[3*ba+k] + t[k] - xpos[3*bo+k];
      hm::rotvecT(o, &xmat[9*bo], w);
      for (int k = 0; k < 3; k++) { eq_data[11*e+k] = E.anchor[k]; eq_data[11*e+3+k] = o[k]; }
      double qi[4] = {xquat[4*b2], -xquat[4*b2+1], -xquat[4*b2+2], -xquat[4*b2+3]};
      hm::mulquat(&eq_data[11*e+6], qi, &xquat[4*b1]);
      eq_data[11*e+10] = E.torquescale;
    }
    eq_solref[2*e] = 0.02; eq_solref[2*e+1] = 1;
    const double si[5] = {0.9, 0.95, 0.001, 0.5, 2};
    for (int k = 0; k < 5; k++) eq_solimp[5*e+k] = si[k];
  }
  int nlimit = 0, nfl = 0, neqrow = 0;
  for (int e = 0; e < neq; e++) neqrow += eq_type[e] == MJH_EQ_WELD ? 6 : (eq_type[e] == MJH_EQ_CONNECT ? 3 : 1);
  for (int j = 0; j < njnt; j++) if (jnt_limited[j]) nlimit++;
  for (int d = 0; d < nv; d++) if (dof_frictionloss[d] > 0) nfl++;

  mjh_model* m = (mjh_model*)std::calloc(1, sizeof(mjh_model));
  m->nq = nq; m->nv = nv; m->nbody = nbody; m->njnt = njnt; m->ngeom = ngeom; m->neq = neq; m->npair = npair;
  m->nM = nM; m->ntree = ntree; m->nexclude = (int)B->excludes.size();
  m->maxcon = B->maxcon > 0 ? B->maxcon : capcon;
  m->maxefc = B->maxefc > 0 ? B->maxefc : (caprow + neqrow + nlimit + nfl);
  if (B->maxcon > 0 && B->maxefc <= 0) {
    // rows for the capped contact count at the worst-case dim present in the pair list
    int maxrows_per_con = 1;
    for (int i = 0; i < npair; i++) { int dim = std::max(geom_condim[pair_geom1[i]], geom_condim[pair_geom2[i]]); maxrows_per_con = std::max(maxrows_per_con, dim == 1 ? 1 : 2*(dim-1)); }
    m->maxefc = std::min(caprow, B->maxcon * maxrows_per_con) + neqrow + nlimit + nfl;
  }
  m->opt = B->opt; m->meaninertia = meaninertia;
#define SETI(f) m->f = dup(f)
  SETI(body_parentid); SETI(body_rootid); SETI(body_weldid); SETI(body_jntadr); SETI(body_jntnum); SETI(body_dofadr);
  SETI(body_dofnum); SETI(body_treeid); SETI(body_level); SETI(body_geomadr); SETI(body_geomnum);
  SETI(body_pos); SETI(body_quat); SETI(body_ipos); SETI(body_iquat); SETI(body_mass); SETI(body_inertia);
  SETI(body_gravcomp); SETI(body_invweight0);
  SETI(jnt_type); SETI(jnt_qposadr); SETI(jnt_dofadr); SETI(jnt_bodyid); SETI(jnt_limited); SETI(jnt_pos); SETI(jnt_axis);
  SETI(jnt_stiffness); SETI(jnt_range); SETI(jnt_margin); SETI(jnt_solref); SETI(jnt_solimp); SETI(qpos0); SETI(qpos_spring);
  SETI(dof_bodyid); SETI(dof_jntid); SETI(dof_parentid); SETI(dof_Madr); SETI(dof_treeid); SETI(dof_armature); SETI(dof_damping);
  SETI(dof_frictionloss); SETI(dof_invweight0); SETI(dof_solref); SETI(dof_solimp);
  SETI(tree_dofadr); SETI(tree_dofnum); SETI(tree_bodyid);
  SETI(geom_type); SETI(geom_bodyid); SETI(geom_condim); SETI(geom_contype); SETI(geom_conaffinity); SETI(geom_priority);
  SETI(geom_pos); SETI(geom_quat); SETI(geom_size); m->geom_rbound = dup(geom_rb); SETI(geom_friction); SETI(geom_solmix);
  SETI(geom_solref); SETI(geom_solimp); SETI(geom_margin); SETI(geom_gap);
  SETI(pair_geom1); SETI(pair_geom2);
  SETI(eq_type); SETI(eq_obj1id); SETI(eq_obj2id); SETI(eq_active); SETI(eq_data); SETI(eq_solref); SETI(eq_solimp);
  m->nmesh = (int)mesh_vertadr.size(); m->nmeshvert = (int)mesh_vert.size() / 3;
  SETI(geom_dataid); SETI(mesh_vertadr); SETI(mesh_vertnum); SETI(mesh_vert);
#undef SETI
  m->body_names = dupnames(body_names); m->jnt_names = dupnames(jnt_names); m->geom_names = dupnames(geom_names);
  {   // sites, force / torque sensors, mocap bodies
    const int nsite = (int)B->sites.size(), nsensor = (int)B->sensors.size();
    std::vector<int> site_bodyid(nsite), sensor_type(nsensor), sensor_objid(nsensor), sensor_adr(nsensor), body_mocapid(nbody, -1);
    std::vector<double> site_pos(3*nsite), site_quat(4*nsite);
    std::vector<std::string> site_names(nsite), sensor_names(nsensor);
    for (int i = 0; i < nsite; i++) {
      const BSite& S = B->sites[i];
      site_bodyid[i] = newid[S.body]; site_names[i] = S.name;
      for (int k = 0; k < 3; k++) site_pos[3*i+k] = S.pos[k];
      for (int k = 0; k < 4; k++) site_quat[4*i+k] = S.quat[k];
    }
    for (int i = 0; i < nsensor; i++) { sensor_type[i] = B->sensors[i].type; sensor_objid[i] = B->sensors[i].site; sensor_adr[i] = 3 * i; sensor_names[i] = B->sensors[i].name; }
    int nmocap = 0;
    for (int i = 1; i < nbody; i++) for (int ob : B->mocap) if (newid[ob] == i) body_mocapid[i] = nmocap++;   // numbered in body order
    m->nsite = nsite; m->nsensor = nsensor; m->nsensordata = 3 * nsensor; m->nmocap = nmocap;
    m->site_bodyid = dup(site_bodyid); m->site_pos = dup(site_pos); m->site_quat = dup(site_quat);
    m->sensor_type = dup(sensor_type); m->sensor_objid = dup(sensor_objid); m->sensor_adr = dup(sensor_adr);
    m->body_mocapid = dup(body_mocapid);
    m->site_names = dupnames(site_names); m->sensor_names = dupnames(sensor_names);
  }
  return m;
}

// ---- sub-wave packing: `copies` instances of the moving bodies in one model (include/mjhip.h)
extern "C" mjh_model* mjh_model_replicate(const mjh_model* a, int G) {
  if (!a || G < 1) { g_err = "mjh_model_replicate: bad arguments"; return nullptr; }
  if (a->nsite || a->nsensor || a->nmocap) { g_err = "mjh_model_replicate: models with sites, sensors or mocap bodies are not packed"; return nullptr; }
  for (int e = 0; e < a->neq; e++) if (a->eq_type[e] != MJH_EQ_JOINT) { g_err = "mjh_model_replicate: connect / weld equalities are not packed"; return nullptr; }
  const int nb = a->nbody, nj = a->njnt, nv = a->nv, nq = a->nq, ng = a->ngeom, nt = a->ntree, ne = a->neq, np = a->npair, nM = a->nM;
  std::vector<int> moving_b, moving_g;
  for (int b = 1; b < nb; b++) if (a->body_weldid[b] != 0) moving_b.push_back(b);
  for (int g = 0; g < ng; g++) if (a->body_weldid[a->geom_bodyid[g]] != 0) moving_g.push_back(g);
  const int mb = (int)moving_b.size(), mg = (int)moving_g.size();
  const int NB = nb + (G - 1) * mb, NG = ng + (G - 1) * mg;
  if (NB > 4096 || (long long)G * nv > 4096) { g_err = "mjh_model_replicate: result too large"; return nullptr; }
  // id maps of copy k
  auto bmap = [&](int k, int b) { if (k == 0 || a->body_weldid[b] == 0) return b; int r = (int)(std::lower_bound(moving_b.begin(), moving_b.end(), b) - moving_b.begin()); return nb + (k - 1) * mb + r; };
  auto gmap = [&](int k, int g) { if (k == 0 || a->body_weldid[a->geom_bodyid[g]] == 0) return g; int r = (int)(std::lower_bound(moving_g.begin(), moving_g.end(), g) - moving_g.begin()); return ng + (k - 1) * mg + r; };
  mjh_model* m = (mjh_model*)std::calloc(1, sizeof(mjh_model));
  m->nq = G * nq; m->nv = G * nv; m->nbody = NB; m->njnt = G * nj; m->ngeom = NG; m->neq = G * ne; m->npair = G * np; m->nM = G * nM;
  m->ntree = G * nt; m->nexclude = a->nexclude; m->maxcon = G * a->maxcon; m->maxefc = G * a->maxefc;
  m->nmesh = a->nmesh; m->nmeshvert = a->nmeshvert; m->opt = a->opt; m->meaninertia = a->meaninertia;
  auto ialloc = [](size_t n) { return (int*)std::calloc(n + 1, sizeof(int)); };
  auto dalloc = [](size_t n) { return (double*)std::calloc(n + 1, sizeof(double)); };
  // ---- bodies
#define BI(f) m->f = ialloc(NB)
#define BD(f, w) m->f = dalloc((size_t)(w) * NB)
  BI(body_parentid); BI(body_rootid); BI(body_weldid); BI(body_jntadr); BI(body_jntnum); BI(body_dofadr); BI(body_dofnum);
  BI(body_treeid); BI(body_level); BI(body_geomadr); BI(body_geomnum);
  BD(body_pos, 3); BD(body_quat, 4); BD(body_ipos, 3); BD(body_iquat, 4); BD(body_mass, 1); BD(body_inertia, 3); BD(body_gravcomp, 1); BD(body_invweight0, 2);
#undef BI
#undef BD
  std::vector<std::string> bnames(NB), jnames((size_t)G * nj), gnames(NG);
  auto nm = [](char** t, int i, int k) { std::string s = (t && t[i]) ? t[i] : ""; if (k > 0 && !s.empty()) s += "#" + std::to_string(k); return s; };
  for (int k = 0; k < G; k++) for (int b = 0; b < nb; b++) {
    if (k > 0 && a->body_weldid[b] == 0) continue;
    const int B2 = bmap(k, b);
    m->body_parentid[B2] = bmap(k, a->body_parentid[b]); m->body_rootid[B2] = bmap(k, a->body_rootid[b]); m->body_weldid[B2] = bmap(k, a->body_weldid[b]);
    m->body_jntadr[B2] = a->body_jntadr[b] >= 0 ? a->body_jntadr[b] + k * nj : -1; m->body_jntnum[B2] = a->body_jntnum[b];
    m->body_dofadr[B2] = a->body_dofadr[b] >= 0 ? a->body_dofadr[b] + k * nv : -1; m->body_dofnum[B2] = a->body_dofnum[b];
    m->body_treeid[B2] = a->body_treeid[b] >= 0 ? a->body_treeid[b] + k * nt : -1; m->body_level[B2] = a->body_level[b];
    m->body_geomnum[B2] = a->body_geomnum[b];
    m->body_geomadr[B2] = a->body_geomadr[b] >= 0 ? gmap(k, a->body_geomadr[b]) : -1;
    for (int q = 0; q < 3; q++) { m->body_pos[3*B2+q] = a->body_pos[3*b+q]; m->body_ipos[3*B2+q] = a->body_ipos[3*b+q]; m->body_inertia[3*B2+q] = a->body_inertia[3*b+q]; }
    for (int q = 0; q < 4; q++) { m->body_quat[4*B2+q] = a->body_quat[4*b+q]; m->body_iquat[4*B2+q] = a->body_iquat[4*b+q]; }
    m->body_mass[B2] = a->body_mass[b]; m->body_gravcomp[B2] = a->body_gravcomp[b];
    m->body_invweight0[2*B2] = a->body_invweight0[2*b]; m->body_invweight0[2*B2+1] = a->body_invweight0[2*b+1];
    bnames[B2] = nm(a->body_names, b, k);
  }
  // ---- joints, dofs, trees, qpos0: whole blocks per copy
  m->jnt_type = ialloc((size_t)G*nj); m->jnt_qposadr = ialloc((size_t)G*nj); m->jnt_dofadr = ialloc((size_t)G*nj); m->jnt_bodyid = ialloc((size_t)G*nj); m->jnt_limited = ialloc((size_t)G*nj);
  m->jnt_pos = dalloc((size_t)3*G*nj); m->jnt_axis = dalloc((size_t)3*G*nj); m->jnt_stiffness = dalloc((size_t)G*nj); m->jnt_range = dalloc((size_t)2*G*nj);
  m->jnt_margin = dalloc((size_t)G*nj); m->jnt_solref = dalloc((size_t)2*G*nj); m->jnt_solimp = dalloc((size_t)5*G*nj);
  m->qpos0 = dalloc((size_t)G*nq); m->qpos_spring = dalloc((size_t)G*nq);
  m->dof_bodyid = ialloc((size_t)G*nv); m->dof_jntid = ialloc((size_t)G*nv); m->dof_parentid = ialloc((size_t)G*nv); m->dof_Madr = ialloc((size_t)G*nv); m->dof_treeid = ialloc((size_t)G*nv);
  m->dof_armature = dalloc((size_t)G*nv); m->dof_damping = dalloc((size_t)G*nv); m->dof_frictionloss = dalloc((size_t)G*nv); m->dof_invweight0 = dalloc((size_t)G*nv);
  m->dof_solref = dalloc((size_t)2*G*nv); m->dof_solimp = dalloc((size_t)5*G*nv);
  m->tree_dofadr = ialloc((size_t)G*nt); m->tree_dofnum = ialloc((size_t)G*nt); m->tree_bodyid = ialloc((size_t)G*nt);
  auto cpd = [](double* dst, const double* src, size_t n) { if (n) std::memcpy(dst, src, n * sizeof(double)); };
  for (int k = 0; k < G; k++) {
    for (int j = 0; j < nj; j++) {
      const int J2 = j + k * nj;
      m->jnt_type[J2] = a->jnt_type[j]; m->jnt_qposadr[J2] = a->jnt_qposadr[j] + k * nq; m->jnt_dofadr[J2] = a->jnt_dofadr[j] + k * nv;
      m->jnt_bodyid[J2] = bmap(k, a->jnt_bodyid[j]); m->jnt_limited[J2] = a->jnt_limited[j];
      jnames[J2] = nm(a->jnt_names, j, k);
    }
    cpd(m->jnt_pos + (size_t)3*k*nj, a->jnt_pos, (size_t)3*nj); cpd(m->jnt_axis + (size_t)3*k*nj, a->jnt_axis, (size_t)3*nj);
    cpd(m->jnt_stiffness + (size_t)k*nj, a->jnt_stiffness, nj); cpd(m->jnt_range + (size_t)2*k*nj, a->jnt_range, (size_t)2*nj);
    cpd(m->jnt_margin + (size_t)k*nj, a->jnt_margin, nj); cpd(m->jnt_solref + (size_t)2*k*nj, a->jnt_solref, (size_t)2*nj); cpd(m->jnt_solimp + (size_t)5*k*nj, a->jnt_solimp, (size_t)5*nj);
    cpd(m->qpos0 + (size_t)k*nq, a->qpos0, nq); cpd(m->qpos_spring + (size_t)k*nq, a->qpos_spring, nq);
    for (int d = 0; d < nv; d++) {
      const int D2 = d + k * nv;
      m->dof_bodyid[D2] = bmap(k, a->dof_bodyid[d]); m->dof_jntid[D2] = a->dof_jntid[d] + k * nj;
      m->dof_parentid[D2] = a->dof_parentid[d] >= 0 ? a->dof_parentid[d] + k * nv : -1;
      m->dof_Madr[D2] = a->dof_Madr[d] + k * nM; m->dof_treeid[D2] = a->dof_treeid[d] >= 0 ? a->dof_treeid[d] + k * nt : -1;
    }
    cpd(m->dof_armature + (size_t)k*nv, a->dof_armature, nv); cpd(m->dof_damping + (size_t)k*nv, a->dof_damping, nv);
    cpd(m->dof_frictionloss + (size_t)k*nv, a->dof_frictionloss, nv); cpd(m->dof_invweight0 + (size_t)k*nv, a->dof_invweight0, nv);
    cpd(m->dof_solref + (size_t)2*k*nv, a->dof_solref, (size_t)2*nv); cpd(m->dof_solimp + (size_t)5*k*nv, a->dof_solimp, (size_t)5*nv);
    for (int t = 0; t < nt; t++) { m->tree_dofadr[t + k*nt] = a->tree_dofadr[t] + k * nv; m->tree_dofnum[t + k*nt] = a->tree_dofnum[t]; m->tree_bodyid[t + k*nt] = bmap(k, a->tree_bodyid[t]); }
  }
  // ---- geoms
  m->geom_type = ialloc(NG); m->geom_bodyid = ialloc(NG); m->geom_condim = ialloc(NG); m->geom_contype = ialloc(NG); m->geom_conaffinity = ialloc(NG);
  m->geom_priority = ialloc(NG); m->geom_dataid = ialloc(NG);
  m->geom_pos = dalloc((size_t)3*NG); m->geom_quat = dalloc((size_t)4*NG); m->geom_size = dalloc((size_t)3*NG); m->geom_rbound = dalloc(NG); m->geom_friction = dalloc((size_t)3*NG);
  m->geom_solmix = dalloc(NG); m->geom_solref = dalloc((size_t)2*NG); m->geom_solimp = dalloc((size_t)5*NG); m->geom_margin = dalloc(NG); m->geom_gap = dalloc(NG);
  for (int k = 0; k < G; k++) for (int g = 0; g < ng; g++) {
    if (k > 0 && a->body_weldid[a->geom_bodyid[g]] == 0) continue;
    const int G2 = gmap(k, g);
    m->geom_type[G2] = a->geom_type[g]; m->geom_bodyid[G2] = bmap(k, a->geom_bodyid[g]); m->geom_condim[G2] = a->geom_condim[g];
    m->geom_contype[G2] = a->geom_contype[g]; m->geom_conaffinity[G2] = a->geom_conaffinity[g]; m->geom_priority[G2] = a->geom_priority[g];
    m->geom_dataid[G2] = a->geom_dataid ? a->geom_dataid[g] : -1;
    for (int q = 0; q < 3; q++) { m->geom_pos[3*G2+q] = a->geom_pos[3*g+q]; m->geom_size[3*G2+q] = a->geom_size[3*g+q]; m->geom_friction[3*G2+q] = a->geom_friction[3*g+q]; }
    for (int q = 0; q < 4; q++) m->geom_quat[4*G2+q] = a->geom_quat[4*g+q];
    m->geom_rbound[G2] = a->geom_rbound[g]; m->geom_solmix[G2] = a->geom_solmix[g]; m->geom_margin[G2] = a->geom_margin[g]; m->geom_gap[G2] = a->geom_gap[g];
    for (int q = 0; q < 2; q++) m->geom_solref[2*G2+q] = a->geom_solref[2*g+q];
    for (int q = 0; q < 5; q++) m->geom_solimp[5*G2+q] = a->geom_solimp[5*g+q];
    gnames[G2] = nm(a->geom_names, g, k);
  }
  // ---- candidate pairs (instances never meet), equalities
  m->pair_geom1 = ialloc((size_t)G*np); m->pair_geom2 = ialloc((size_t)G*np);
  for (int k = 0; k < G; k++) for (int i = 0; i < np; i++) { m->pair_geom1[i + k*np] = gmap(k, a->pair_geom1[i]); m->pair_geom2[i + k*np] = gmap(k, a->pair_geom2[i]); }
  m->eq_type = ialloc((size_t)G*ne); m->eq_obj1id = ialloc((size_t)G*ne); m->eq_obj2id = ialloc((size_t)G*ne); m->eq_active = ialloc((size_t)G*ne);
  m->eq_data = dalloc((size_t)11*G*ne); m->eq_solref = dalloc((size_t)2*G*ne); m->eq_solimp = dalloc((size_t)5*G*ne);
  for (int k = 0; k < G; k++) {
    for (int e = 0; e < ne; e++) {
      m->eq_type[e + k*ne] = a->eq_type[e]; m->eq_active[e + k*ne] = a->eq_active[e];
      m->eq_obj1id[e + k*ne] = a->eq_obj1id[e] + k * nj; m->eq_obj2id[e + k*ne] = a->eq_obj2id[e] >= 0 ? a->eq_obj2id[e] + k * nj : -1;
    }
    cpd(m->eq_data + (size_t)11*k*ne, a->eq_data, (size_t)11*ne); cpd(m->eq_solref + (size_t)2*k*ne, a->eq_solref, (size_t)2*ne); cpd(m->eq_solimp + (size_t)5*k*ne, a->eq_solimp, (size_t)5*ne);
  }
  // ---- mesh assets: shared
  m->mesh_vertadr = ialloc(a->nmesh); m->mesh_vertnum = ialloc(a->nmesh); m->mesh_vert = dalloc((size_t)3 * a->nmeshvert);
  for (int i = 0; i < a->nmesh; i++) { m->mesh_vertadr[i] = a->mesh_vertadr[i]; m->mesh_vertnum[i] = a->mesh_vertnum[i]; }
  cpd(m->mesh_vert, a->mesh_vert, (size_t)3 * a->nmeshvert);
  m->body_names = dupnames(bnames); m->jnt_names = dupnames(jnames); m->geom_names = dupnames(gnames);
  m->body_mocapid = ialloc(NB);
  for (int b = 0; b < NB; b++) m->body_mocapid[b] = -1;
  return m;
}

extern "C" void mjh_model_destroy(mjh_model* m) {
  if (!m) return;
  void* ptrs[] = {m->body_parentid, m->body_rootid, m->body_weldid, m->body_jntadr, m->body_jntnum, m->body_dofadr, m->body_dofnum,
    m->body_treeid, m->body_level, m->body_geomadr, m->body_geomnum, m->body_pos, m->body_quat, m->body_ipos, m->body_iquat,
    m->body_mass, m->body_inertia, m->body_gravcomp, m->body_invweight0, m->jnt_type, m->jnt_qposadr, m->jnt_dofadr, m->jnt_bodyid,
    m->jnt_limited, m->jnt_pos, m->jnt_axis, m->jnt_stiffness, m->jnt_range, m->jnt_margin, m->jnt_solref, m->jnt_solimp, m->qpos0,
    m->qpos_spring, m->dof_bodyid, m->dof_jntid, m->dof_parentid, m->dof_Madr, m->dof_treeid, m->dof_armature, m->dof_damping,
    m->dof_frictionloss, m->dof_invweight0, m->dof_solref, m->dof_solimp, m->tree_dofadr, m->tree_dofnum, m->tree_bodyid,
    m->geom_type, m->geom_bodyid, m->geom_condim, m->geom_contype, m->geom_conaffinity, m->geom_priority, m->geom_pos, m->geom_quat,
    m->geom_size, m->geom_rbound, m->geom_friction, m->geom_solmix, m->geom_solref, m->geom_solimp, m->geom_margin, m->geom_gap,
    m->pair_geom1, m->pair_geom2, m->eq_type, m->eq_obj1id, m->eq_obj2id, m->eq_active, m->eq_data, m->eq_solref, m->eq_solimp,
    m->geom_dataid, m->mesh_vertadr, m->mesh_vertnum, m->mesh_vert,
    m->site_bodyid, m->site_pos, m->site_quat, m->sensor_type, m->sensor_objid, m->sensor_adr, m->body_mocapid};
  for (void* p : ptrs) std::free(p);
  auto freen = [](char** n, int c) { if (!n) return; for (int i = 0; i < c; i++) std::free(n[i]); std::free(n); };
  freen(m->body_names, m->nbody); freen(m->jnt_names, m->njnt); freen(m->geom_names, m->ngeom);
  freen(m->site_names, m->nsite); freen(m->sensor_names, m->nsensor);
  std::free(m);
}

extern "C" int mjh_name2id(const mjh_model* m, int objtype, const char* name) {
  if (!m || !name) return -1;
  char** t = objtype == 0 ? m->body_names : objtype == 1 ? m->jnt_names : objtype == 2 ? m->geom_names : objtype == 3 ? m->site_names : m->sensor_names;
  int n = objtype == 0 ? m->nbody : objtype == 1 ? m->njnt : objtype == 2 ? m->ngeom : objtype == 3 ? m->nsite : m->nsensor;
  if (!t || objtype < 0 || objtype > 4) return -1;     // a model assembled without name tables
  for (int i = 0; i < n; i++) if (t[i] && std::strcmp(t[i], name) == 0) return i;
  return -1;
}
extern "C" const char* mjh_id2name(const mjh_model* m, int objtype, int id) {
  if (!m) return nullptr;
  char** t = objtype == 0 ? m->body_names : objtype == 1 ? m->jnt_names : objtype == 2 ? m->geom_names : objtype == 3 ? m->site_names : m->sensor_names;
  int n = objtype == 0 ? m->nbody : objtype == 1 ? m->njnt : objtype == 2 ? m->ngeom : objtype == 3 ? m->nsite : m->nsensor;
  if (!t || objtype < 0 || objtype > 4 || id < 0 || id >= n) return nullptr;
  return t[id];
}
