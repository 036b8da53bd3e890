// scenes.cpp — programmatic restatements of the SURVEY.md §8-d benchmark scenes.
//
// S24 is the headline workload of BASELINE.json ("24-DoF / ~30-contact scene,
// 4096 envs"): 4 free boxes (spawned-primitive size range of
// test/test_spawn_and_destroy.py:32,39-41) in a walled pen on the floor of
// model/world/empty.xml:12 (plane, condim 4, friction 2/0.05/0.01), dt 0.005,
// g -9.81 (model/world/empty.xml:2).  The other builders restate
// model/test/pendulum.xml (C1/C5) and the Panda arm chain of
// model/test/ridgeback_panda/ridgeback_panda.xml:53-87 (C3) as data.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/mjhip.h"
#include "hmath.h"

namespace {

struct Pcg32 {
  uint64_t state, inc;
  explicit Pcg32(uint64_t seed, uint64_t seq = 54u) {
    state = 0u; inc = (seq << 1u) | 1u; next(); state += seed; next();
  }
  uint32_t next() {
    uint64_t old = state;
    state = old * 6364136223846793005ULL + inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
  }
  double uniform() { return next() * (1.0 / 4294967296.0); }
  double uniform(double lo, double hi) { return lo + (hi - lo) * uniform(); }
};

void empty_world_options(mjh_builder* b, double gz) {
  mjh_option o; mjh_builder_get_option(b, &o);
  o.timestep = 0.005; o.gravity[0] = 0; o.gravity[1] = 0; o.gravity[2] = gz;  // world/empty.xml:2, pendulum.xml:2
  o.iterations = 100; o.tolerance = 1e-8;                                      // MuJoCo defaults (no solver attrs in any reference XML)
  mjh_builder_set_option(b, &o);
}
void add_floor(mjh_builder* b, bool empty_xml_params) {
  const double size[3] = {0, 0, 0.05};
  const double fr[3] = {2, 0.05, 0.01};  // world/empty.xml:12
  // pendulum.xml:14 floor has default friction/condim
  mjh_builder_add_geom(b, "floor", 0, MJH_GEOM_PLANE, size, nullptr, nullptr, empty_xml_params ? fr : nullptr,
                       empty_xml_params ? 4 : 3, -1, -1, -1);
}

}  // namespace

static const int S24_NBOX = 4;
static const double S24_PEN_HALF = 0.175;  // 0.35 m square pen

// S24 with another pen: the same four boxes (sizes, poses and seeds come from mjh_scene_s24_randomize) between walls pen_half
// from the centre.  bench.py's `s24d` narrows the pen until the boxes have to stand on each other and lean on all four walls:
// the ~30 contacts / >= 130 rows the metric's name speaks of (the D2-exact pen settles at ~17 contacts).
extern "C" mjh_model* mjh_scene_s24_pen(double pen_half, int maxcon) {
  if (!(pen_half > 0.05) || maxcon < 8) return nullptr;
  const double S24_PEN_HALF = pen_half;
  mjh_builder* b = mjh_builder_create();
  empty_world_options(b, -9.81);
  add_floor(b, true);
  const double t = 0.025, h = 0.75, L = S24_PEN_HALF + 2 * t;
  const double wsx[3] = {t, L, h}, wsy[3] = {L, t, h};
  const double px[3] = {S24_PEN_HALF + t, 0, h}, nx[3] = {-(S24_PEN_HALF + t), 0, h};
  const double py[3] = {0, S24_PEN_HALF + t, h}, ny[3] = {0, -(S24_PEN_HALF + t), h};
  mjh_builder_add_geom(b, "wall_px", 0, MJH_GEOM_BOX, wsx, px, nullptr, nullptr, -1, -1, -1, -1);
  mjh_builder_add_geom(b, "wall_nx", 0, MJH_GEOM_BOX, wsx, nx, nullptr, nullptr, -1, -1, -1, -1);
  mjh_builder_add_geom(b, "wall_py", 0, MJH_GEOM_BOX, wsy, py, nullptr, nullptr, -1, -1, -1, -1);
  mjh_builder_add_geom(b, "wall_ny", 0, MJH_GEOM_BOX, wsy, ny, nullptr, nullptr, -1, -1, -1, -1);
  for (int k = 0; k < S24_NBOX; k++) {
    char name[32]; std::snprintf(name, sizeof name, "box%d", k);
    const double pos[3] = {0, 0, 0.15 + 0.30 * k};
    int body = mjh_builder_add_body(b, name, 0, pos, nullptr, 0);
    std::snprintf(name, sizeof name, "box%d_free", k);
    mjh_builder_add_joint(b, name, body, MJH_JNT_FREE, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0);
    const double sz[3] = {0.0875, 0.0875, 0.0875};
    std::snprintf(name, sizeof name, "box%d_geom", k);
    mjh_builder_add_geom(b, name, body, MJH_GEOM_BOX, sz, nullptr, nullptr, nullptr, -1, -1, -1, -1);
  }
  // capacity: over 4096 envs the pile stays below 31 contacts in the benchmark window (1400 steps) and reaches 35 in a
  // 20 000-step soak (the piles keep compacting); 40 contacts x 6 rows leaves a margin (overflow is flagged, not silent)
  // (tools/ncon_hist.py); overflow drops the excess contacts and raises the per-env flag
  mjh_builder_set_capacity(b, maxcon, maxcon * 6);
  mjh_model* m = mjh_builder_compile(b);
  mjh_builder_destroy(b);
  return m;
}
extern "C" mjh_model* mjh_scene_s24(void) { return mjh_scene_s24_pen(S24_PEN_HALF, 40); }

extern "C" int mjh_scene_s24_randomize(const mjh_model* m, int env0, int nenv, unsigned seed_base,
                                       double* qpos, double* geom_size, double* geom_rbound,
                                       double* body_mass, double* body_inertia,
                                       double* body_invweight0, double* dof_invweight0) {
  if (!m || m->nv != 6 * S24_NBOX) return MJH_ERR_ARG;
  const int nq = m->nq, nv = m->nv, nb = m->nbody, ng = m->ngeom;
  for (int e = 0; e < nenv; e++) {
    Pcg32 rng((uint64_t)seed_base + (uint64_t)(env0 + e));
    double* q = qpos ? qpos + (size_t)e * nq : nullptr;
    // start from the shared model tables
    if (geom_size) for (int i = 0; i < 3 * ng; i++) geom_size[(size_t)e * 3 * ng + i] = m->geom_size[i];
    if (geom_rbound) for (int i = 0; i < ng; i++) geom_rbound[(size_t)e * ng + i] = m->geom_rbound[i];
    if (body_mass) for (int i = 0; i < nb; i++) body_mass[(size_t)e * nb + i] = m->body_mass[i];
    if (body_inertia) for (int i = 0; i < 3 * nb; i++) body_inertia[(size_t)e * 3 * nb + i] = m->body_inertia[i];
    if (body_invweight0) for (int i = 0; i < 2 * nb; i++) body_invweight0[(size_t)e * 2 * nb + i] = m->body_invweight0[i];
    if (dof_invweight0) for (int i = 0; i < nv; i++) dof_invweight0[(size_t)e * nv + i] = m->dof_invweight0[i];
    for (int k = 0; k < S24_NBOX; k++) {
      int body = 1 + k, geom = m->body_geomadr[body], da = m->body_dofadr[body], qa = m->jnt_qposadr[m->body_jntadr[body]];
      double hx = rng.uniform(0.05, 0.125), hy = rng.uniform(0.05, 0.125), hz = rng.uniform(0.05, 0.125);
      double x = rng.uniform(-0.05, 0.05), y = rng.uniform(-0.05, 0.05);
      double u1 = rng.uniform(), u2 = rng.uniform(), u3 = rng.uniform();
      const double twopi = 6.283185307179586476925;
      double a = std::sqrt(1 - u1), bq = std::sqrt(u1);
      double quat[4] = {a * std::sin(twopi * u2), a * std::cos(twopi * u2), bq * std::sin(twopi * u3), bq * std::cos(twopi * u3)};
      hm::normalize4(quat);
      if (q) { q[qa] = x; q[qa+1] = y; q[qa+2] = 0.15 + 0.30 * k; for (int i = 0; i < 4; i++) q[qa+3+i] = quat[i]; }
      double mass = 1000.0 * 8 * hx * hy * hz;
      double I[3] = {mass / 3 * (hy*hy + hz*hz), mass / 3 * (hx*hx + hz*hz), mass / 3 * (hx*hx + hy*hy)};
      if (geom_size) { double* s = geom_size + (size_t)e * 3 * ng + 3 * geom; s[0] = hx; s[1] = hy; s[2] = hz; }
      if (geom_rbound) geom_rbound[(size_t)e * ng + geom] = std::sqrt(hx*hx + hy*hy + hz*hz);
      if (body_mass) body_mass[(size_t)e * nb + body] = mass;
      if (body_inertia) for (int i = 0; i < 3; i++) body_inertia[(size_t)e * 3 * nb + 3 * body + i] = I[i];
      double tr = 1 / mass, rr = (1 / I[0] + 1 / I[1] + 1 / I[2]) / 3;
      if (body_invweight0) { body_invweight0[(size_t)e * 2 * nb + 2 * body] = tr; body_invweight0[(size_t)e * 2 * nb + 2 * body + 1] = rr; }
      if (dof_invweight0) for (int i = 0; i < 3; i++) { dof_invweight0[(size_t)e * nv + da + i] = tr; dof_invweight0[(size_t)e * nv + da + 3 + i] = rr; }
    }
  }
  return MJH_OK;
}

// Per-env randomisation of a scene of free boxes (BASELINE config C2, SURVEY.md §8-d D3: "64 boxes, half-extents
// U[0.05,0.125]^3, released from a 4x4x4 lattice (pitch 0.3 m, z0 in [0.5, 1.4]) with +-0.01 m jitter and random
// orientation"): every body that carries one free joint and one box geom gets its own half-extents (and the mass, inertia and
// inverse weights that follow at density 1000), its lattice position from the model's qpos0 plus U[-jitter, jitter]^3 and a
// uniformly random orientation.  PCG32 seeded `seed_base + env`, like S24.  Every out pointer may be NULL.
extern "C" int mjh_scene_boxes_randomize(const mjh_model* m, int env0, int nenv, unsigned seed_base, double jitter,
                                         double* qpos, double* geom_size, double* geom_rbound,
                                         double* body_mass, double* body_inertia,
                                         double* body_invweight0, double* dof_invweight0) {
  if (!m || nenv < 0) return MJH_ERR_ARG;
  const int nq = m->nq, nv = m->nv, nb = m->nbody, ng = m->ngeom;
  for (int e = 0; e < nenv; e++) {
    Pcg32 rng((uint64_t)seed_base + (uint64_t)(env0 + e));
    if (qpos) for (int i = 0; i < nq; i++) qpos[(size_t)e * nq + i] = m->qpos0[i];
    if (geom_size) for (int i = 0; i < 3 * ng; i++) geom_size[(size_t)e * 3 * ng + i] = m->geom_size[i];
    if (geom_rbound) for (int i = 0; i < ng; i++) geom_rbound[(size_t)e * ng + i] = m->geom_rbound[i];
    if (body_mass) for (int i = 0; i < nb; i++) body_mass[(size_t)e * nb + i] = m->body_mass[i];
    if (body_inertia) for (int i = 0; i < 3 * nb; i++) body_inertia[(size_t)e * 3 * nb + i] = m->body_inertia[i];
    if (body_invweight0) for (int i = 0; i < 2 * nb; i++) body_invweight0[(size_t)e * 2 * nb + i] = m->body_invweight0[i];
    if (dof_invweight0) for (int i = 0; i < nv; i++) dof_invweight0[(size_t)e * nv + i] = m->dof_invweight0[i];
    for (int body = 1; body < nb; body++) {
      if (m->body_jntnum[body] != 1 || m->jnt_type[m->body_jntadr[body]] != MJH_JNT_FREE || m->body_geomnum[body] != 1) continue;
      const int geom = m->body_geomadr[body];
      if (m->geom_type[geom] != MJH_GEOM_BOX) continue;
      const int da = m->body_dofadr[body], qa = m->jnt_qposadr[m->body_jntadr[body]];
      const double hx = rng.uniform(0.05, 0.125), hy = rng.uniform(0.05, 0.125), hz = rng.uniform(0.05, 0.125);
      const double jx = rng.uniform(-jitter, jitter), jy = rng.uniform(-jitter, jitter), jz = rng.uniform(-jitter, jitter);
      const double u1 = rng.uniform(), u2 = rng.uniform(), u3 = rng.uniform();
      const double twopi = 6.283185307179586476925;
      const double a = std::sqrt(1 - u1), bq = std::sqrt(u1);
      double quat[4] = {a * std::sin(twopi * u2), a * std::cos(twopi * u2), bq * std::sin(twopi * u3), bq * std::cos(twopi * u3)};
      hm::normalize4(quat);
      if (qpos) { double* q = qpos + (size_t)e * nq + qa; q[0] += jx; q[1] += jy; q[2] += jz; for (int i = 0; i < 4; i++) q[3 + i] = quat[i]; }
      const double mass = 1000.0 * 8 * hx * hy * hz;
      const double I[3] = {mass / 3 * (hy*hy + hz*hz), mass / 3 * (hx*hx + hz*hz), mass / 3 * (hx*hx + hy*hy)};
      if (geom_size) { double* sz = geom_size + (size_t)e * 3 * ng + 3 * geom; sz[0] = hx; sz[1] = hy; sz[2] = hz; }
      if (geom_rbound) geom_rbound[(size_t)e * ng + geom] = std::sqrt(hx*hx + hy*hy + hz*hz);
      if (body_mass) body_mass[(size_t)e * nb + body] = mass;
      if (body_inertia) for (int i = 0; i < 3; i++) body_inertia[(size_t)e * 3 * nb + 3 * body + i] = I[i];
      const double tr = 1 / mass, rr = (1 / I[0] + 1 / I[1] + 1 / I[2]) / 3;
      if (body_invweight0) { body_invweight0[(size_t)e * 2 * nb + 2 * body] = tr; body_invweight0[(size_t)e * 2 * nb + 2 * body + 1] = rr; }
      if (dof_invweight0) for (int i = 0; i < 3; i++) { dof_invweight0[(size_t)e * nv + da + i] = tr; dof_invweight0[(size_t)e * nv + da + 3 + i] = rr; }
    }
  }
  return MJH_OK;
}

// model/test/pendulum.xml:18-29 — three bodies on ball joints (damping 0.5) sharing the anchor (0,0,2),
// gravity -0.1; geoms sphere / box / cylinder of "size .1 .1 .1".
extern "C" mjh_model* mjh_scene_pendulum(void) {
  mjh_builder* b = mjh_builder_create();
  empty_world_options(b, -0.1);
  add_floor(b, false);
  const double bp[3][3] = {{1, 0, 2}, {-0.5, 0.866, 2}, {-0.5, -0.866, 2}};
  const double jp[3][3] = {{-1, 0, 0}, {0.5, -0.866, 0}, {0.5, 0.866, 0}};
  const char* names[3] = {"sphere", "cube", "cylinder"};
  const int gt[3] = {MJH_GEOM_SPHERE, MJH_GEOM_BOX, MJH_GEOM_CYLINDER};
  for (int k = 0; k < 3; k++) {
    int body = mjh_builder_add_body(b, names[k], 0, bp[k], nullptr, 0);
    std::string jn = std::string(names[k]) + "_ball";
    mjh_builder_add_joint(b, jn.c_str(), body, MJH_JNT_BALL, jp[k], nullptr, nullptr, 0.5, 0, 0, 0, 0);
    const double sz[3] = {0.1, 0.1, 0.1};
    std::string gn = std::string(names[k]) + "_geom";
    mjh_builder_add_geom(b, gn.c_str(), body, gt[k], sz, nullptr, nullptr, nullptr, -1, -1, -1, -1);
  }
  mjh_model* m = mjh_builder_compile(b);
  mjh_builder_destroy(b);
  return m;
}

// 7-hinge chain with the Panda kinematics and limits (ridgeback_panda.xml:53-87); fixed base.
// Link inertias come from the main cylinder geom of each link at density 1000 (the file gives
// none for arm links).  Geoms carry contype = conaffinity = 0: adjacent link geoms overlap by
// construction, and C3 exercises limits + the computed-torque controller, not contacts.
extern "C" mjh_model* mjh_scene_arm7(int gravcomp) {
  mjh_builder* b = mjh_builder_create();
  empty_world_options(b, -9.81);
  add_floor(b, true);
  struct L { double pos[3], quat[4], range[2], gsize[2], gpos[3]; };
  const double s = 0.707107;
  const L links[7] = {
      {{0.33, 0, 0.919499}, {1, 0, 0, 0}, {-2.8973, 2.8973}, {0.06, 0.1415}, {0, 0, -0.1915}},
      {{0, 0, 0}, {s, -s, 0, 0}, {-1.7628, 1.7628}, {0.06, 0.06}, {0, 0, 0}},
      {{0, -0.316, 0}, {s, s, 0, 0}, {-2.8973, 2.8973}, {0.06, 0.075}, {0, 0, -0.145}},
      {{0.0825, 0, 0}, {s, s, 0, 0}, {-3.0718, -0.0698}, {0.06, 0.06}, {0, 0, 0}},
      {{-0.0825, 0.384, 0}, {s, -s, 0, 0}, {-2.8973, 2.8973}, {0.06, 0.05}, {0, 0, -0.26}},
      {{0, 0, 0}, {s, s, 0, 0}, {-0.0175, 3.7525}, {0.05, 0.04}, {0, 0, -0.03}},
      {{0.088, 0, 0}, {s, s, 0, 0}, {-2.8973, 2.8973}, {0.04, 0.07}, {0, 0, 0.01}}};
  int parent = 0;
  for (int k = 0; k < 7; k++) {
    char name[32]; std::snprintf(name, sizeof name, "panda_link%d", k + 1);
    int body = mjh_builder_add_body(b, name, parent, links[k].pos, links[k].quat, gravcomp ? 1.0 : 0.0);
    std::snprintf(name, sizeof name, "panda_joint%d", k + 1);
    const double axis[3] = {0, 0, 1};
    // joint 4's range excludes 0: start it inside the range
    double ref = 0;
    mjh_builder_add_joint(b, name, body, MJH_JNT_HINGE, nullptr, axis, links[k].range, 0, 0, 0, 0, ref);
    const double sz[3] = {links[k].gsize[0], links[k].gsize[1], 0};
    std::snprintf(name, sizeof name, "panda_link%d_geom", k + 1);
    mjh_builder_add_geom(b, name, body, MJH_GEOM_CYLINDER, sz, links[k].gpos, nullptr, nullptr, -1, 0, 0, -1);
    parent = body;
  }
  mjh_model* m = mjh_builder_compile(b);
  mjh_builder_destroy(b);
  return m;
}

// nbox free boxes released over the empty.xml floor (C2 family, test_spawn_and_destroy.py:35-41 sizes).
extern "C" mjh_model* mjh_scene_boxpile(int nbox) {
  mjh_builder* b = mjh_builder_create();
  empty_world_options(b, -9.81);
  add_floor(b, true);
  int side = 1; while (side * side * side < nbox) side++;
  for (int k = 0; k < nbox; k++) {
    char name[32]; std::snprintf(name, sizeof name, "box%d", k);
    int ix = k % side, iy = (k / side) % side, iz = k / (side * side);
    const double pos[3] = {0.3 * (ix - 0.5 * (side - 1)), 0.3 * (iy - 0.5 * (side - 1)), 0.5 + 0.3 * iz};
    int body = mjh_builder_add_body(b, name, 0, pos, nullptr, 0);
    std::snprintf(name, sizeof name, "box%d_free", k);
    mjh_builder_add_joint(b, name, body, MJH_JNT_FREE, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0);
    const double sz[3] = {0.0875, 0.0875, 0.0875};
    std::snprintf(name, sizeof name, "box%d_geom", k);
    mjh_builder_add_geom(b, name, body, MJH_GEOM_BOX, sz, nullptr, nullptr, nullptr, -1, -1, -1, -1);
  }
  mjh_model* m = mjh_builder_compile(b);
  mjh_builder_destroy(b);
  return m;
}
