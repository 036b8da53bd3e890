// mjcf_loader.cpp — MJCF-subset loader: the model-ingest boundary of the reference
// (load_XML -> mj_loadXML, include/mujoco_sim/mj_util.h:185-193) for the element subset its models use on
// the step path: <compiler angle autolimits>, <option timestep gravity impratio iterations tolerance>,
// root <default> (<geom>, <joint>), <worldbody>/<body> trees with <inertial>, <joint> (free, ball, hinge, slide),
// <freejoint>, <geom> (plane, sphere, capsule, cylinder, box, ellipsoid, mesh: binary / ASCII STL and OBJ assets become convex hulls), gravcomp,
// <contact><exclude>, <equality><joint polycoef> / <weld> / <connect>, <body mocap>, <site>, <sensor><force> / <torque>.  Everything is translated into mjh_builder_* calls; physics
// defaults follow MuJoCo's documented defaults (angle = degree, hinge axis 0 0 1, geom type sphere, ...).
// Not handled (reported in the returned note, mjh_load_note): tendons, actuators, height fields, sensors other than force / torque.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <array>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/mjhip.h"
#include "hmath.h"

void mjh_set_error(const std::string& s);

namespace {

struct Node {
  std::string tag;
  std::vector<std::pair<std::string, std::string>> attr;
  std::vector<std::unique_ptr<Node>> kids;
  const char* get(const char* k) const { for (auto& a : attr) if (a.first == k) return a.second.c_str(); return nullptr; }
};

// ---- minimal XML reader: elements, attributes, comments, declarations; character data is ignored
struct Xml {
  const char* p; const char* end; std::string err;
  void skip_ws() { while (p < end && std::isspace((unsigned char)*p)) p++; }
  bool starts(const char* s) const { size_t n = std::strlen(s); return (size_t)(end - p) >= n && std::strncmp(p, s, n) == 0; }
  bool skip_misc() {  // whitespace, comments, <? ?>, <! >, text
    for (;;) {
      while (p < end && *p != '<') p++;
      if (p >= end) return true;
      if (starts("<!--")) { const char* q = std::strstr(p + 4, "-->"); if (!q) { err = "unterminated comment"; return false; } p = q + 3; continue; }
      if (starts("<?")) { const char* q = std::strstr(p + 2, "?>"); if (!q) { err = "unterminated declaration"; return false; } p = q + 2; continue; }
      if (starts("<!")) { const char* q = std::strchr(p, '>'); if (!q) { err = "unterminated <!"; return false; } p = q + 1; continue; }
      return true;
    }
  }
  std::unique_ptr<Node> element() {
    if (!skip_misc() || p >= end || *p != '<' || p[1] == '/') return nullptr;
    p++;
    auto n = std::make_unique<Node>();
    while (p < end && !std::isspace((unsigned char)*p) && *p != '>' && *p != '/') n->tag.push_back(*p++);
    for (;;) {
      skip_ws();
      if (p >= end) { err = "unexpected end in <" + n->tag + ">"; return nullptr; }
      if (*p == '/') { if (p + 1 < end && p[1] == '>') { p += 2; return n; } err = "bad '/' in <" + n->tag + ">"; return nullptr; }
      if (*p == '>') { p++; break; }
      std::string k, v;
      while (p < end && *p != '=' && !std::isspace((unsigned char)*p)) k.push_back(*p++);
      skip_ws();
      if (p >= end || *p != '=') { err = "attribute without value in <" + n->tag + ">"; return nullptr; }
      p++; skip_ws();
      if (p >= end || (*p != '"' && *p != '\'')) { err = "unquoted attribute in <" + n->tag + ">"; return nullptr; }
      const char q = *p++;
      while (p < end && *p != q) v.push_back(*p++);
      if (p >= end) { err = "unterminated attribute in <" + n->tag + ">"; return nullptr; }
      p++;
      n->attr.push_back({k, v});
    }
    for (;;) {  // children until the matching close tag
      if (!skip_misc()) return nullptr;
      if (p >= end) { err = "missing </" + n->tag + ">"; return nullptr; }
      if (p[1] == '/') {
        const char* q = std::strchr(p, '>');
        if (!q) { err = "unterminated close tag"; return nullptr; }
        std::string t(p + 2, q); while (!t.empty() && std::isspace((unsigned char)t.back())) t.pop_back();
        if (t != n->tag) { err = "mismatched </" + t + "> for <" + n->tag + ">"; return nullptr; }
        p = q + 1; return n;
      }
      auto c = element();
      if (!c) return nullptr;
      n->kids.push_back(std::move(c));
    }
  }
};

int nums(const char* s, double* out, int maxn) {
  int n = 0; if (!s) return 0;
  char* e;
  while (n < maxn) { while (*s && (std::isspace((unsigned char)*s) || *s == ',')) s++; if (!*s) break; double v = std::strtod(s, &e); if (e == s) break; out[n++] = v; s = e; }
  return n;
}

struct Defaults {
  double geom_friction[3] = {1, 0.005, 0.0001}; int geom_condim = 3, geom_contype = 1, geom_conaffinity = 1; double geom_density = 1000;
  double jnt_damping = 0, jnt_stiffness = 0, jnt_armature = 0, jnt_frictionloss = 0;
};

// floor applied to every file loaded afterwards: MjSim::init() writes boundmass = boundinertia = 1e-6 into the <compiler>
// element of whatever it loads (mj_sim.cpp:584-590)
// (thread_local: the options of one thread's loads never leak into another thread's; mjh_load_mjcf_files_opt takes them per
// call and leaves the thread's settings as it found them)
static thread_local double g_boundmass = 0, g_boundinertia = 0;
static thread_local int g_robot_gravcomp = -1;   // mjh_load_set_robot_gravcomp: -1 keep the files' values, 0 / 1 force it on every robot body
static thread_local std::map<std::string, std::array<double, 6>> g_robot_pose;   // mjh_load_set_robot_pose: root body name -> x y z roll pitch yaw
static thread_local unsigned g_odom_mask = 0;      // mjh_load_set_odom_joints: bits 0..5 = lin x y z, ang x y z
static thread_local int g_pc_exclude_level = 0;   // mjh_load_set_parent_child_exclude: a body does not collide with its first k ancestors (mujoco_compile.cpp:250-290)
static thread_local int g_load_meshes = 1;   // mjh_load_set_mesh_mode: 0 = skip mesh assets (their geoms are reported and dropped)

struct Loader {
  mjh_builder* b = nullptr;
  bool degree = true, autolimits = false, balance = false, robot_file = false;
  int nfiles = 1;                    // files of this load (mjh_load_mjcf_files): a single file is treated as the robot file by the per-robot rules that say so
  double bmass = 0, binertia = 0;   // <compiler boundmass boundinertia>, raised to the process-wide floor of mjh_load_set_bounds
  Defaults def;
  std::map<std::string, int> body_id, joint_id, mesh_id, site_id;
  std::vector<int> body_parent;      // parent body of every body added so far
  // <default class="..."> tables: class -> element tag -> attributes (a nested class starts from its parent's); the
  // unnamed top-level <default> is class "main".  An element takes the attributes it does not set itself from its class
  // (its own class="" attribute, else the nearest enclosing body's childclass, else "main").
  typedef std::vector<std::pair<std::string, std::string>> Attrs;
  std::map<std::string, std::map<std::string, Attrs>> classes;
  std::string childclass;
  void read_defaults(const Node& d, const std::string& parent) {
    const std::string cls = d.get("class") ? d.get("class") : (parent.empty() ? "main" : parent);
    if (!parent.empty() && cls != parent) classes[cls] = classes[parent];
    for (auto& k : d.kids) {
      if (k->tag == "default") { read_defaults(*k, cls); continue; }
      Attrs& dst = classes[cls][k->tag];
      for (auto& a : k->attr) {
        bool found = false;
        for (auto& e : dst) if (e.first == a.first) { e.second = a.second; found = true; }
        if (!found) dst.push_back(a);
      }
    }
  }
  Node with_defaults(const Node& n) const {
    Node m; m.tag = n.tag; m.attr = n.attr;
    const std::string cls = n.get("class") ? n.get("class") : (childclass.empty() ? "main" : childclass);
    auto ci = classes.find(cls);
    if (ci == classes.end()) return m;
    auto ti = ci->second.find(n.tag);
    if (ti == ci->second.end()) return m;
    for (auto& a : ti->second) if (!n.get(a.first.c_str())) m.attr.push_back(a);
    return m;
  }
  std::string note, basedir, meshdir;   // directory of the file being read (empty for a string), <compiler meshdir>
  int nameless = 0;

  double ang(double v) const { return degree ? v * 3.14159265358979323846 / 180.0 : v; }
  bool orientation(const Node& n, double* quat) {   // quat | euler (xyz) ; returns false if neither
    double v[4];
    if (nums(n.get("quat"), v, 4) == 4) { for (int i = 0; i < 4; i++) quat[i] = v[i]; hm::normalize4(quat); return true; }
    if (nums(n.get("euler"), v, 3) == 3) {
      double q[4] = {1, 0, 0, 0};
      for (int k = 0; k < 3; k++) { double ax[3] = {0, 0, 0}; ax[k] = 1; double r[4], t[4]; hm::axisangle2quat(r, ax, ang(v[k])); hm::mulquat(t, q, r); std::memcpy(q, t, sizeof q); }
      std::memcpy(quat, q, sizeof q); return true;
    }
    return false;
  }
  void geom(const Node& n0, int body) {
    const Node n = with_defaults(n0);
    const char* type = n.get("type");
    int gt = MJH_GEOM_SPHERE;
    if (type) {
      std::string t = type;
      if (t == "plane") gt = MJH_GEOM_PLANE; else if (t == "sphere") gt = MJH_GEOM_SPHERE; else if (t == "capsule") gt = MJH_GEOM_CAPSULE;
      else if (t == "cylinder") gt = MJH_GEOM_CYLINDER; else if (t == "box") gt = MJH_GEOM_BOX; else if (t == "ellipsoid") gt = MJH_GEOM_ELLIPSOID;
      else if (t == "mesh") gt = MJH_GEOM_MESH;
      else { note += "skipped <geom type=\"" + t + "\">; "; return; }
    } else if (n.get("mesh")) gt = MJH_GEOM_MESH;
    double size[3] = {0, 0, 0}, pos[3] = {0, 0, 0}, quat[4] = {1, 0, 0, 0}, fr[3];
    nums(n.get("size"), size, 3); nums(n.get("pos"), pos, 3); orientation(n, quat);
    double ft[6];
    if (nums(n.get("fromto"), ft, 6) == 6) {
      // fromto: the geom's z axis runs from the first point to the second; size = radius (capsule, cylinder) or the two
      // lateral half-sizes (box, ellipsoid), the half-length comes from the distance
      if (gt != MJH_GEOM_CAPSULE && gt != MJH_GEOM_CYLINDER && gt != MJH_GEOM_BOX && gt != MJH_GEOM_ELLIPSOID) { note += "skipped <geom fromto> of this type; "; return; }
      double d[3] = {ft[3] - ft[0], ft[4] - ft[1], ft[5] - ft[2]};
      const double len = hm::norm3(d);
      for (int k = 0; k < 3; k++) pos[k] = 0.5 * (ft[k] + ft[3 + k]);
      if (len > 1e-12) {
        for (int k = 0; k < 3; k++) d[k] /= len;
        const double zax[3] = {0, 0, 1}; double ax[3]; hm::cross(ax, zax, d);
        const double sn = hm::norm3(ax), cs = d[2];
        if (sn < 1e-12) { quat[0] = cs > 0 ? 1 : 0; quat[1] = cs > 0 ? 0 : 1; quat[2] = quat[3] = 0; }
        else { for (int k = 0; k < 3; k++) ax[k] /= sn; hm::axisangle2quat(quat, ax, std::atan2(sn, cs)); }
      }
      if (gt == MJH_GEOM_CAPSULE || gt == MJH_GEOM_CYLINDER) size[1] = 0.5 * len; else { size[1] = size[1] > 0 ? size[1] : size[0]; size[2] = 0.5 * len; }
    }
    for (const char* a : {"solref", "solimp", "margin", "gap", "solmix", "priority", "fluidshape"})
      if (n.get(a)) note += std::string("ignored geom attribute ") + a + " (MuJoCo defaults used); ";
    std::memcpy(fr, def.geom_friction, sizeof fr);
    double t3[3]; int nf = nums(n.get("friction"), t3, 3); for (int i = 0; i < nf; i++) fr[i] = t3[i];
    double v; int condim = def.geom_condim, contype = def.geom_contype, conaff = def.geom_conaffinity; double density = def.geom_density;
    if (nums(n.get("condim"), &v, 1)) condim = (int)v;
    if (nums(n.get("contype"), &v, 1)) contype = (int)v;
    if (nums(n.get("conaffinity"), &v, 1)) conaff = (int)v;
    if (nums(n.get("density"), &v, 1)) density = v;
    if (gt == MJH_GEOM_MESH) {
      auto it = mesh_id.find(n.get("mesh") ? n.get("mesh") : "");
      if (it == mesh_id.end()) { note += std::string("skipped mesh geom (mesh ") + (n.get("mesh") ? n.get("mesh") : "?") + " not loaded); "; return; }
      mjh_builder_add_mesh_geom(b, n.get("name"), body, it->second, pos, quat, fr, condim, contype, conaff, density);
      return;
    }
    mjh_builder_add_geom(b, n.get("name"), body, gt, size, pos, quat, fr, condim, contype, conaff, density);
  }
  bool joint(const Node& n0, int body, bool freejoint) {
    const Node n = freejoint ? with_defaults(Node()) : with_defaults(n0);
    int type = MJH_JNT_HINGE;
    if (freejoint) type = MJH_JNT_FREE;
    else if (const char* t = n.get("type")) {
      std::string s = t;
      if (s == "free") type = MJH_JNT_FREE; else if (s == "ball") type = MJH_JNT_BALL; else if (s == "slide") type = MJH_JNT_SLIDE; else if (s == "hinge") type = MJH_JNT_HINGE;
      else { mjh_set_error("unknown joint type " + s); return false; }
    }
    double pos[3] = {0, 0, 0}, axis[3] = {0, 0, 1}, range[2] = {0, 0}, v;
    nums(n.get("pos"), pos, 3); nums(n.get("axis"), axis, 3);
    const bool hasrange = nums(n.get("range"), range, 2) == 2;
    bool limited = false;
    if (const char* l = n.get("limited")) { std::string s = l; limited = s == "true" || (s == "auto" && hasrange); }
    else limited = autolimits && hasrange;
    if (limited && (type == MJH_JNT_HINGE || type == MJH_JNT_BALL)) { range[0] = ang(range[0]); range[1] = ang(range[1]); }   // angles for both
    if (limited && type == MJH_JNT_BALL) note += "ball joint range of " + std::string(n0.get("name") ? n0.get("name") : "?") + " read but not enforced (ball limits are not implemented); ";
    for (const char* a : {"springref", "solreflimit", "solimplimit", "solreffriction", "solimpfriction", "margin", "springdamper"})
      if (n.get(a)) note += std::string("ignored joint attribute ") + a + "; ";
    double damping = def.jnt_damping, stiffness = def.jnt_stiffness, armature = def.jnt_armature, floss = def.jnt_frictionloss, ref = 0;
    if (nums(n.get("damping"), &v, 1)) damping = v;
    if (nums(n.get("stiffness"), &v, 1)) stiffness = v;
    if (nums(n.get("armature"), &v, 1)) armature = v;
    if (nums(n.get("frictionloss"), &v, 1)) floss = v;
    if (nums(n.get("ref"), &v, 1)) ref = type == MJH_JNT_HINGE ? ang(v) : v;
    std::string name = n0.get("name") ? n0.get("name") : ("joint" + std::to_string(nameless++));
    int id = mjh_builder_add_joint(b, name.c_str(), body, type, pos, axis, limited ? range : nullptr, damping, stiffness, armature, floss, ref);
    if (id < 0) return false;
    joint_id[name] = id;
    return true;
  }
  bool body(const Node& n, int parent) {
    double pos[3] = {0, 0, 0}, quat[4] = {1, 0, 0, 0}, gc = 0;
    nums(n.get("pos"), pos, 3); orientation(n, quat); nums(n.get("gravcomp"), &gc, 1);
    if (robot_file && g_robot_gravcomp >= 0) gc = g_robot_gravcomp;   // MjSim::init_tmp overwrites it on every robot body (mj_sim.cpp:301-310)
    std::string name = n.get("name") ? n.get("name") : ("body" + std::to_string(nameless++));
    if (robot_file && parent == 0) {          // rosparam ~pose_init: position + roll / pitch / yaw of the robot's root body (mj_sim.cpp:312-335)
      auto it = g_robot_pose.find(name);
      if (it != g_robot_pose.end()) {
        for (int k = 0; k < 3; k++) pos[k] = it->second[k];
        double q[4] = {1, 0, 0, 0};          // tf2::Quaternion::setRPY: q = Rz(yaw) Ry(pitch) Rx(roll)
        for (int k = 2; k >= 0; k--) { double ax[3] = {0, 0, 0}; ax[k] = 1; double r[4], t[4]; hm::axisangle2quat(r, ax, it->second[3 + k]); hm::mulquat(t, q, r); std::memcpy(q, t, sizeof q); }
        std::memcpy(quat, q, sizeof q);
      }
    }
    int id = mjh_builder_add_body(b, name.c_str(), parent, pos, quat, gc);
    if (id < 0) return false;
    body_id[name] = id;
    if ((int)body_parent.size() <= id) body_parent.resize(id + 1, 0);
    body_parent[id] = parent;
    const bool is_mocap = n.get("mocap") && std::string(n.get("mocap")) == "true";   // MjSim::init_references: the *_ref clones (mj_sim.cpp:903)
    const std::string saved = childclass;
    if (n.get("childclass")) childclass = n.get("childclass");
    const bool ok = children(n, id);
    if (ok && is_mocap && mjh_builder_set_mocap(b, id) < 0) return false;
    // rosparam ~add_odom_joints (mj_sim.cpp:337-415): slide / hinge joints "<robot>_lin_odom_x_joint" ... appended to the
    // root body of a robot file, after the body's own children as InsertEndChild does; a linear axis is also added when
    // the other planar axis and the matching rotation are asked for (the reference's rule, :355,:365,:375)
    if (ok && robot_file && parent == 0 && g_odom_mask) {
      const unsigned m = g_odom_mask;
      const bool lx = (m & 1) || ((m & 2) && (m & 32)), ly = (m & 2) || ((m & 1) && (m & 32)), lz = (m & 4) || ((m & 1) && (m & 16));
      const bool on[6] = {lx, ly, lz, (m & 8) != 0, (m & 16) != 0, (m & 32) != 0};
      static const char* nm[6] = {"_lin_odom_x_joint", "_lin_odom_y_joint", "_lin_odom_z_joint", "_ang_odom_x_joint", "_ang_odom_y_joint", "_ang_odom_z_joint"};
      for (int k = 0; k < 6; k++) if (on[k]) {
        double axis[3] = {0, 0, 0}; axis[k % 3] = 1;
        const std::string jn = name + nm[k];
        const int jid = mjh_builder_add_joint(b, jn.c_str(), id, k < 3 ? MJH_JNT_SLIDE : MJH_JNT_HINGE, nullptr, axis, nullptr, 0, 0, 0, 0, 0);
        if (jid < 0) return false;
        joint_id[jn] = jid;
      }
    }
    childclass = saved;
    return ok;
  }
  bool children(const Node& n, int body) {
    for (auto& c : n.kids) {
      if (c->tag == "geom") geom(*c, body);
      else if (c->tag == "body") { if (!this->body(*c, body)) return false; }
      else if (c->tag == "joint" || c->tag == "freejoint") {
        if (body == 0) { mjh_set_error("joint in worldbody"); return false; }
        if (!joint(*c, body, c->tag == "freejoint")) return false;
      } else if (c->tag == "inertial") {
        double ipos[3] = {0, 0, 0}, iq[4] = {1, 0, 0, 0}, mass = 0, di[3] = {0, 0, 0};
        nums(c->get("pos"), ipos, 3); orientation(*c, iq); nums(c->get("mass"), &mass, 1);
        if (nums(c->get("diaginertia"), di, 3) != 3) { mjh_set_error("<inertial> needs diaginertia (fullinertia is not supported)"); return false; }
        mjh_builder_set_inertial(b, body, mass, ipos, iq, di);
      } else if (c->tag == "site") {
        double pos[3] = {0, 0, 0}, quat[4] = {1, 0, 0, 0};
        nums(c->get("pos"), pos, 3); orientation(*c, quat);
        std::string name = c->get("name") ? c->get("name") : ("site" + std::to_string(nameless++));
        const int id = mjh_builder_add_site(b, name.c_str(), body, pos, quat);
        if (id < 0) return false;
        site_id[name] = id;
      } else if (c->tag != "light" && c->tag != "camera") note += "ignored <" + c->tag + ">; ";
    }
    return true;
  }
  // One model from several files, the way the reference composes a world file and robot files (MjSim::init,
  // mj_sim.cpp:573-710): every file contributes its <worldbody>, <contact>, <equality> under its own <compiler> and
  // <default> settings; <option> is taken from the first file (the world) only.
  bool add(const Node& root, bool first, const std::string& dir = std::string()) {
    if (root.tag != "mujoco") { mjh_set_error("root element must be <mujoco>"); return false; }
    if (first) b = mjh_builder_create();
    robot_file = !first;      // files after the world file are robots (MjSim::init composition)
    degree = true; autolimits = false; def = Defaults(); basedir = dir; meshdir.clear(); mesh_id.clear(); classes.clear(); childclass.clear();
    mjh_option o; mjh_builder_get_option(b, &o);
    bool solver_named = false;
    // first pass: compiler / option / default (they may appear after worldbody in a file)
    for (auto& c : root.kids) {
      if (c->tag == "compiler") {
        if (const char* a = c->get("angle")) degree = std::string(a) != "radian";
        if (const char* a = c->get("autolimits")) autolimits = std::string(a) == "true";
        if (const char* a = c->get("balanceinertia")) if (std::string(a) == "true") balance = true;
        if (const char* a = c->get("meshdir")) { meshdir = a; if (!meshdir.empty() && meshdir.back() != '/') meshdir += '/'; }
        double v;
        if (nums(c->get("boundmass"), &v, 1)) bmass = std::max(bmass, v);
        if (nums(c->get("boundinertia"), &v, 1)) binertia = std::max(binertia, v);
      } else if (c->tag == "option") {
        double v, g[3];
        if (nums(c->get("timestep"), &v, 1)) o.timestep = v;
        if (nums(c->get("gravity"), g, 3) == 3) std::memcpy(o.gravity, g, sizeof g);
        if (nums(c->get("impratio"), &v, 1)) o.impratio = v;
        if (nums(c->get("iterations"), &v, 1)) o.iterations = (int)v;
        if (nums(c->get("tolerance"), &v, 1)) o.tolerance = v;
        if (nums(c->get("noslip_iterations"), &v, 1)) o.noslip_iterations = (int)v;
        if (nums(c->get("noslip_tolerance"), &v, 1)) o.noslip_tolerance = v;
        if (const char* s = c->get("solver")) { solver_named = true; if (std::string(s) != "PGS") note += std::string("solver=") + s + " -> PGS; "; }
        if (const char* s = c->get("cone")) if (std::string(s) != "pyramidal") note += std::string("cone=") + s + " -> pyramidal; ";
        if (const char* s = c->get("integrator")) if (std::string(s) != "Euler") note += std::string("integrator=") + s + " ignored: the step1 / step2 split of the reference loop always integrates with Euler (mj_main.cpp:83,108); ";
        for (auto& f : c->kids) if (f->tag == "flag") {
          if (const char* s = f->get("gravity")) if (std::string(s) == "disable") o.disableflags |= MJH_DSBL_GRAVITY;
          if (const char* s = f->get("contact")) if (std::string(s) == "disable") o.disableflags |= MJH_DSBL_CONTACT;
          if (const char* s = f->get("limit")) if (std::string(s) == "disable") o.disableflags |= MJH_DSBL_LIMIT;
          if (const char* s = f->get("warmstart")) if (std::string(s) == "disable") o.disableflags |= MJH_DSBL_WARMSTART;
        }
      } else if (c->tag == "default") {
        read_defaults(*c, "");
        for (auto& dflt : c->kids) {
          double v, t3[3];
          if (dflt->tag == "geom") {
            int nf = nums(dflt->get("friction"), t3, 3); for (int i = 0; i < nf; i++) def.geom_friction[i] = t3[i];
            if (nums(dflt->get("condim"), &v, 1)) def.geom_condim = (int)v;
            if (nums(dflt->get("contype"), &v, 1)) def.geom_contype = (int)v;
            if (nums(dflt->get("conaffinity"), &v, 1)) def.geom_conaffinity = (int)v;
            if (nums(dflt->get("density"), &v, 1)) def.geom_density = v;
          } else if (dflt->tag == "joint") {
            if (nums(dflt->get("damping"), &v, 1)) def.jnt_damping = v;
            if (nums(dflt->get("stiffness"), &v, 1)) def.jnt_stiffness = v;
            if (nums(dflt->get("armature"), &v, 1)) def.jnt_armature = v;
            if (nums(dflt->get("frictionloss"), &v, 1)) def.jnt_frictionloss = v;
          }
        }
      }
    }
    if (first) mjh_builder_set_option(b, &o);
    // said ALWAYS (no reference model names a solver, so every one of them silently ran on a different solver otherwise)
    if (first && !solver_named) note += "no <option solver>: MuJoCo's default is Newton, this engine solves the same dual problem with PGS (100 iterations, tolerance 1e-8); ";
    // <asset><mesh>: binary STL files next to the MJCF file (pr2.xml:4-23); a mesh that cannot be read is reported and
    // the geoms that use it are skipped
    for (auto& c : root.kids) if (c->tag == "asset") for (auto& a : c->kids) if (a->tag == "mesh") {
      const char* file = a->get("file");
      if (!file) { note += "skipped <mesh> without file; "; continue; }
      std::string fn = file, name;
      if (const char* nm = a->get("name")) name = nm;
      else { size_t sl = fn.find_last_of('/'), dot = fn.find_last_of('.'); name = fn.substr(sl == std::string::npos ? 0 : sl + 1, dot == std::string::npos ? std::string::npos : dot - (sl == std::string::npos ? 0 : sl + 1)); }
      if (!g_load_meshes) { note += "mesh " + name + " skipped (mesh mode 0); "; continue; }
      if (basedir.empty() && fn[0] != '/') { note += "mesh " + name + " not loaded (MJCF given as a string: no directory); "; continue; }
      const std::string path = fn[0] == '/' ? fn : basedir + meshdir + fn;
      double sc[3] = {1, 1, 1};
      nums(a->get("scale"), sc, 3);
      const int id = mjh_builder_add_mesh_stl(b, path.c_str(), sc);
      if (id < 0) { note += "mesh " + name + " not loaded (" + mjh_last_error() + "); "; continue; }
      mesh_id[name] = id;
    }
    const int body_first = std::max(1, (int)body_parent.size());      // first body this file adds
    for (auto& c : root.kids) if (c->tag == "worldbody") if (!children(*c, 0)) return false;
    // the wrapper's mujoco_compile writes <exclude> pairs between every body and its first `level` ancestors into the compiled ROBOT file
    // (disable_parent_child_collision, /root/reference/src/mujoco_compile.cpp:250-290; launch argument disable_parent_child_collision_level,
    // default 1): the same rule as a load option, for robot files that come without their own exclude list.  Applied to the bodies THIS
    // file adds, and only to robot files — the files after the world file, or the one file of a single-file load (what mujoco_compile itself
    // works on).  Where the reference's walk reaches the world it names the robot's wrapper body (`body1 = model name`,
    // mujoco_compile.cpp:272-276; add_robot_body :195-216 creates it around the file's top-level bodies): that body has no geom of its own,
    // so the pair excludes no collision and the walk simply ends here.
    if (g_pc_exclude_level > 0 && (robot_file || nfiles == 1))
      for (int id = body_first; id < (int)body_parent.size(); id++) {
        int p = id;
        for (int k = 0; k < g_pc_exclude_level; k++) { p = body_parent[p]; if (p <= 0) break; mjh_builder_add_exclude(b, p, id); }
      }
    for (auto& c : root.kids) {
      if (c->tag == "contact") {
        for (auto& e : c->kids) if (e->tag == "exclude") {
          auto i1 = body_id.find(e->get("body1") ? e->get("body1") : ""), i2 = body_id.find(e->get("body2") ? e->get("body2") : "");
          if (i1 == body_id.end() || i2 == body_id.end()) { mjh_set_error("<exclude> names an unknown body"); return false; }
          mjh_builder_add_exclude(b, i1->second, i2->second);
        }
      } else if (c->tag == "equality") {
        for (auto& e : c->kids) {
          if (e->tag == "weld" || e->tag == "connect") {
            auto i1 = body_id.find(e->get("body1") ? e->get("body1") : "");
            if (i1 == body_id.end()) { mjh_set_error("<equality><" + e->tag + "> names an unknown body1"); return false; }
            int b2 = 0;                                      // body2 omitted: the world
            if (e->get("body2")) { auto i2 = body_id.find(e->get("body2")); if (i2 == body_id.end()) { mjh_set_error("<equality><" + e->tag + "> names an unknown body2"); return false; } b2 = i2->second; }
            double anchor[3] = {0, 0, 0}, ts = 1.0;
            nums(e->get("anchor"), anchor, 3); nums(e->get("torquescale"), &ts, 1);
            if (e->get("relpose")) note += "<weld relpose> ignored (the relative pose of the reference configuration is used); ";
            const int rc = e->tag == "weld" ? mjh_builder_add_eq_weld(b, i1->second, b2, anchor, ts) : mjh_builder_add_eq_connect(b, i1->second, b2, anchor);
            if (rc < 0) return false;
            continue;
          }
          if (e->tag != "joint") { note += "ignored <equality><" + e->tag + ">; "; continue; }
          auto j1 = joint_id.find(e->get("joint1") ? e->get("joint1") : "");
          if (j1 == joint_id.end()) { mjh_set_error("<equality><joint> names an unknown joint1"); return false; }
          int j2 = -1;
          if (e->get("joint2")) { auto it = joint_id.find(e->get("joint2")); if (it == joint_id.end()) { mjh_set_error("unknown joint2"); return false; } j2 = it->second; }
          double poly[5] = {0, 1, 0, 0, 0}, t5[5]; int np = nums(e->get("polycoef"), t5, 5); for (int i = 0; i < np; i++) poly[i] = t5[i];
          mjh_builder_add_eq_joint(b, j1->second, j2, poly);
        }
      } else if (c->tag == "sensor") {
        for (auto& e : c->kids) {
          if (e->tag != "force" && e->tag != "torque") { note += "ignored <sensor><" + e->tag + "> (the reference publishes force and torque sensors only, mj_sim.cpp:973-1014); "; continue; }
          auto it = site_id.find(e->get("site") ? e->get("site") : "");
          if (it == site_id.end()) { mjh_set_error("<sensor><" + e->tag + "> names an unknown site"); return false; }
          if (mjh_builder_add_sensor(b, e->get("name"), e->tag == "force" ? MJH_SENS_FORCE : MJH_SENS_TORQUE, it->second) < 0) return false;
        }
      } else if (c->tag != "compiler" && c->tag != "option" && c->tag != "default" && c->tag != "worldbody" && c->tag != "asset" && c->tag != "visual" && c->tag != "size" && c->tag != "statistic")
        note += "ignored <" + c->tag + ">; ";
    }
    return true;
  }
  mjh_model* finish() {
    mjh_builder_set_bounds(b, std::max(bmass, g_boundmass), std::max(binertia, g_boundinertia));
    mjh_builder_set_balanceinertia(b, balance ? 1 : 0);
    mjh_model* m = mjh_builder_compile(b);
    mjh_builder_destroy(b); b = nullptr;
    // a compile error after assets were skipped is usually their consequence (a body whose only geom is a missing mesh has no mass)
    if (!m && note.find("not loaded") != std::string::npos) mjh_set_error(std::string(mjh_last_error()) + " [loader notes: " + note + "]");
    return m;
  }
  void abort() { if (b) mjh_builder_destroy(b); b = nullptr; }
  mjh_model* run(const Node& root) {
    if (!add(root, true)) { abort(); return nullptr; }
    return finish();
  }
};

thread_local std::string g_note;

}  // namespace

extern "C" mjh_model* mjh_load_mjcf_string(const char* xml) {
  g_note.clear();
  if (!xml) { mjh_set_error("null xml"); return nullptr; }
  Xml x{xml, xml + std::strlen(xml), {}};
  auto root = x.element();
  if (!root) { mjh_set_error("MJCF parse error: " + (x.err.empty() ? std::string("no root element") : x.err)); return nullptr; }
  Loader L;
  mjh_model* m = L.run(*root);
  g_note = L.note;
  return m;
}
// <include file="..."/>: the children of the included file's <mujoco> root take the place of the element (any depth; paths
// relative to the including file)
static bool expand_includes(Node& n, const std::string& dir, int depth, std::string& err) {
  if (depth > 16) { err = "<include> nesting too deep"; return false; }
  for (size_t i = 0; i < n.kids.size();) {
    Node& c = *n.kids[i];
    if (c.tag != "include") { if (!expand_includes(c, dir, depth, err)) return false; i++; continue; }
    const char* file = c.get("file");
    if (!file) { err = "<include> without file"; return false; }
    const std::string path = file[0] == '/' ? std::string(file) : dir + file;
    std::ifstream f(path);
    if (!f) { err = "cannot open included file " + path; return false; }
    std::stringstream ss; ss << f.rdbuf();
    const std::string text = ss.str();
    Xml x{text.c_str(), text.c_str() + text.size(), {}};
    auto root = x.element();
    if (!root) { err = "parse error in included file " + path + ": " + x.err; return false; }
    size_t sl = path.find_last_of('/');
    if (!expand_includes(*root, sl == std::string::npos ? std::string("./") : path.substr(0, sl + 1), depth + 1, err)) return false;
    std::vector<std::unique_ptr<Node>> repl;
    for (auto& k : root->kids) repl.push_back(std::move(k));
    n.kids.erase(n.kids.begin() + (long)i);
    for (size_t k = 0; k < repl.size(); k++) n.kids.insert(n.kids.begin() + (long)(i + k), std::move(repl[k]));
    i += repl.size();
  }
  return true;
}
static std::string dir_of(const char* path) {
  std::string p = path ? path : "";
  size_t sl = p.find_last_of('/');
  return sl == std::string::npos ? std::string("./") : p.substr(0, sl + 1);
}
extern "C" mjh_model* mjh_load_mjcf_file(const char* path) { return mjh_load_mjcf_files(&path, 1); }
extern "C" mjh_model* mjh_load_mjcf_files(const char* const* paths, int n) {
  g_note.clear();
  if (!paths || n <= 0) { mjh_set_error("mjh_load_mjcf_files: no files"); return nullptr; }
  Loader L; L.nfiles = n;
  for (int i = 0; i < n; i++) {
    std::ifstream f(paths[i] ? paths[i] : "");
    if (!f) { mjh_set_error(std::string("cannot open ") + (paths[i] ? paths[i] : "(null)")); L.abort(); return nullptr; }
    std::stringstream ss; ss << f.rdbuf();
    const std::string text = ss.str();
    Xml x{text.c_str(), text.c_str() + text.size(), {}};
    auto root = x.element();
    if (!root) { mjh_set_error(std::string("MJCF parse error in ") + paths[i] + ": " + (x.err.empty() ? std::string("no root element") : x.err)); L.abort(); return nullptr; }
    { std::string ierr;
      if (!expand_includes(*root, dir_of(paths[i]), 0, ierr)) { mjh_set_error(ierr); L.abort(); return nullptr; } }
    if (!L.add(*root, i == 0, dir_of(paths[i]))) { L.abort(); return nullptr; }
  }
  mjh_model* m = L.finish();
  g_note = L.note;
  return m;
}
// the same with the options of THIS call only (the rosparams MjSim::init_tmp reads: mj_sim.cpp:301-415,584-590)
extern "C" mjh_model* mjh_load_mjcf_files_opt(const char* const* paths, int n, const mjh_load_options* o) {
  if (!o) return mjh_load_mjcf_files(paths, n);
  const double sb = g_boundmass, si = g_boundinertia; const int sg = g_robot_gravcomp, sm = g_load_meshes; const unsigned so = g_odom_mask;
  const auto sp = g_robot_pose; const int sx = g_pc_exclude_level;
  g_pc_exclude_level = o->parent_child_exclude < 0 ? 0 : o->parent_child_exclude;
  g_boundmass = o->boundmass; g_boundinertia = o->boundinertia; g_robot_gravcomp = o->robot_gravcomp < 0 ? -1 : (o->robot_gravcomp ? 1 : 0);
  g_load_meshes = o->load_meshes != 0; g_odom_mask = o->odom_joints & 63u;
  g_robot_pose.clear();
  for (int k = 0; k < o->nrobot_pose; k++) if (o->robot_pose_body && o->robot_pose_body[k] && o->robot_pose) {
    std::array<double, 6> a; for (int q = 0; q < 6; q++) a[q] = o->robot_pose[6 * k + q];
    g_robot_pose[o->robot_pose_body[k]] = a;
  }
  mjh_model* m = mjh_load_mjcf_files(paths, n);
  g_boundmass = sb; g_boundinertia = si; g_robot_gravcomp = sg; g_load_meshes = sm; g_odom_mask = so; g_robot_pose = sp; g_pc_exclude_level = sx;
  return m;
}
extern "C" void mjh_load_default_options(mjh_load_options* o) {
  if (!o) return;
  o->boundmass = 0; o->boundinertia = 0; o->robot_gravcomp = -1; o->load_meshes = 1; o->odom_joints = 0; o->nrobot_pose = 0; o->robot_pose_body = nullptr; o->robot_pose = nullptr; o->parent_child_exclude = 0;
}
extern "C" const char* mjh_load_note(void) { return g_note.c_str(); }
extern "C" void mjh_load_set_bounds(double boundmass, double boundinertia) { g_boundmass = boundmass; g_boundinertia = boundinertia; }
extern "C" void mjh_load_set_mesh_mode(int mode) { g_load_meshes = mode != 0; }
extern "C" void mjh_load_set_robot_pose(const char* root_body, const double pose[6]) {
  if (!root_body) { g_robot_pose.clear(); return; }
  if (!pose) { g_robot_pose.erase(root_body); return; }
  std::array<double, 6> a; for (int k = 0; k < 6; k++) a[k] = pose[k];
  g_robot_pose[root_body] = a;
}
extern "C" void mjh_load_set_odom_joints(unsigned mask) { g_odom_mask = mask & 63u; }
extern "C" void mjh_load_set_parent_child_exclude(int level) { g_pc_exclude_level = level < 0 ? 0 : level; }
extern "C" void mjh_load_set_robot_gravcomp(int mode) { g_robot_gravcomp = mode < 0 ? -1 : (mode ? 1 : 0); }
