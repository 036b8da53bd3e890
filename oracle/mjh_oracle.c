/* mjh_oracle.c — TEST INFRASTRUCTURE (see mjh_oracle.h): fp64 CPU restatement of the
 * MuJoCo 2.3.7 pipeline the reference drives.  PARITY UNPINNED (library absent; no
 * reference vectors exist) — anchored by analytic KATs in tests/test_oracle_kat.py.
 *
 * Every stage cites the reference call site whose arithmetic it restates
 * (paths under /root/reference); [UPSTREAM] marks MuJoCo-internal structure
 * restated from its public documentation.
 * Where MuJoCo 2.3.7 is installed, tests/test_mujoco_reference.py steps it beside this file on the same models
 * (emitted as MJCF by tests/mjcf_emit.py); it is absent here, so the parity of this file stays UNPINNED.
 */
#include "mjh_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define MINVAL 1e-15 /* mjMINVAL, used by the wrapper at mj_hw_interface.cpp:81, mj_sim.cpp:1069 */
#define MAXVAL 1e10
#define MINIMP 0.0001
#define MAXIMP 0.9999

/* ------------------------------------------------------------------ small math */
static void zero(double* r, int n) { memset(r, 0, sizeof(double) * (size_t)n); }
static void copyv(double* r, const double* a, int n) { memcpy(r, a, sizeof(double) * (size_t)n); }
static double dot3(const double* a, const double* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static double dotn(const double* a, const double* b, int n) { double s = 0; for (int i = 0; i < n; i++) s += a[i]*b[i]; return s; }
static void cross3(double* r, const double* a, const double* b) {
  double x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2], z = a[0]*b[1] - a[1]*b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double norm3(const double* a) { return sqrt(dot3(a, a)); }
static double normalize3(double* a) {
  double n = norm3(a);
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; } else { a[0] /= n; a[1] /= n; a[2] /= n; }
  return n;
}
static void normalize4(double* q) {
  double n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
static void mulquat(double* r, const double* a, const double* b) {
  double w = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3];
  double x = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
  double y = a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1];
  double z = a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void quat2mat(double* m, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w*w + x*x - y*y - z*z; m[1] = 2*(x*y - w*z);         m[2] = 2*(x*z + w*y);
  m[3] = 2*(x*y + w*z);         m[4] = w*w - x*x + y*y - z*z; m[5] = 2*(y*z - w*x);
  m[6] = 2*(x*z - w*y);         m[7] = 2*(y*z + w*x);         m[8] = w*w - x*x - y*y + z*z;
}
static void rotvec(double* r, const double* m, const double* v) {
  double x = m[0]*v[0] + m[1]*v[1] + m[2]*v[2], y = m[3]*v[0] + m[4]*v[1] + m[5]*v[2], z = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void rotvecT(double* r, const double* m, const double* v) {
  double x = m[0]*v[0] + m[3]*v[1] + m[6]*v[2], y = m[1]*v[0] + m[4]*v[1] + m[7]*v[2], z = m[2]*v[0] + m[5]*v[1] + m[8]*v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void axisangle2quat(double* q, const double* axis, double angle) {
  double s = sin(0.5 * angle); q[0] = cos(0.5 * angle); q[1] = axis[0]*s; q[2] = axis[1]*s; q[3] = axis[2]*s;
}
/* q <- q * exp(h*w/2), w in the local (body) frame [UPSTREAM mju_quatIntegrate] */
static void quat_integrate(double* q, const double* w, double h) {
  double ax[3] = {w[0], w[1], w[2]};
  double n = norm3(ax);
  if (n < MINVAL) { normalize4(q); return; }
  ax[0] /= n; ax[1] /= n; ax[2] /= n;
  double dq[4], r[4]; axisangle2quat(dq, ax, h * n);
  normalize4(q); mulquat(r, q, dq); copyv(q, r, 4); normalize4(q);
}
/* spatial vectors are (rotational 3, translational 3) */
static void cross_motion(double* r, const double* v, const double* x) { /* v x x */
  double a[3], b[3], c[3];
  cross3(a, v, x); cross3(b, v, x + 3); cross3(c, v + 3, x);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void cross_force(double* r, const double* v, const double* f) { /* v x* f */
  double a[3], b[3], c[3];
  cross3(a, v, f); cross3(b, v + 3, f + 3); cross3(c, v, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
/* 10-number spatial inertia about a point offset `r` from the COM: [Ixx Iyy Izz Ixy Ixz Iyz, m*r(3), m] */
static void inert_com(double* res, const double* diagI, const double* mat, const double* off, double mass) {
  double I[9];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
    double v = 0; for (int k = 0; k < 3; k++) v += mat[3*r+k] * diagI[k] * mat[3*c+k];
    I[3*r+c] = v;
  }
  double d2 = dot3(off, off);
  res[0] = I[0] + mass * (d2 - off[0]*off[0]); res[1] = I[4] + mass * (d2 - off[1]*off[1]);
  res[2] = I[8] + mass * (d2 - off[2]*off[2]);
  res[3] = I[1] - mass * off[0]*off[1]; res[4] = I[2] - mass * off[0]*off[2]; res[5] = I[5] - mass * off[1]*off[2];
  res[6] = mass * off[0]; res[7] = mass * off[1]; res[8] = mass * off[2]; res[9] = mass;
}
static void mul_inert_vec(double* res, const double* i, const double* v) {
  double r0 = i[0]*v[0] + i[3]*v[1] + i[4]*v[2] - i[8]*v[4] + i[7]*v[5];
  double r1 = i[3]*v[0] + i[1]*v[1] + i[5]*v[2] + i[8]*v[3] - i[6]*v[5];
  double r2 = i[4]*v[0] + i[5]*v[1] + i[2]*v[2] - i[7]*v[3] + i[6]*v[4];
  double r3 = i[8]*v[1] - i[7]*v[2] + i[9]*v[3];
  double r4 = i[6]*v[2] - i[8]*v[0] + i[9]*v[4];
  double r5 = i[7]*v[0] - i[6]*v[1] + i[9]*v[5];
  res[0] = r0; res[1] = r1; res[2] = r2; res[3] = r3; res[4] = r4; res[5] = r5;
}

/* ------------------------------------------------------------------ data */
static double* dalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }
static int* ialloc(size_t n) { return (int*)calloc(n ? n : 1, sizeof(int)); }

static const double* p_geom_size(const orc_data* d) { return d->geom_size ? d->geom_size : d->m->geom_size; }
static const double* p_geom_rbound(const orc_data* d) { return d->geom_rbound ? d->geom_rbound : d->m->geom_rbound; }
static const double* p_body_mass(const orc_data* d) { return d->body_mass ? d->body_mass : d->m->body_mass; }
static const double* p_body_inertia(const orc_data* d) { return d->body_inertia ? d->body_inertia : d->m->body_inertia; }
static const double* p_body_invweight0(const orc_data* d) { return d->body_invweight0 ? d->body_invweight0 : d->m->body_invweight0; }
static const double* p_dof_invweight0(const orc_data* d) { return d->dof_invweight0 ? d->dof_invweight0 : d->m->dof_invweight0; }

/* mj_makeData, mj_sim.cpp:816,835 + init_malloc, mj_sim.cpp:563-571 */
orc_data* orc_make_data(const mjh_model* m) {
  orc_data* d = (orc_data*)calloc(1, sizeof(orc_data));
  int nq = m->nq, nv = m->nv, nb = m->nbody, nj = m->njnt, ng = m->ngeom, ne = m->maxefc, nc = m->maxcon;
  d->m = m;
  d->qpos = dalloc(nq); d->qvel = dalloc(nv); d->qacc = dalloc(nv); d->qacc_warmstart = dalloc(nv);
  d->qfrc_applied = dalloc(nv); d->qfrc_bias = dalloc(nv); d->qfrc_passive = dalloc(nv); d->qfrc_smooth = dalloc(nv);
  d->qacc_smooth = dalloc(nv); d->qfrc_constraint = dalloc(nv); d->qfrc_inverse = dalloc(nv);
  d->xpos = dalloc(3*nb); d->xquat = dalloc(4*nb); d->xmat = dalloc(9*nb); d->xipos = dalloc(3*nb); d->ximat = dalloc(9*nb);
  d->xanchor = dalloc(3*nj); d->xaxis = dalloc(3*nj); d->geom_xpos = dalloc(3*ng); d->geom_xmat = dalloc(9*ng);
  d->subtree_com = dalloc(3*nb); d->cinert = dalloc(10*nb); d->crb = dalloc(10*nb); d->cdof = dalloc(6*nv);
  d->cdof_dot = dalloc(6*nv); d->cvel = dalloc(6*nb);
  d->qM = dalloc(m->nM); d->qLD = dalloc(m->nM); d->qLDiagInv = dalloc(nv);
  d->contact = (orc_contact*)calloc(nc ? nc : 1, sizeof(orc_contact));
  d->efc_type = ialloc(ne); d->efc_id = ialloc(ne);
  d->efc_J = dalloc((size_t)ne*nv); d->efc_pos = dalloc(ne); d->efc_margin = dalloc(ne); d->efc_frictionloss = dalloc(ne);
  d->efc_diagApprox = dalloc(ne); d->efc_R = dalloc(ne); d->efc_D = dalloc(ne); d->efc_KBIP = dalloc(4*(size_t)ne);
  d->efc_vel = dalloc(ne); d->efc_aref = dalloc(ne); d->efc_b = dalloc(ne); d->efc_force = dalloc(ne);
  d->efc_AR = dalloc((size_t)ne*ne);
  d->ddq = dalloc(nv); d->dq = dalloc(nv); d->tau = dalloc(nv); d->controlled = ialloc(nv);
  d->initial_qpos = dalloc(nq);
  for (int i = 0; i < 6; i++) d->scr_nv[i] = dalloc(nv);
  d->scr_nM = dalloc(m->nM);
  /* per-step scratch that used to be malloc'ed inside the step (Jacobian rows of one point; the Gauss-Seidel order builder) */
  d->scr_jac = dalloc(12 * (size_t)(nv ? nv : 1));
  d->scr_int = ialloc((size_t)(ne + 1) * 12);
  d->scr_key = (long long*)calloc((size_t)ne + 1, sizeof(long long));
  d->scr_nz = ialloc((size_t)ne * (nv ? nv : 1));
  for (int i = 0; i < 3; i++) d->scr_efc[i] = dalloc(ne);
  d->scr_B = dalloc((size_t)(ne > 6 ? ne : 6)*nv);
  for (int i = 0; i < 3; i++) d->scr_body6[i] = dalloc(6*nb);
  for (int i = 0; i < 3; i++) { d->odom_lin[i] = -1; d->odom_ang[i] = -1; d->odom_angq[i] = -1; }
  d->xfrc_applied = dalloc(6*nb); d->mocap_pos = dalloc(3*(size_t)m->nmocap); d->mocap_quat = dalloc(4*(size_t)m->nmocap);
  d->sensordata = dalloc(m->nsensordata); d->site_xpos = dalloc(3*(size_t)m->nsite); d->site_xmat = dalloc(9*(size_t)m->nsite);
  d->cacc = dalloc(6*nb); d->cfrc_int = dalloc(6*nb); d->cfrc_ext = dalloc(6*nb);
  for (int b = 0; b < nb; b++) if (m->body_mocapid && m->body_mocapid[b] >= 0) {   /* mocap pose starts at the body's model pose */
    copyv(d->mocap_pos + 3*m->body_mocapid[b], m->body_pos + 3*b, 3); copyv(d->mocap_quat + 4*m->body_mocapid[b], m->body_quat + 4*b, 4);
  }
  copyv(d->initial_qpos, m->qpos0, nq);
  orc_reset(d);
  return d;
}

void orc_free_data(orc_data* d) {
  if (!d) return;
  double* ps[] = {d->qpos, d->qvel, d->qacc, d->qacc_warmstart, d->qfrc_applied, d->qfrc_bias, d->qfrc_passive, d->qfrc_smooth,
    d->qacc_smooth, d->qfrc_constraint, d->qfrc_inverse, d->xpos, d->xquat, d->xmat, d->xipos, d->ximat, d->xanchor, d->xaxis,
    d->geom_xpos, d->geom_xmat, d->subtree_com, d->cinert, d->crb, d->cdof, d->cdof_dot, d->cvel, d->qM, d->qLD, d->qLDiagInv,
    d->efc_J, d->efc_pos, d->efc_margin, d->efc_frictionloss, d->efc_diagApprox, d->efc_R, d->efc_D, d->efc_KBIP, d->efc_vel,
    d->efc_aref, d->efc_b, d->efc_force, d->efc_AR, d->ddq, d->dq, d->tau, d->initial_qpos, d->scr_nM, d->scr_B,
    d->geom_size, d->geom_rbound, d->body_mass, d->body_inertia, d->body_invweight0, d->dof_invweight0,
    d->xfrc_applied, d->mocap_pos, d->mocap_quat, d->sensordata, d->site_xpos, d->site_xmat, d->cacc, d->cfrc_int, d->cfrc_ext};
  for (size_t i = 0; i < sizeof(ps)/sizeof(ps[0]); i++) free(ps[i]);
  for (int i = 0; i < 6; i++) free(d->scr_nv[i]);
  for (int i = 0; i < 3; i++) { free(d->scr_efc[i]); free(d->scr_body6[i]); }
  free(d->scr_jac); free(d->scr_int); free(d->scr_key); free(d->scr_nz);
  free(d->contact); free(d->efc_type); free(d->efc_id); free(d->controlled); free(d->pd_target);
  free(d);
}

void orc_set_env_param(orc_data* d, int which, const double* v) {
  const mjh_model* m = d->m;
  double** slot = NULL; int n = 0;
  switch (which) {
    case MJH_EP_GEOM_SIZE: slot = &d->geom_size; n = 3*m->ngeom; break;
    case MJH_EP_GEOM_RBOUND: slot = &d->geom_rbound; n = m->ngeom; break;
    case MJH_EP_BODY_MASS: slot = &d->body_mass; n = m->nbody; break;
    case MJH_EP_BODY_INERTIA: slot = &d->body_inertia; n = 3*m->nbody; break;
    case MJH_EP_BODY_INVWEIGHT0: slot = &d->body_invweight0; n = 2*m->nbody; break;
    case MJH_EP_DOF_INVWEIGHT0: slot = &d->dof_invweight0; n = m->nv; break;
    default: return;
  }
  if (!*slot) *slot = dalloc(n);
  copyv(*slot, v, n);
}

/* MjRos::reset_robot, mj_ros.cpp:569-609: fresh data, qpos <- initial, zero velocities */
void orc_reset(orc_data* d) {
  const mjh_model* m = d->m;
  d->time = 0;
  copyv(d->qpos, d->initial_qpos, m->nq);
  zero(d->qvel, m->nv); zero(d->qacc, m->nv); zero(d->qacc_warmstart, m->nv); zero(d->qfrc_applied, m->nv);
  zero(d->ddq, m->nv); zero(d->dq, m->nv); zero(d->tau, m->nv);
  d->ncon = 0; d->nefc = 0; d->solver_iter = 0;
}

/* ------------------------------------------------------------------ position stage */
/* FK [UPSTREAM mj_kinematics], first stage of mj_step1 (mj_main.cpp:83) */
void orc_kinematics(orc_data* d) {
  const mjh_model* m = d->m;
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0; zero(d->xpos, 3);
  quat2mat(d->xmat, d->xquat); zero(d->xipos, 3); quat2mat(d->ximat, d->xquat);
  for (int i = 1; i < m->nbody; i++) {
    double xpos[3], xquat[4];
    int jn = m->body_jntnum[i], ja = m->body_jntadr[i];
    if (jn == 1 && m->jnt_type[ja] == MJH_JNT_FREE) {
      int qa = m->jnt_qposadr[ja];
      copyv(xpos, d->qpos + qa, 3);
      normalize4(d->qpos + qa + 3);
      copyv(xquat, d->qpos + qa + 3, 4);
      copyv(d->xanchor + 3*ja, xpos, 3);
      double ax[3] = {0, 0, 1}; copyv(d->xaxis + 3*ja, ax, 3);
    } else if (m->body_mocapid && m->body_mocapid[i] >= 0) {   /* [UPSTREAM mj_kinematics]: mocap bodies take mocap_pos / mocap_quat */
      copyv(xpos, d->mocap_pos + 3*m->body_mocapid[i], 3);
      copyv(xquat, d->mocap_quat + 4*m->body_mocapid[i], 4);
    } else {
      int p = m->body_parentid[i];
      double t[3]; rotvec(t, d->xmat + 9*p, m->body_pos + 3*i);
      for (int k = 0; k < 3; k++) xpos[k] = d->xpos[3*p+k] + t[k];
      mulquat(xquat, d->xquat + 4*p, m->body_quat + 4*i);
      for (int j = ja; j < ja + jn; j++) {
        int qa = m->jnt_qposadr[j];
        double mat[9], vec[3], xanchor[3], xaxis[3];
        quat2mat(mat, xquat);
        rotvec(vec, mat, m->jnt_pos + 3*j);
        for (int k = 0; k < 3; k++) xanchor[k] = xpos[k] + vec[k];
        rotvec(xaxis, mat, m->jnt_axis + 3*j);
        switch (m->jnt_type[j]) {
          case MJH_JNT_SLIDE:
            for (int k = 0; k < 3; k++) xpos[k] += xaxis[k] * (d->qpos[qa] - m->qpos0[qa]);
            break;
          case MJH_JNT_BALL:
          case MJH_JNT_HINGE: {
            double qloc[4], r[4];
            if (m->jnt_type[j] == MJH_JNT_BALL) { normalize4(d->qpos + qa); copyv(qloc, d->qpos + qa, 4); }
            else axisangle2quat(qloc, m->jnt_axis + 3*j, d->qpos[qa] - m->qpos0[qa]);
            mulquat(r, xquat, qloc); copyv(xquat, r, 4);
            /* keep the anchor fixed: xpos = xanchor - R_new * jnt_pos */
            quat2mat(mat, xquat); rotvec(vec, mat, m->jnt_pos + 3*j);
            for (int k = 0; k < 3; k++) xpos[k] = xanchor[k] - vec[k];
          } break;
          default: break;
        }
        copyv(d->xanchor + 3*j, xanchor, 3); copyv(d->xaxis + 3*j, xaxis, 3);
      }
    }
    normalize4(xquat);
    copyv(d->xpos + 3*i, xpos, 3); copyv(d->xquat + 4*i, xquat, 4);
    quat2mat(d->xmat + 9*i, xquat);
    double t[3], q[4];
    rotvec(t, d->xmat + 9*i, m->body_ipos + 3*i);
    for (int k = 0; k < 3; k++) d->xipos[3*i+k] = xpos[k] + t[k];
    mulquat(q, xquat, m->body_iquat + 4*i); quat2mat(d->ximat + 9*i, q);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double t[3], q[4];
    rotvec(t, d->xmat + 9*b, m->geom_pos + 3*g);
    for (int k = 0; k < 3; k++) d->geom_xpos[3*g+k] = d->xpos[3*b+k] + t[k];
    mulquat(q, d->xquat + 4*b, m->geom_quat + 4*g); quat2mat(d->geom_xmat + 9*g, q);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    double t[3], q[4];
    rotvec(t, d->xmat + 9*b, m->site_pos + 3*s);
    for (int k = 0; k < 3; k++) d->site_xpos[3*s+k] = d->xpos[3*b+k] + t[k];
    mulquat(q, d->xquat + 4*b, m->site_quat + 4*s); quat2mat(d->site_xmat + 9*s, q);
  }
}

/* subtree COM, COM-based inertias and motion axes [UPSTREAM mj_comPos] */
void orc_com_pos(orc_data* d) {
  const mjh_model* m = d->m;
  const double* mass = p_body_mass(d); const double* inertia = p_body_inertia(d);
  int nb = m->nbody;
  double* smass = d->scr_body6[0];
  for (int i = 0; i < nb; i++) {
    smass[i] = mass[i];
    for (int k = 0; k < 3; k++) d->subtree_com[3*i+k] = mass[i] * d->xipos[3*i+k];
  }
  for (int i = nb - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    smass[p] += smass[i];
    for (int k = 0; k < 3; k++) d->subtree_com[3*p+k] += d->subtree_com[3*i+k];
  }
  for (int i = 0; i < nb; i++) {
    if (smass[i] < MINVAL) copyv(d->subtree_com + 3*i, d->xipos + 3*i, 3);
    else for (int k = 0; k < 3; k++) d->subtree_com[3*i+k] /= smass[i];
  }
  zero(d->cinert, 10);
  for (int i = 1; i < nb; i++) {
    double off[3]; const double* com = d->subtree_com + 3*m->body_rootid[i];
    for (int k = 0; k < 3; k++) off[k] = d->xipos[3*i+k] - com[k];
    inert_com(d->cinert + 10*i, inertia + 3*i, d->ximat + 9*i, off, mass[i]);
  }
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    const double* com = d->subtree_com + 3*m->body_rootid[b];
    double off[3]; for (int k = 0; k < 3; k++) off[k] = com[k] - d->xanchor[3*j+k];
    switch (m->jnt_type[j]) {
      case MJH_JNT_FREE:
        for (int a = 0; a < 3; a++) { zero(d->cdof + 6*(da+a), 6); d->cdof[6*(da+a) + 3 + a] = 1; }
        da += 3; /* fallthrough: rotations about the body axes */
      case MJH_JNT_BALL:
        for (int a = 0; a < 3; a++) {
          double ax[3] = {d->xmat[9*b + a], d->xmat[9*b + 3 + a], d->xmat[9*b + 6 + a]};
          copyv(d->cdof + 6*(da+a), ax, 3); cross3(d->cdof + 6*(da+a) + 3, ax, off);
        }
        break;
      case MJH_JNT_SLIDE:
        zero(d->cdof + 6*da, 3); copyv(d->cdof + 6*da + 3, d->xaxis + 3*j, 3);
        break;
      case MJH_JNT_HINGE:
        copyv(d->cdof + 6*da, d->xaxis + 3*j, 3); cross3(d->cdof + 6*da + 3, d->xaxis + 3*j, off);
        break;
    }
  }
}

/* composite rigid body algorithm [UPSTREAM mj_crb] */
void orc_crb(orc_data* d) {
  const mjh_model* m = d->m;
  copyv(d->crb, d->cinert, 10*m->nbody);
  for (int i = m->nbody - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    if (p > 0) for (int k = 0; k < 10; k++) d->crb[10*p+k] += d->crb[10*i+k];
  }
  zero(d->qM, m->nM);
  for (int i = 0; i < m->nv; i++) {
    double buf[6];
    int adr = m->dof_Madr[i];
    mul_inert_vec(buf, d->crb + 10*m->dof_bodyid[i], d->cdof + 6*i);
    d->qM[adr] += m->dof_armature[i];
    for (int j = i; j >= 0; j = m->dof_parentid[j]) d->qM[adr++] += dotn(d->cdof + 6*j, buf, 6);
  }
}

/* sparse L^T D L factorisation along dof_parentid chains [UPSTREAM mj_factorM] */
static void factor_i(const mjh_model* m, double* qLD, double* qLDiagInv) {
  for (int k = m->nv - 1; k >= 0; k--) {
    int Madr_kk = m->dof_Madr[k], Madr_ki = Madr_kk + 1;
    double Mkk = qLD[Madr_kk];
    for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) {
      double tmp = qLD[Madr_ki] / Mkk;
      int cnt = 0;
      for (int j = i; j >= 0; j = m->dof_parentid[j]) { qLD[m->dof_Madr[i] + cnt] -= tmp * qLD[Madr_ki + cnt]; cnt++; }
      qLD[Madr_ki] = tmp;
      Madr_ki++;
    }
    qLDiagInv[k] = 1.0 / Mkk;
  }
}
static void solve_ld(const mjh_model* m, double* x, const double* qLD, const double* qLDiagInv) {
  for (int k = m->nv - 1; k >= 0; k--) {
    if (x[k] == 0) continue;
    int adr = m->dof_Madr[k] + 1;
    for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) x[i] -= qLD[adr++] * x[k];
  }
  for (int k = 0; k < m->nv; k++) x[k] *= qLDiagInv[k];
  for (int k = 0; k < m->nv; k++) {
    int adr = m->dof_Madr[k] + 1;
    for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) x[k] -= qLD[adr++] * x[i];
  }
}
void orc_factor_m(orc_data* d) { copyv(d->qLD, d->qM, d->m->nM); factor_i(d->m, d->qLD, d->qLDiagInv); }
void orc_solve_m(const orc_data* d, double* x) { solve_ld(d->m, x, d->qLD, d->qLDiagInv); }

/* mj_mulM as called by MjSim::controller, mj_sim.cpp:1057 */
void orc_mul_m(const orc_data* d, double* res, const double* vec) {
  const mjh_model* m = d->m;
  zero(res, m->nv);
  for (int i = 0; i < m->nv; i++) {
    int adr = m->dof_Madr[i];
    res[i] += d->qM[adr] * vec[i];
    int k = 1;
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) {
      res[i] += d->qM[adr + k] * vec[j]; res[j] += d->qM[adr + k] * vec[i]; k++;
    }
  }
}

/* ------------------------------------------------------------------ collision */
static void make_frame(double* f) { /* [UPSTREAM mju_makeFrame]: normal in f[0..2], tangent seed in f[3..5] */
  normalize3(f);
  if (norm3(f + 3) < 0.5) {
    f[3] = f[4] = f[5] = 0;
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  }
  double dp = dot3(f, f + 3);
  for (int k = 0; k < 3; k++) f[3+k] -= dp * f[k];
  normalize3(f + 3);
  cross3(f + 6, f, f + 3);
}

typedef struct { double dist, pos[3], n[3]; } rawcon;

static int c_plane_sphere(const double* pp, const double* pm, const double* c, double r, double margin, rawcon* out) {
  double n[3] = {pm[2], pm[5], pm[8]}, t[3] = {c[0]-pp[0], c[1]-pp[1], c[2]-pp[2]};
  double dist = dot3(t, n) - r;
  if (dist > margin) return 0;
  out->dist = dist; copyv(out->n, n, 3);
  for (int k = 0; k < 3; k++) out->pos[k] = c[k] - n[k] * (r + 0.5 * dist);
  return 1;
}
static int c_plane_capsule(const double* pp, const double* pm, const double* c, const double* cm, const double* size, double margin, rawcon* out) {
  double ax[3] = {cm[2]*size[1], cm[5]*size[1], cm[8]*size[1]}, e[3];
  int n = 0;
  for (int k = 0; k < 3; k++) e[k] = c[k] + ax[k];
  n += c_plane_sphere(pp, pm, e, size[0], margin, out + n);
  for (int k = 0; k < 3; k++) e[k] = c[k] - ax[k];
  n += c_plane_sphere(pp, pm, e, size[0], margin, out + n);
  return n;
}
/* [UPSTREAM mjc_PlaneCylinder]: deepest rim point of the cap facing the plane, the same rim direction on the other cap,
 * and two more points of the near cap at +-120 degrees (a triangle under a standing cylinder); at most 4 */
static int c_plane_cylinder(const double* pp, const double* pm, const double* c, const double* cm, const double* size, double margin, rawcon* out) {
  double n[3] = {pm[2], pm[5], pm[8]}, ax[3] = {cm[2], cm[5], cm[8]}, t[3] = {c[0]-pp[0], c[1]-pp[1], c[2]-pp[2]};
  const double r = size[0], h = size[1];
  double prjaxis = dot3(n, ax);
  if (prjaxis > 0) { for (int k = 0; k < 3; k++) ax[k] = -ax[k]; prjaxis = -prjaxis; }   /* axis points towards the plane */
  const double dist0 = dot3(t, n);
  double vec[3], len2 = 0;
  for (int k = 0; k < 3; k++) { vec[k] = ax[k] * prjaxis - n[k]; len2 += vec[k] * vec[k]; }   /* -normal without its axial part */
  /* (the parallel-cap test uses 1e-10 instead of mjMINVAL^2 so that the fp32 engine, whose rotation matrices carry
   * 1e-7 noise, takes the same branch as this fp64 restatement) */
  if (len2 >= 1e-10) { const double sc = r / sqrt(len2); for (int k = 0; k < 3; k++) vec[k] *= sc; }
  else { vec[0] = cm[0] * r; vec[1] = cm[3] * r; vec[2] = cm[6] * r; }                       /* cap parallel to the plane: cylinder x axis */
  const double prjvec = dot3(vec, n);
  for (int k = 0; k < 3; k++) ax[k] *= h;
  prjaxis *= h;
  int cnt = 0;
  double d1 = dist0 + prjaxis + prjvec;
  if (d1 > margin) return 0;
  out[cnt].dist = d1; copyv(out[cnt].n, n, 3);
  for (int k = 0; k < 3; k++) out[cnt].pos[k] = c[k] + vec[k] + ax[k] - n[k] * 0.5 * d1;
  cnt++;
  double d2 = dist0 - prjaxis + prjvec;
  if (d2 <= margin) {
    out[cnt].dist = d2; copyv(out[cnt].n, n, 3);
    for (int k = 0; k < 3; k++) out[cnt].pos[k] = c[k] + vec[k] - ax[k] - n[k] * 0.5 * d2;
    cnt++;
  }
  double d3 = dist0 + prjaxis - 0.5 * prjvec;
  if (d3 <= margin) {
    double v1[3]; cross3(v1, vec, ax);
    double l = sqrt(dot3(v1, v1));
    if (l > MINVAL) {
      const double sc = r * 0.8660254037844386 / l;
      for (int k = 0; k < 3; k++) v1[k] *= sc;
      for (int sgn = 1; sgn >= -1; sgn -= 2) {
        out[cnt].dist = d3; copyv(out[cnt].n, n, 3);
        for (int k = 0; k < 3; k++) out[cnt].pos[k] = c[k] + sgn * v1[k] + ax[k] - 0.5 * vec[k] - n[k] * 0.5 * d3;
        cnt++;
      }
    }
  }
  return cnt;
}
/* [UPSTREAM mjc_PlaneBox]: corners below the centre, at most 4 */
/* [UPSTREAM mjc_PlaneConvex for an ellipsoid]: the support point of the ellipsoid against the plane normal */
static int c_plane_ellipsoid(const double* pp, const double* pm, const double* c, const double* em, const double* size, double margin, rawcon* out) {
  double n[3] = {pm[2], pm[5], pm[8]}, nn[3] = {-n[0], -n[1], -n[2]}, dl[3], pl[3], pw[3], w[3];
  rotvecT(dl, em, nn);
  for (int k = 0; k < 3; k++) w[k] = size[k]*size[k]*dl[k];
  double den = sqrt(fmax(w[0]*dl[0] + w[1]*dl[1] + w[2]*dl[2], 1e-30));
  for (int k = 0; k < 3; k++) pl[k] = w[k] / den;
  rotvec(pw, em, pl);
  double t[3];
  for (int k = 0; k < 3; k++) { pw[k] += c[k]; t[k] = pw[k] - pp[k]; }
  double dist = dot3(t, n);
  if (dist > margin) return 0;
  out->dist = dist; copyv(out->n, n, 3);
  for (int k = 0; k < 3; k++) out->pos[k] = pw[k] - n[k] * 0.5 * dist;
  return 1;
}
/* plane - convex mesh (this project's definition; [UPSTREAM mjc_PlaneConvex] also returns the support vertex first and
 * adds up to three more): vertices below the margin; the deepest one, the one farthest from it, the one farthest from
 * the line through those two, and the one farthest on the other side of that line; at most 4 */
#define PLANE_MESH_EPS2 1e-8
static int c_plane_mesh(const double* pp, const double* pm, const double* c, const double* mm, const double* vert, int nvert,
                        double margin, rawcon* out) {
  double n[3] = {pm[2], pm[5], pm[8]}, nl[3], t[3] = {c[0]-pp[0], c[1]-pp[1], c[2]-pp[2]};
  rotvecT(nl, mm, n);
  const double d0 = dot3(t, n);
  int pick[4], cnt = 0;
  double best = 1e300; int bi = -1;
  for (int i = 0; i < nvert; i++) { double di = d0 + dot3(vert + 3*i, nl); if (di < best) { best = di; bi = i; } }
  if (bi < 0 || best > margin) return 0;
  pick[cnt++] = bi;
  const double* v1 = vert + 3*bi;
  best = PLANE_MESH_EPS2; bi = -1;
  for (int i = 0; i < nvert; i++) {
    if (d0 + dot3(vert + 3*i, nl) > margin) continue;
    double e[3] = {vert[3*i]-v1[0], vert[3*i+1]-v1[1], vert[3*i+2]-v1[2]}, l2 = dot3(e, e);
    if (l2 > best) { best = l2; bi = i; }
  }
  if (bi >= 0) {
    pick[cnt++] = bi;
    const double* v2 = vert + 3*bi;
    double e12[3] = {v2[0]-v1[0], v2[1]-v1[1], v2[2]-v1[2]}, side[3];
    cross3(side, e12, nl);                       /* in-plane direction across the line */
    double bpos = sqrt(PLANE_MESH_EPS2 * dot3(e12, e12)), bneg = bpos; int ipos = -1, ineg = -1;
    for (int i = 0; i < nvert; i++) {
      if (d0 + dot3(vert + 3*i, nl) > margin) continue;
      double e[3] = {vert[3*i]-v1[0], vert[3*i+1]-v1[1], vert[3*i+2]-v1[2]}, sd = dot3(e, side);
      if (sd > bpos) { bpos = sd; ipos = i; }
      if (-sd > bneg) { bneg = -sd; ineg = i; }
    }
    if (ipos >= 0 && ineg >= 0) { if (bpos >= bneg) { pick[cnt++] = ipos; pick[cnt++] = ineg; } else { pick[cnt++] = ineg; pick[cnt++] = ipos; } }
    else if (ipos >= 0) pick[cnt++] = ipos;
    else if (ineg >= 0) pick[cnt++] = ineg;
  }
  for (int q = 0; q < cnt; q++) {
    double w[3];
    rotvec(w, mm, vert + 3*pick[q]);
    double di = d0 + dot3(vert + 3*pick[q], nl);
    out[q].dist = di; copyv(out[q].n, n, 3);
    for (int k = 0; k < 3; k++) out[q].pos[k] = c[k] + w[k] - n[k] * 0.5 * di;
  }
  return cnt;
}
static int c_plane_box(const double* pp, const double* pm, const double* c, const double* bm, const double* size, double margin, rawcon* out) {
  double n[3] = {pm[2], pm[5], pm[8]}, t[3] = {c[0]-pp[0], c[1]-pp[1], c[2]-pp[2]};
  double dist = dot3(t, n);
  int cnt = 0;
  for (int i = 0; i < 8; i++) {
    double v[3] = {(i & 1) ? size[0] : -size[0], (i & 2) ? size[1] : -size[1], (i & 4) ? size[2] : -size[2]}, corner[3];
    rotvec(corner, bm, v);
    double ldist = dot3(n, corner);
    if (dist + ldist > margin || ldist > 0) continue;
    out[cnt].dist = dist + ldist; copyv(out[cnt].n, n, 3);
    for (int k = 0; k < 3; k++) out[cnt].pos[k] = corner[k] + c[k] - n[k] * 0.5 * out[cnt].dist;
    if (++cnt >= 4) return 4;
  }
  return cnt;
}
static int c_sphere_sphere(const double* c1, double r1, const double* c2, double r2, double margin, rawcon* out) {
  double t[3] = {c2[0]-c1[0], c2[1]-c1[1], c2[2]-c1[2]};
  double len = norm3(t), dist = len - r1 - r2;
  if (dist > margin) return 0;
  if (len < MINVAL) { t[0] = 1; t[1] = 0; t[2] = 0; } else { t[0] /= len; t[1] /= len; t[2] /= len; }
  out->dist = dist; copyv(out->n, t, 3);
  for (int k = 0; k < 3; k++) out->pos[k] = c1[k] + t[k] * (r1 + 0.5 * dist);
  return 1;
}
static int c_sphere_capsule(const double* c1, double r1, const double* c2, const double* m2, const double* s2, double margin, rawcon* out) {
  double ax[3] = {m2[2], m2[5], m2[8]}, t[3] = {c1[0]-c2[0], c1[1]-c2[1], c1[2]-c2[2]};
  double x = dot3(ax, t); if (x > s2[1]) x = s2[1]; if (x < -s2[1]) x = -s2[1];
  double p[3] = {c2[0] + ax[0]*x, c2[1] + ax[1]*x, c2[2] + ax[2]*x};
  return c_sphere_sphere(c1, r1, p, s2[0], margin, out);
}
static int c_capsule_capsule(const double* c1, const double* m1, const double* s1, const double* c2, const double* m2, const double* s2, double margin, rawcon* out) {
  double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]}, dif[3] = {c1[0]-c2[0], c1[1]-c2[1], c1[2]-c2[2]};
  double ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif);
  double det = ma * mc - mb * mb, x1, x2;
  if (fabs(det) >= 1e-12) {
    x1 = (mc * u - mb * v) / det; x2 = (ma * v - mb * u) / det;
    if (x1 > s1[1]) { x1 = s1[1]; x2 = (v - mb * x1) / mc; } else if (x1 < -s1[1]) { x1 = -s1[1]; x2 = (v - mb * x1) / mc; }
    if (x2 > s2[1]) { x2 = s2[1]; x1 = (u - mb * x2) / ma; if (x1 > s1[1]) x1 = s1[1]; else if (x1 < -s1[1]) x1 = -s1[1]; }
    else if (x2 < -s2[1]) { x2 = -s2[1]; x1 = (u - mb * x2) / ma; if (x1 > s1[1]) x1 = s1[1]; else if (x1 < -s1[1]) x1 = -s1[1]; }
  } else { /* parallel: midpoint of the overlap, one contact */
    x1 = 0; x2 = v / mc; if (x2 > s2[1]) x2 = s2[1]; if (x2 < -s2[1]) x2 = -s2[1];
    x1 = (u - mb * x2) / ma; if (x1 > s1[1]) x1 = s1[1]; if (x1 < -s1[1]) x1 = -s1[1];
  }
  double p1[3], p2[3];
  for (int k = 0; k < 3; k++) { p1[k] = c1[k] + a1[k]*x1; p2[k] = c2[k] + a2[k]*x2; }
  return c_sphere_sphere(p1, s1[0], p2, s2[0], margin, out);
}
static int c_sphere_box(const double* c1, double r1, const double* c2, const double* m2, const double* s2, double margin, rawcon* out) {
  double t[3] = {c1[0]-c2[0], c1[1]-c2[1], c1[2]-c2[2]}, loc[3], clamped[3];
  rotvecT(loc, m2, t);
  int inside = 1;
  for (int k = 0; k < 3; k++) {
    clamped[k] = loc[k];
    if (clamped[k] > s2[k]) { clamped[k] = s2[k]; inside = 0; } else if (clamped[k] < -s2[k]) { clamped[k] = -s2[k]; inside = 0; }
  }
  double nloc[3], dist, ploc[3];
  if (!inside) {
    double dv[3] = {loc[0]-clamped[0], loc[1]-clamped[1], loc[2]-clamped[2]};
    double len = norm3(dv);
    dist = len - r1;
    if (dist > margin) return 0;
    for (int k = 0; k < 3; k++) { nloc[k] = -dv[k] / len; ploc[k] = clamped[k] - nloc[k] * 0.5 * dist; } /* normal sphere -> box */
  } else {
    int best = 0; double bd = 1e300;
    for (int k = 0; k < 3; k++) { double dd = s2[k] - fabs(loc[k]); if (dd < bd) { bd = dd; best = k; } }
    nloc[0] = nloc[1] = nloc[2] = 0; nloc[best] = loc[best] >= 0 ? -1 : 1;
    dist = -bd - r1;
    for (int k = 0; k < 3; k++) ploc[k] = loc[k];
    ploc[best] = (loc[best] >= 0 ? s2[best] : -s2[best]) - nloc[best] * 0.5 * dist;
  }
  out->dist = dist; rotvec(out->n, m2, nloc);
  double pw[3]; rotvec(pw, m2, ploc);
  for (int k = 0; k < 3; k++) out->pos[k] = pw[k] + c2[k];
  return 1;
}

/* Box-box: 15-axis SAT + incident-face/reference-face manifold (<= 8 points).
 * NOT MuJoCo's mjc_BoxBox point selection (SURVEY.md App. B.5 flags that as not
 * reproducible without the library); this definition is shared with the HIP path.
 * m1/m2 are row-major rotation matrices (columns = box axes). Normal points 1 -> 2. */
int orc_box_box(const double* p1, const double* m1, const double* s1, const double* p2, const double* m2,
                const double* s2, double margin, double* dist, double* pos, double* normal) {
  double t[3] = {p2[0]-p1[0], p2[1]-p1[1], p2[2]-p1[2]};
  double C[3][3], AC[3][3], ta[3], tb[3];
  for (int i = 0; i < 3; i++) {
    double ai[3] = {m1[i], m1[3+i], m1[6+i]};
    ta[i] = dot3(t, ai);
    for (int j = 0; j < 3; j++) { double bj[3] = {m2[j], m2[3+j], m2[6+j]}; C[i][j] = dot3(ai, bj); AC[i][j] = fabs(C[i][j]) + 1e-9; }
  }
  for (int j = 0; j < 3; j++) { double bj[3] = {m2[j], m2[3+j], m2[6+j]}; tb[j] = dot3(t, bj); }
  double sface = -1e300; int cface = -1;
  for (int i = 0; i < 3; i++) {
    double s = fabs(ta[i]) - (s1[i] + s2[0]*AC[i][0] + s2[1]*AC[i][1] + s2[2]*AC[i][2]);
    if (s > margin) return 0;
    if (s > sface) { sface = s; cface = i; }
  }
  for (int j = 0; j < 3; j++) {
    double s = fabs(tb[j]) - (s2[j] + s1[0]*AC[0][j] + s1[1]*AC[1][j] + s1[2]*AC[2][j]);
    if (s > margin) return 0;
    if (s > sface) { sface = s; cface = 3 + j; }
  }
  double sedge = -1e300; int cedge = -1;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double l2 = 1 - C[i][j]*C[i][j];
    if (l2 < 1e-6) continue;
    double l = sqrt(l2);
    int i1 = (i+1)%3, i2 = (i+2)%3, j1 = (j+1)%3, j2 = (j+2)%3;
    double tl = ta[i2]*C[i1][j] - ta[i1]*C[i2][j];
    double ra = s1[i1]*AC[i2][j] + s1[i2]*AC[i1][j], rb = s2[j1]*AC[i][j2] + s2[j2]*AC[i][j1];
    double s = (fabs(tl) - (ra + rb)) / l;
    if (s > margin) return 0;
    if (s > sedge) { sedge = s; cedge = 3*i + j; }
  }
  if (cedge >= 0 && sedge > sface + 0.05 * fabs(sface) + 1e-9) {
    /* edge-edge: one point midway between the closest points of the two supporting edges */
    int i = cedge / 3, j = cedge % 3;
    double ai[3] = {m1[i], m1[3+i], m1[6+i]}, bj[3] = {m2[j], m2[3+j], m2[6+j]}, L[3];
    cross3(L, ai, bj); normalize3(L);
    if (dot3(L, t) < 0) { L[0] = -L[0]; L[1] = -L[1]; L[2] = -L[2]; }
    double P1[3] = {p1[0], p1[1], p1[2]}, P2[3] = {p2[0], p2[1], p2[2]};
    for (int k = 0; k < 3; k++) if (k != i) {
      double ak[3] = {m1[k], m1[3+k], m1[6+k]}; double sg = dot3(L, ak) >= 0 ? 1 : -1;
      for (int q = 0; q < 3; q++) P1[q] += sg * s1[k] * ak[q];
    }
    for (int k = 0; k < 3; k++) if (k != j) {
      double bk[3] = {m2[k], m2[3+k], m2[6+k]}; double sg = dot3(L, bk) >= 0 ? -1 : 1;
      for (int q = 0; q < 3; q++) P2[q] += sg * s2[k] * bk[q];
    }
    double dd[3] = {P2[0]-P1[0], P2[1]-P1[1], P2[2]-P1[2]};
    double c = C[i][j], da = dot3(dd, ai), db = dot3(dd, bj), den = 1 - c*c;
    double sa = (da - c*db) / den, sb = (c*da - db) / den;
    if (sa > s1[i]) sa = s1[i]; if (sa < -s1[i]) sa = -s1[i];
    if (sb > s2[j]) sb = s2[j]; if (sb < -s2[j]) sb = -s2[j];
    for (int q = 0; q < 3; q++) pos[q] = 0.5 * ((P1[q] + sa*ai[q]) + (P2[q] + sb*bj[q]));
    dist[0] = sedge; copyv(normal, L, 3);
    return 1;
  }
  /* face case: reference box R (its face axis won), incident box I */
  const double *pR, *mR, *sR, *pI, *mI, *sI; int k, flip;
  if (cface < 3) { pR = p1; mR = m1; sR = s1; pI = p2; mI = m2; sI = s2; k = cface; flip = 0; }
  else { pR = p2; mR = m2; sR = s2; pI = p1; mI = m1; sI = s1; k = cface - 3; flip = 1; }
  double dpw[3] = {pI[0]-pR[0], pI[1]-pR[1], pI[2]-pR[2]}, p[3], M[3][3];
  rotvecT(p, mR, dpw);
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { /* M = mR^T mI: column c = I's axis c in R's frame */
    double v = 0; for (int q = 0; q < 3; q++) v += mR[3*q+r] * mI[3*q+c]; M[r][c] = v;
  }
  double sg = p[k] >= 0 ? 1.0 : -1.0;
  int u = (k+1)%3, v = (k+2)%3;
  int js = 0; double best = -1;
  for (int j = 0; j < 3; j++) if (fabs(M[k][j]) > best) { best = fabs(M[k][j]); js = j; }
  double tau = (sg * M[k][js] > 0) ? -1.0 : 1.0;
  int j1 = (js+1)%3, j2 = (js+2)%3;
  double cI[3], e1[3], e2[3], mn[3];
  for (int q = 0; q < 3; q++) { mn[q] = M[q][js]; cI[q] = p[q] + tau * sI[js] * mn[q]; e1[q] = M[q][j1]; e2[q] = M[q][j2]; }
  double vq[4][3];
  for (int q = 0; q < 4; q++) {
    double a = (q == 0 || q == 3) ? 1.0 : -1.0, b = (q < 2) ? 1.0 : -1.0;
    for (int r = 0; r < 3; r++) vq[q][r] = cI[r] + a * sI[j1] * e1[r] + b * sI[j2] * e2[r];
  }
  int cnt = 0; double cand[3];
  double nw[3] = {sg * mR[k], sg * mR[3+k], sg * mR[6+k]}; /* from R towards I */
#define EMIT_CAND() do { \
    double delta = sg * cand[k] - sR[k]; \
    if (delta < margin && cnt < 8) { \
      double mid[3] = {cand[0], cand[1], cand[2]}; mid[k] -= sg * 0.5 * delta; \
      double w[3]; rotvec(w, mR, mid); \
      for (int q_ = 0; q_ < 3; q_++) pos[3*cnt+q_] = w[q_] + pR[q_]; \
      dist[cnt] = delta; cnt++; } } while (0)
  /* (a) incident-face vertices inside the reference rectangle */
  for (int q = 0; q < 4; q++) if (fabs(vq[q][u]) <= sR[u] && fabs(vq[q][v]) <= sR[v]) { copyv(cand, vq[q], 3); EMIT_CAND(); }
  /* (b) reference-face corners projected along the reference axis onto the incident face */
  for (int q = 0; q < 4; q++) {
    double ru = ((q == 0 || q == 3) ? 1.0 : -1.0) * sR[u], rv = ((q < 2) ? 1.0 : -1.0) * sR[v];
    double xk = cI[k] - ((ru - cI[u]) * mn[u] + (rv - cI[v]) * mn[v]) / mn[k];
    double x[3]; x[u] = ru; x[v] = rv; x[k] = xk;
    double dv[3] = {x[0]-cI[0], x[1]-cI[1], x[2]-cI[2]};
    if (fabs(dot3(dv, e1)) <= sI[j1] && fabs(dot3(dv, e2)) <= sI[j2]) { copyv(cand, x, 3); EMIT_CAND(); }
  }
  /* (c) incident edges crossing the sides of the reference rectangle (projection along axis k) */
  for (int e = 0; e < 4; e++) {
    const double* P = vq[e]; const double* Q = vq[(e+1)%4];
    for (int side = 0; side < 4; side++) {
      int ax = (side < 2) ? u : v, ox = (side < 2) ? v : u;
      double hh = ((side & 1) ? -1.0 : 1.0) * sR[ax];
      double fp = P[ax] - hh, fq = Q[ax] - hh;
      if (fp * fq >= 0) continue;
      double s = fp / (fp - fq);
      for (int r = 0; r < 3; r++) cand[r] = P[r] + s * (Q[r] - P[r]);
      if (fabs(cand[ox]) < sR[ox]) EMIT_CAND();
    }
  }
#undef EMIT_CAND
  for (int q = 0; q < 3; q++) normal[q] = flip ? -nw[q] : nw[q];
  return cnt;
}

/* ---- generic convex - convex narrow phase: Minkowski portal refinement (MPR, G. Snethen, "XenoCollide", Game
 * Programming Gems 7) over support mappings, one contact per pair.  [UPSTREAM mjc_Convex: MuJoCo 2.3.7 sends every
 * pair without an analytic routine (cylinder-x, capsule-box, ellipsoid-x, mesh-x) through libccd's MPR with
 * mpr_tolerance 1e-6 and mpr_iterations 50 and keeps a single contact; margin is handled by inflating both geoms by
 * margin/2.]  The restatement below is this project's own; the device routine (csrc/dev_convex.h) follows it step
 * for step in fp32. */
#define MPR_TOL 1e-6
#define MPR_ITER 50
#define MPR_EPS 1e-10
typedef struct { int type; const double *pos, *mat, *size; double pad; const double* vert; int nvert; } cvx_geom;
typedef struct { double v[3], a[3], b[3]; } mpr_pt;   /* v = a - b: point of the Minkowski difference with its witnesses */

/* farthest point of the geom along the unit world direction `dir` */
static void cvx_support(const cvx_geom* g, const double* dir, double* out) {
  double dl[3], pl[3] = {0, 0, 0};
  rotvecT(dl, g->mat, dir);
  const double* s = g->size;
  switch (g->type) {
    case MJH_GEOM_SPHERE: for (int k = 0; k < 3; k++) pl[k] = s[0] * dl[k]; break;
    case MJH_GEOM_CAPSULE: for (int k = 0; k < 3; k++) pl[k] = s[0] * dl[k]; pl[2] += dl[2] >= 0 ? s[1] : -s[1]; break;
    case MJH_GEOM_CYLINDER: {
      double r2 = dl[0]*dl[0] + dl[1]*dl[1];
      if (r2 > MPR_EPS) { double sc = s[0] / sqrt(r2); pl[0] = dl[0] * sc; pl[1] = dl[1] * sc; }
      pl[2] = dl[2] >= 0 ? s[1] : -s[1];
    } break;
    case MJH_GEOM_BOX: for (int k = 0; k < 3; k++) pl[k] = dl[k] >= 0 ? s[k] : -s[k]; break;
    case MJH_GEOM_ELLIPSOID: {
      double w[3] = {s[0]*s[0]*dl[0], s[1]*s[1]*dl[1], s[2]*s[2]*dl[2]};
      double den = sqrt(w[0]*dl[0] + w[1]*dl[1] + w[2]*dl[2]);
      if (den > MINVAL) for (int k = 0; k < 3; k++) pl[k] = w[k] / den;
    } break;
    case MJH_GEOM_MESH: {
      double best = -1e300; int bi = 0;
      for (int i = 0; i < g->nvert; i++) { double dp = dot3(g->vert + 3*i, dl); if (dp > best) { best = dp; bi = i; } }
      if (g->nvert) copyv(pl, g->vert + 3*bi, 3);
    } break;
    default: break;
  }
  rotvec(out, g->mat, pl);
  for (int k = 0; k < 3; k++) out[k] += g->pos[k] + g->pad * dir[k];
}
/* test hook: support-mapping evaluations (what the narrow phase costs), total and the maximum over the pairs since the last reset */
static long mpr_calls_total = 0, mpr_calls_pair = 0, mpr_calls_max = 0, mpr_pairs = 0;
long orc_debug_mpr_calls(int what, int reset) {
  long v = what == 0 ? mpr_calls_total : what == 1 ? mpr_calls_max : mpr_pairs;
  if (reset) { mpr_calls_total = 0; mpr_calls_max = 0; mpr_pairs = 0; }
  return v;
}
static void mpr_support(const cvx_geom* g1, const cvx_geom* g2, const double* dir, mpr_pt* p) {
  mpr_calls_total++; mpr_calls_pair++;
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  cvx_support(g1, dir, p->a); cvx_support(g2, nd, p->b);
  for (int k = 0; k < 3; k++) p->v[k] = p->a[k] - p->b[k];
}
static void tri_normal(double* n, const mpr_pt* p1, const mpr_pt* p2, const mpr_pt* p3) {
  double e1[3], e2[3];
  for (int k = 0; k < 3; k++) { e1[k] = p2->v[k] - p1->v[k]; e2[k] = p3->v[k] - p1->v[k]; }
  cross3(n, e1, e2); normalize3(n);
}
/* the support point p4 is no further out along n than the portal by more than the tolerance */
static int mpr_converged(const mpr_pt* p1, const mpr_pt* p2, const mpr_pt* p3, const mpr_pt* p4, const double* n) {
  double d4 = dot3(p4->v, n), m = d4 - dot3(p1->v, n), m2 = d4 - dot3(p2->v, n), m3 = d4 - dot3(p3->v, n);
  if (m2 < m) m = m2;
  if (m3 < m) m = m3;
  return m <= MPR_TOL;
}
/* replace the portal vertex that keeps the ray (interior point -> origin) inside the portal */
static void mpr_expand(const mpr_pt* p0, mpr_pt* p1, mpr_pt* p2, mpr_pt* p3, const mpr_pt* p4) {
  double c[3];
  cross3(c, p4->v, p0->v);
  if (dot3(p1->v, c) > 0) { if (dot3(p2->v, c) > 0) *p1 = *p4; else *p3 = *p4; }
  else { if (dot3(p3->v, c) > 0) *p2 = *p4; else *p1 = *p4; }
}
/* closest point of triangle (a,b,c) to the origin (Voronoi-region walk) */
static void tri_closest_to_origin(const double* a, const double* b, const double* c, double* out) {
  double ab[3], ac[3], ap[3], bp[3], cp[3];
  for (int k = 0; k < 3; k++) { ab[k] = b[k]-a[k]; ac[k] = c[k]-a[k]; ap[k] = -a[k]; bp[k] = -b[k]; cp[k] = -c[k]; }
  double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  if (d1 <= 0 && d2 <= 0) { copyv(out, a, 3); return; }
  double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
  if (d3 >= 0 && d4 <= d3) { copyv(out, b, 3); return; }
  double vc = d1*d4 - d3*d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { double t = d1 / (d1 - d3); for (int k = 0; k < 3; k++) out[k] = a[k] + t*ab[k]; return; }
  double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
  if (d6 >= 0 && d5 <= d6) { copyv(out, c, 3); return; }
  double vb = d5*d2 - d1*d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { double t = d2 / (d2 - d6); for (int k = 0; k < 3; k++) out[k] = a[k] + t*ac[k]; return; }
  double va = d3*d6 - d5*d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { double t = (d4 - d3) / ((d4 - d3) + (d5 - d6)); for (int k = 0; k < 3; k++) out[k] = b[k] + t*(c[k]-b[k]); return; }
  double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den;
  for (int k = 0; k < 3; k++) out[k] = a[k] + v*ab[k] + w*ac[k];
}
/* 1 = the geoms overlap: depth, direction (geom1 -> geom2) and position of the single contact */
static int mpr_penetration(const cvx_geom* g1, const cvx_geom* g2, double* depth, double* dir, double* pos) {
  mpr_pt p0, p1, p2, p3, p4;
  double n[3], c[3];
  /* interior point of the Minkowski difference: the centre difference */
  for (int k = 0; k < 3; k++) { p0.a[k] = g1->pos[k]; p0.b[k] = g2->pos[k]; p0.v[k] = p0.a[k] - p0.b[k]; }
  if (dot3(p0.v, p0.v) < MPR_EPS) p0.v[0] += 1e-4;
  for (int k = 0; k < 3; k++) n[k] = -p0.v[k];
  normalize3(n);
  mpr_support(g1, g2, n, &p1);
  if (dot3(p1.v, n) <= 0) return 0;
  cross3(n, p0.v, p1.v);
  if (dot3(n, n) < 1e-12 * dot3(p0.v, p0.v) * dot3(p1.v, p1.v)) {   /* the origin lies on the ray through p0 and p1: penetration along that ray */
    *depth = norm3(p1.v);
    copyv(dir, p1.v, 3); normalize3(dir);
    for (int k = 0; k < 3; k++) pos[k] = 0.5 * (p1.a[k] + p1.b[k]);
    return 1;
  }
  normalize3(n);
  mpr_support(g1, g2, n, &p2);
  if (dot3(p2.v, n) <= 0) return 0;
  /* portal discovery: a triangle (p1,p2,p3) that the ray p0 -> origin passes through */
  { double e1[3], e2[3];
    for (int k = 0; k < 3; k++) { e1[k] = p1.v[k] - p0.v[k]; e2[k] = p2.v[k] - p0.v[k]; }
    cross3(n, e1, e2); normalize3(n);
    if (dot3(n, p0.v) > 0) { mpr_pt t = p1; p1 = p2; p2 = t; for (int k = 0; k < 3; k++) n[k] = -n[k]; } }
  for (int it = 0;; it++) {
    if (it > MPR_ITER) return 0;
    mpr_support(g1, g2, n, &p3);
    if (dot3(p3.v, n) <= 0) return 0;
    int again = 0;
    cross3(c, p1.v, p3.v);
    if (dot3(c, p0.v) < -MPR_EPS) { p2 = p3; again = 1; }
    else { cross3(c, p3.v, p2.v); if (dot3(c, p0.v) < -MPR_EPS) { p1 = p3; again = 1; } }
    if (!again) break;
    double e1[3], e2[3];
    for (int k = 0; k < 3; k++) { e1[k] = p1.v[k] - p0.v[k]; e2[k] = p2.v[k] - p0.v[k]; }
    cross3(n, e1, e2); normalize3(n);
  }
  /* portal refinement until the origin is on the inner side of the portal (hit) or provably outside (miss) */
  for (int it = 0;; it++) {
    tri_normal(n, &p1, &p2, &p3);
    if (dot3(n, p1.v) >= -MPR_EPS) break;
    mpr_support(g1, g2, n, &p4);
    if (dot3(p4.v, n) < -MPR_EPS || mpr_converged(&p1, &p2, &p3, &p4, n) || it > MPR_ITER) return 0;
    mpr_expand(&p0, &p1, &p2, &p3, &p4);
  }
  /* push the portal to the surface */
  for (int it = 0;; it++) {
    tri_normal(n, &p1, &p2, &p3);
    mpr_support(g1, g2, n, &p4);
    if (mpr_converged(&p1, &p2, &p3, &p4, n) || it > MPR_ITER) break;
    mpr_expand(&p0, &p1, &p2, &p3, &p4);
  }
  tri_closest_to_origin(p1.v, p2.v, p3.v, c);
  *depth = norm3(c);
  if (*depth < MPR_EPS) copyv(dir, n, 3); else for (int k = 0; k < 3; k++) dir[k] = c[k] / *depth;
  /* position: barycentric coordinates of the origin in the tetrahedron (p0,p1,p2,p3) applied to the witnesses */
  { double b[4], x[3], sum;
    cross3(x, p1.v, p2.v); b[0] = dot3(x, p3.v);
    cross3(x, p3.v, p2.v); b[1] = dot3(x, p0.v);
    cross3(x, p0.v, p1.v); b[2] = dot3(x, p3.v);
    cross3(x, p2.v, p1.v); b[3] = dot3(x, p0.v);
    sum = b[0] + b[1] + b[2] + b[3];
    if (sum <= 0) {
      b[0] = 0;
      cross3(x, p2.v, p3.v); b[1] = dot3(x, n);
      cross3(x, p3.v, p1.v); b[2] = dot3(x, n);
      cross3(x, p1.v, p2.v); b[3] = dot3(x, n);
      sum = b[1] + b[2] + b[3];
    }
    const mpr_pt* P[4] = {&p0, &p1, &p2, &p3};
    for (int k = 0; k < 3; k++) {
      double acc = 0;
      for (int q = 0; q < 4; q++) acc += b[q] * (P[q]->a[k] + P[q]->b[k]);
      pos[k] = 0.5 * acc / sum;
    } }
  return 1;
}
static int c_convex(const cvx_geom* g1, const cvx_geom* g2, double margin, rawcon* out) {
  cvx_geom a = *g1, b = *g2;
  a.pad = b.pad = 0.5 * margin;
  double depth, dir[3], pos[3];
  mpr_calls_pair = 0; mpr_pairs++;
  const int hit = mpr_penetration(&a, &b, &depth, dir, pos);
  if (mpr_calls_pair > mpr_calls_max) mpr_calls_max = mpr_calls_pair;
  if (!hit) return 0;
  out->dist = margin - depth; copyv(out->pos, pos, 3); copyv(out->n, dir, 3);
  return 1;
}
/* test hook: one convex pair */
int orc_convex_pair(int t1, const double* p1, const double* m1, const double* s1, const double* v1, int n1,
                    int t2, const double* p2, const double* m2, const double* s2, const double* v2, int n2,
                    double margin, double* dist, double* pos, double* normal) {
  cvx_geom a = {t1, p1, m1, s1, 0, v1, n1}, b = {t2, p2, m2, s2, 0, v2, n2};
  rawcon rc;
  int n = c_convex(&a, &b, margin, &rc);
  if (n) { *dist = rc.dist; copyv(pos, rc.pos, 3); copyv(normal, rc.n, 3); }
  return n;
}

/* pairs without an analytic routine go through the generic convex narrow phase (types ordered t1 <= t2) */
static int pair_is_convex(int t1, int t2) {
  if (t1 == MJH_GEOM_PLANE || t1 == MJH_GEOM_HFIELD || t2 == MJH_GEOM_HFIELD) return 0;
  if (t1 == MJH_GEOM_ELLIPSOID || t2 == MJH_GEOM_ELLIPSOID || t1 == MJH_GEOM_CYLINDER || t2 == MJH_GEOM_CYLINDER) return 1;
  if (t2 == MJH_GEOM_MESH) return 1;
  return t1 == MJH_GEOM_CAPSULE && t2 == MJH_GEOM_BOX;
}
static double mixd(double a, double b, double mix) { return mix * a + (1 - mix) * b; }

/* broad phase over the compiled pair list + narrow phase [UPSTREAM mj_collision],
 * part of mj_step1 (mj_main.cpp:83) */
void orc_collision(orc_data* d) {
  const mjh_model* m = d->m;
  const double* gsize = p_geom_size(d); const double* rbound = p_geom_rbound(d);
  d->ncon = 0;
  if (m->opt.disableflags & (MJH_DSBL_CONTACT | MJH_DSBL_CONSTRAINT)) return;
  for (int ip = 0; ip < m->npair; ip++) {
    int g1 = m->pair_geom1[ip], g2 = m->pair_geom2[ip];
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]), gap = fmax(m->geom_gap[g1], m->geom_gap[g2]);
    const double *p1 = d->geom_xpos + 3*g1, *p2 = d->geom_xpos + 3*g2, *m1 = d->geom_xmat + 9*g1, *m2 = d->geom_xmat + 9*g2;
    const double *s1 = gsize + 3*g1, *s2 = gsize + 3*g2;
    /* bounding-sphere cull */
    if (t1 == MJH_GEOM_PLANE) {
      double n[3] = {m1[2], m1[5], m1[8]}, tt[3] = {p2[0]-p1[0], p2[1]-p1[1], p2[2]-p1[2]};
      if (dot3(tt, n) > rbound[g2] + margin) continue;
    } else {
      double tt[3] = {p2[0]-p1[0], p2[1]-p1[1], p2[2]-p1[2]}, bound = rbound[g1] + rbound[g2] + margin;
      if (dot3(tt, tt) > bound * bound) continue;
    }
    rawcon rc[8]; int n = 0;
    double bd[8], bp[24], bn[3];
    if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_SPHERE) n = c_plane_sphere(p1, m1, p2, s2[0], margin, rc);
    else if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_CAPSULE) n = c_plane_capsule(p1, m1, p2, m2, s2, margin, rc);
    else if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_BOX) n = c_plane_box(p1, m1, p2, m2, s2, margin, rc);
    else if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_CYLINDER) n = c_plane_cylinder(p1, m1, p2, m2, s2, margin, rc);
    else if (t1 == MJH_GEOM_SPHERE && t2 == MJH_GEOM_SPHERE) n = c_sphere_sphere(p1, s1[0], p2, s2[0], margin, rc);
    else if (t1 == MJH_GEOM_SPHERE && t2 == MJH_GEOM_CAPSULE) n = c_sphere_capsule(p1, s1[0], p2, m2, s2, margin, rc);
    else if (t1 == MJH_GEOM_CAPSULE && t2 == MJH_GEOM_CAPSULE) n = c_capsule_capsule(p1, m1, s1, p2, m2, s2, margin, rc);
    else if (t1 == MJH_GEOM_SPHERE && t2 == MJH_GEOM_BOX) n = c_sphere_box(p1, s1[0], p2, m2, s2, margin, rc);
    else if (t1 == MJH_GEOM_BOX && t2 == MJH_GEOM_BOX) {
      n = orc_box_box(p1, m1, s1, p2, m2, s2, margin, bd, bp, bn);
      for (int q = 0; q < n; q++) { rc[q].dist = bd[q]; copyv(rc[q].pos, bp + 3*q, 3); copyv(rc[q].n, bn, 3); }
    }
    else if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_ELLIPSOID) n = c_plane_ellipsoid(p1, m1, p2, m2, s2, margin, rc);
    else if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_MESH) {
      int id = m->geom_dataid[g2];
      n = c_plane_mesh(p1, m1, p2, m2, m->mesh_vert + 3*m->mesh_vertadr[id], m->mesh_vertnum[id], margin, rc);
    }
    else if (pair_is_convex(t1, t2)) {
      cvx_geom a = {t1, p1, m1, s1, 0, 0, 0}, b = {t2, p2, m2, s2, 0, 0, 0};
      if (t1 == MJH_GEOM_MESH) { int id = m->geom_dataid[g1]; a.vert = m->mesh_vert + 3*m->mesh_vertadr[id]; a.nvert = m->mesh_vertnum[id]; }
      if (t2 == MJH_GEOM_MESH) { int id = m->geom_dataid[g2]; b.vert = m->mesh_vert + 3*m->mesh_vertadr[id]; b.nvert = m->mesh_vertnum[id]; }
      n = c_convex(&a, &b, margin, rc);
    }
    if (!n) continue;
    { int sb1 = m->geom_bodyid[g1], sb2 = m->geom_bodyid[g2];   /* inactive spawn/destroy slots do not collide */
      unsigned sbase = m->nbody > 32 ? (unsigned)(m->nbody - 32) : 0u, r1 = (unsigned)sb1 - sbase, r2 = (unsigned)sb2 - sbase;   /* bit i = body sbase + i */
      if ((r1 < 32u && ((d->slot_mask >> r1) & 1u)) || (r2 < 32u && ((d->slot_mask >> r2) & 1u))) continue; }
    /* contact parameters [UPSTREAM mj_contactParam]: max condim, max friction, solmix-weighted solref/solimp */
    int dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
    double fr[3], mix;
    for (int k = 0; k < 3; k++) fr[k] = fmax(m->geom_friction[3*g1+k], m->geom_friction[3*g2+k]);
    {
      double a = m->geom_solmix[g1], b = m->geom_solmix[g2];
      if (a >= MINVAL && b >= MINVAL) mix = a / (a + b); else if (a < MINVAL && b < MINVAL) mix = 0.5; else mix = a < MINVAL ? 0 : 1;
    }
    for (int q = 0; q < n; q++) {
      if (d->ncon >= m->maxcon) { d->warn |= 1; break; }
      orc_contact* c = d->contact + d->ncon;
      c->dist = rc[q].dist; copyv(c->pos, rc[q].pos, 3);
      copyv(c->frame, rc[q].n, 3); c->frame[3] = c->frame[4] = c->frame[5] = 0;
      make_frame(c->frame);
      c->includemargin = margin - gap;
      c->dim = dim; c->geom1 = g1; c->geom2 = g2;
      c->friction[0] = c->friction[1] = fr[0]; c->friction[2] = fr[1]; c->friction[3] = c->friction[4] = fr[2];
      for (int k = 0; k < 2; k++) c->solref[k] = mixd(m->geom_solref[2*g1+k], m->geom_solref[2*g2+k], mix);
      for (int k = 0; k < 5; k++) c->solimp[k] = mixd(m->geom_solimp[5*g1+k], m->geom_solimp[5*g2+k], mix);
      c->exclude = (c->dist >= c->includemargin);
      c->efc_address = -1; c->mu = 0;
      d->ncon++;
    }
  }
}

/* ------------------------------------------------------------------ constraints */
/* translational / rotational Jacobian of a world point attached to body b (dense 3 x nv) [UPSTREAM mj_jac] */
static void jac_point(const orc_data* d, double* jp, double* jr, const double* point, int body) {
  const mjh_model* m = d->m; int nv = m->nv;
  if (jp) zero(jp, 3*nv);
  if (jr) zero(jr, 3*nv);
  if (body <= 0) return;
  double off[3]; const double* com = d->subtree_com + 3*m->body_rootid[body];
  for (int k = 0; k < 3; k++) off[k] = point[k] - com[k];
  while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
  if (body <= 0) return;
  for (int i = m->body_dofadr[body] + m->body_dofnum[body] - 1; i >= 0; i = m->dof_parentid[i]) {
    const double* c = d->cdof + 6*i;
    if (jr) { jr[i] = c[0]; jr[nv+i] = c[1]; jr[2*nv+i] = c[2]; }
    if (jp) {
      double t[3]; cross3(t, c, off);
      jp[i] = c[3] + t[0]; jp[nv+i] = c[4] + t[1]; jp[2*nv+i] = c[5] + t[2];
    }
  }
}

static int add_row(orc_data* d, int type, int id, double pos, double margin, double floss) {
  if (d->nefc >= d->m->maxefc) { d->warn |= 2; return -1; }
  int i = d->nefc++;
  d->efc_type[i] = type; d->efc_id[i] = id; d->efc_pos[i] = pos; d->efc_margin[i] = margin; d->efc_frictionloss[i] = floss;
  zero(d->efc_J + (size_t)i * d->m->nv, d->m->nv);
  return i;
}

static void get_impedance(const double* solimp_in, double pos, double margin, double* imp) {
  double s[5]; copyv(s, solimp_in, 5);
  s[0] = fmin(MAXIMP, fmax(MINIMP, s[0])); s[1] = fmin(MAXIMP, fmax(MINIMP, s[1]));
  s[2] = fmax(0, s[2]); s[3] = fmin(MAXIMP, fmax(MINIMP, s[3])); s[4] = fmax(1, s[4]);
  if (s[0] == s[1] || s[2] <= MINVAL) { *imp = 0.5 * (s[0] + s[1]); return; }
  double x = fabs((pos - margin) / s[2]);
  if (x >= 1) { *imp = s[1]; return; }
  if (x <= 0) { *imp = s[0]; return; }
  double y;
  if (s[4] == 1) y = x;
  else if (x <= s[3]) y = pow(x, s[4]) / pow(s[3], s[4] - 1);
  else y = 1 - pow(1 - x, s[4]) / pow(1 - s[3], s[4] - 1);
  *imp = s[0] + y * (s[1] - s[0]);
}

/* rows in the order equality, friction loss, limit, contact; then diagApprox, R/D, KBIP
 * [UPSTREAM mj_makeConstraint + mj_diagApprox + mj_makeImpedance] */
void orc_make_constraint(orc_data* d) {
  const mjh_model* m = d->m; int nv = m->nv;
  const double* binv = p_body_invweight0(d); const double* dinv = p_dof_invweight0(d);
  d->nefc = 0;
  if (m->opt.disableflags & MJH_DSBL_CONSTRAINT) return;
  /* equality (joint coupling, as produced by mujoco_compile.cpp:235-242) */
  if (!(m->opt.disableflags & MJH_DSBL_EQUALITY))
    for (int e = 0; e < m->neq; e++) {
      if (!m->eq_active[e]) continue;
      if (m->eq_type[e] == MJH_EQ_CONNECT || m->eq_type[e] == MJH_EQ_WELD) {
        /* [UPSTREAM mj_makeConstraint, mjEQ_CONNECT / mjEQ_WELD — restated, conventions in include/mjhip.h]: 3 rows
         * p1 - p2 of the two anchors (world axes), weld: 3 more rows torquescale * vec(q1^-1 q2 relpose), whose time
         * derivative is 1/2 torquescale vec(q1^-1 (w2 - w1) q2 relpose) */
        const int weld = m->eq_type[e] == MJH_EQ_WELD, b1 = m->eq_obj1id[e], b2 = m->eq_obj2id[e];
        const double* dat = m->eq_data + 11*e;
        const double *a1 = weld ? dat + 3 : dat, *a2 = weld ? dat : dat + 3;
        double p1[3], p2[3], t[3];
        rotvec(t, d->xmat + 9*b1, a1); for (int k = 0; k < 3; k++) p1[k] = d->xpos[3*b1+k] + t[k];
        rotvec(t, d->xmat + 9*b2, a2); for (int k = 0; k < 3; k++) p2[k] = d->xpos[3*b2+k] + t[k];
        double* jb = d->scr_jac;
        double *jp1 = jb, *jp2 = jb + 3*nv, *jr1 = jb + 6*nv, *jr2 = jb + 9*nv;
        jac_point(d, jp1, jr1, p1, b1); jac_point(d, jp2, jr2, p2, b2);
        const double tran = binv[2*b1] + binv[2*b2], rot = binv[2*b1+1] + binv[2*b2+1];
        int fail = 0;
        for (int k = 0; k < 3 && !fail; k++) {
          int r = add_row(d, MJH_CNSTR_EQUALITY, e, p1[k] - p2[k], 0, 0);
          if (r < 0) { fail = 1; break; }
          for (int q = 0; q < nv; q++) d->efc_J[(size_t)r*nv + q] = jp1[k*nv+q] - jp2[k*nv+q];
          d->efc_diagApprox[r] = tran;
        }
        if (weld && !fail) {
          const double ts = dat[10];
          double q1i[4] = {d->xquat[4*b1], -d->xquat[4*b1+1], -d->xquat[4*b1+2], -d->xquat[4*b1+3]}, q2r[4], qe[4];
          mulquat(q2r, d->xquat + 4*b2, dat + 6); mulquat(qe, q1i, q2r);
          for (int k = 0; k < 3 && !fail; k++) {
            int r = add_row(d, MJH_CNSTR_EQUALITY, e, ts * qe[1+k], 0, 0);
            if (r < 0) { fail = 1; break; }
            for (int q = 0; q < nv; q++) {
              double w[4] = {0, jr2[q] - jr1[q], jr2[nv+q] - jr1[nv+q], jr2[2*nv+q] - jr1[2*nv+q]}, u[4], v[4];
              if (w[1] == 0 && w[2] == 0 && w[3] == 0) continue;
              mulquat(u, q1i, w); mulquat(v, u, q2r);
              d->efc_J[(size_t)r*nv + q] = 0.5 * ts * v[1+k];
            }
            d->efc_diagApprox[r] = rot;
          }
        }
        if (fail) break;
        continue;
      }
      if (m->eq_type[e] != MJH_EQ_JOINT) continue;
      int j1 = m->eq_obj1id[e], j2 = m->eq_obj2id[e];
      const double* dat = m->eq_data + 11*e;
      double pos1 = d->qpos[m->jnt_qposadr[j1]] - m->qpos0[m->jnt_qposadr[j1]], cpos, deriv = 0;
      if (j2 >= 0) {
        double p2 = d->qpos[m->jnt_qposadr[j2]] - m->qpos0[m->jnt_qposadr[j2]];
        cpos = pos1 - (dat[0] + dat[1]*p2 + dat[2]*p2*p2 + dat[3]*p2*p2*p2 + dat[4]*p2*p2*p2*p2);
        deriv = dat[1] + 2*dat[2]*p2 + 3*dat[3]*p2*p2 + 4*dat[4]*p2*p2*p2;
      } else cpos = pos1 - dat[0];
      int r = add_row(d, MJH_CNSTR_EQUALITY, e, cpos, 0, 0);
      if (r < 0) break;
      d->efc_J[(size_t)r*nv + m->jnt_dofadr[j1]] = 1;
      d->efc_diagApprox[r] = dinv[m->jnt_dofadr[j1]];
      if (j2 >= 0) { d->efc_J[(size_t)r*nv + m->jnt_dofadr[j2]] = -deriv; d->efc_diagApprox[r] += dinv[m->jnt_dofadr[j2]]; }
    }
  /* dof friction loss */
  if (!(m->opt.disableflags & MJH_DSBL_FRICTIONLOSS))
    for (int i = 0; i < nv; i++) if (m->dof_frictionloss[i] > 0) {
      int r = add_row(d, MJH_CNSTR_FRICTION_DOF, i, 0, 0, m->dof_frictionloss[i]);
      if (r < 0) break;
      d->efc_J[(size_t)r*nv + i] = 1; d->efc_diagApprox[r] = dinv[i];
    }
  /* joint limits (hinge / slide) */
  if (!(m->opt.disableflags & MJH_DSBL_LIMIT))
    for (int j = 0; j < m->njnt; j++) {
      if (!m->jnt_limited[j]) continue;
      if (m->jnt_type[j] != MJH_JNT_HINGE && m->jnt_type[j] != MJH_JNT_SLIDE) continue;
      double value = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
      for (int side = -1; side <= 1; side += 2) {
        double dist = side * (m->jnt_range[2*j + (side + 1)/2] - value);
        if (dist < margin) {
          int r = add_row(d, MJH_CNSTR_LIMIT_JOINT, j, dist, margin, 0);
          if (r < 0) break;
          d->efc_J[(size_t)r*nv + m->jnt_dofadr[j]] = -side;
          d->efc_diagApprox[r] = dinv[m->jnt_dofadr[j]];
        }
      }
    }
  /* contacts: pyramidal friction cones */
  double* jbuf = d->scr_jac;
  double *jp1 = jbuf, *jp2 = jbuf + 3*nv, *jr1 = jbuf + 6*nv, *jr2 = jbuf + 9*nv;
  for (int ic = 0; ic < d->ncon; ic++) {
    orc_contact* c = d->contact + ic;
    if (c->exclude) continue;
    int b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2], dim = c->dim;
    jac_point(d, jp1, jr1, c->pos, b1); jac_point(d, jp2, jr2, c->pos, b2);
    /* difference (body2 - body1), rotated into the contact frame: rows 0..2 translational, 3..5 rotational */
    double* Jc = d->scr_B; /* 6 x nv */
    for (int r = 0; r < 3; r++) for (int q = 0; q < nv; q++) {
      double vp = 0, vr = 0;
      for (int k = 0; k < 3; k++) { vp += c->frame[3*r+k] * (jp2[k*nv+q] - jp1[k*nv+q]); vr += c->frame[3*r+k] * (jr2[k*nv+q] - jr1[k*nv+q]); }
      Jc[(size_t)r*nv+q] = vp; Jc[(size_t)(3+r)*nv+q] = vr;
    }
    double tran = binv[2*b1] + binv[2*b2], rot = binv[2*b1+1] + binv[2*b2+1];
    c->mu = c->friction[0] / sqrt(m->opt.impratio);
    if (dim == 1) {
      int r = add_row(d, MJH_CNSTR_CONTACT_FRICTIONLESS, ic, c->dist, c->includemargin, 0);
      if (r < 0) break;
      c->efc_address = r;
      copyv(d->efc_J + (size_t)r*nv, Jc, nv);
      d->efc_diagApprox[r] = tran;
    } else {
      if (d->nefc + 2*(dim-1) > m->maxefc) { d->warn |= 2; break; }
      c->efc_address = d->nefc;
      for (int k = 1; k < dim; k++) {
        double mu = c->friction[k-1];
        for (int sgn = 1; sgn >= -1; sgn -= 2) {
          int r = add_row(d, MJH_CNSTR_CONTACT_PYRAMIDAL, ic, c->dist, c->includemargin, 0);
          double* J = d->efc_J + (size_t)r*nv;
          for (int q = 0; q < nv; q++) J[q] = Jc[q] + sgn * mu * Jc[(size_t)k*nv+q];
          d->efc_diagApprox[r] = tran + mu*mu * (k < 3 ? tran : rot);
        }
      }
    }
  }
  /* impedance, regulariser, reference parameters */
  int nefc = d->nefc;
  for (int i = 0; i < nefc; i++) {
    const double *solref, *solimp; int id = d->efc_id[i];
    switch (d->efc_type[i]) {
      case MJH_CNSTR_EQUALITY: solref = m->eq_solref + 2*id; solimp = m->eq_solimp + 5*id; break;
      case MJH_CNSTR_FRICTION_DOF: solref = m->dof_solref + 2*id; solimp = m->dof_solimp + 5*id; break;
      case MJH_CNSTR_LIMIT_JOINT: solref = m->jnt_solref + 2*id; solimp = m->jnt_solimp + 5*id; break;
      default: solref = d->contact[id].solref; solimp = d->contact[id].solimp; break;
    }
    double imp; get_impedance(solimp, d->efc_pos[i], d->efc_margin[i], &imp);
    d->efc_R[i] = fmax(MINVAL, (1 - imp) * d->efc_diagApprox[i] / imp);
    double sr0 = solref[0], sr1 = solref[1], dmax = fmin(MAXIMP, fmax(MINIMP, solimp[1])), K, B;
    if (sr0 > 0) {
      if (!(m->opt.disableflags & MJH_DSBL_REFSAFE)) sr0 = fmax(sr0, 2 * m->opt.timestep);
      K = 1 / fmax(MINVAL, dmax*dmax * sr0*sr0 * sr1*sr1); B = 2 / fmax(MINVAL, dmax * sr0);
    } else { K = -sr0 / fmax(MINVAL, dmax*dmax); B = -sr1 / fmax(MINVAL, dmax); }
    if (d->efc_type[i] == MJH_CNSTR_FRICTION_DOF) K = 0;
    d->efc_KBIP[4*i] = K; d->efc_KBIP[4*i+1] = B; d->efc_KBIP[4*i+2] = imp; d->efc_KBIP[4*i+3] = 0;
  }
  /* pyramidal rows share one regulariser: Rpy = 2 mu^2 R(first row) */
  for (int ic = 0; ic < d->ncon; ic++) {
    orc_contact* c = d->contact + ic;
    if (c->exclude || c->efc_address < 0 || c->dim == 1) continue;
    int a = c->efc_address; double Rpy = fmax(MINVAL, 2 * c->mu * c->mu * d->efc_R[a]);
    for (int r = 0; r < 2*(c->dim - 1); r++) d->efc_R[a + r] = Rpy;
  }
  for (int i = 0; i < nefc; i++) d->efc_D[i] = 1 / d->efc_R[i];
}

/* AR = J M^-1 J^T + diag(R) for the dual solver [UPSTREAM mj_projectConstraint] */
void orc_project_constraint(orc_data* d) {
  const mjh_model* m = d->m; int nv = m->nv, nefc = d->nefc;
  for (int i = 0; i < nefc; i++) {
    double* B = d->scr_B + (size_t)i*nv;
    copyv(B, d->efc_J + (size_t)i*nv, nv);
    orc_solve_m(d, B);
  }
  /* AR_ij = J_i . B_j over the NONZEROS of J_i only (a contact row of a 64-box pile has 12 of 384): the skipped terms are
   * exact zeros, so every sum is the same sequence of additions — bit-identical to the dense dot, 30x fewer of them */
  int* nz = d->scr_nz; int* nzadr = d->scr_int;        /* (scr_int is free here: the order builder runs in orc_fwd_constraint) */
  int cnt = 0;
  for (int i = 0; i < nefc; i++) {
    nzadr[i] = cnt;
    const double* Ji = d->efc_J + (size_t)i*nv;
    for (int k = 0; k < nv; k++) if (Ji[k] != 0) nz[cnt++] = k;
  }
  nzadr[nefc] = cnt;
  for (int i = 0; i < nefc; i++) {
    const double* Ji = d->efc_J + (size_t)i*nv; const int* zi = nz + nzadr[i]; const int ni = nzadr[i+1] - nzadr[i];
    for (int j = 0; j <= i; j++) {
      const double* Bj = d->scr_B + (size_t)j*nv;
      double v = 0;
      for (int k = 0; k < ni; k++) v += Ji[zi[k]] * Bj[zi[k]];
      d->efc_AR[(size_t)i*nefc + j] = v; d->efc_AR[(size_t)j*nefc + i] = v;
    }
  }
  for (int i = 0; i < nefc; i++) d->efc_AR[(size_t)i*nefc + i] += d->efc_R[i];
}

/* ------------------------------------------------------------------ velocity stage */
/* [UPSTREAM mj_comVel] */
void orc_com_vel(orc_data* d) {
  const mjh_model* m = d->m;
  zero(d->cvel, 6);
  for (int i = 1; i < m->nbody; i++) {
    double cvel[6]; copyv(cvel, d->cvel + 6*m->body_parentid[i], 6);
    int bda = m->body_dofadr[i];
    for (int j = 0; j < m->body_jntnum[i]; j++) {
      int jid = m->body_jntadr[i] + j;
      switch (m->jnt_type[jid]) {
        case MJH_JNT_FREE:
          zero(d->cdof_dot + 6*bda, 18);
          for (int k = 0; k < 3; k++) for (int q = 0; q < 6; q++) cvel[q] += d->cdof[6*(bda+k)+q] * d->qvel[bda+k];
          bda += 3; /* fallthrough */
        case MJH_JNT_BALL:
          for (int k = 0; k < 3; k++) cross_motion(d->cdof_dot + 6*(bda+k), cvel, d->cdof + 6*(bda+k));
          for (int k = 0; k < 3; k++) for (int q = 0; q < 6; q++) cvel[q] += d->cdof[6*(bda+k)+q] * d->qvel[bda+k];
          bda += 3;
          break;
        default:
          cross_motion(d->cdof_dot + 6*bda, cvel, d->cdof + 6*bda);
          for (int q = 0; q < 6; q++) cvel[q] += d->cdof[6*bda+q] * d->qvel[bda];
          bda++;
      }
    }
    copyv(d->cvel + 6*i, cvel, 6);
  }
}

/* spring / damper / gravity compensation [UPSTREAM mj_passive]; gravcomp is what the wrapper
 * switches on for every robot body when ~disable_gravity is set (mj_sim.cpp:301-310) */
void orc_passive(orc_data* d) {
  const mjh_model* m = d->m; int nv = m->nv;
  const double* mass = p_body_mass(d);
  zero(d->qfrc_passive, nv);
  if (m->opt.disableflags & MJH_DSBL_PASSIVE) return;
  for (int j = 0; j < m->njnt; j++) {
    double k = m->jnt_stiffness[j];
    if (k == 0) continue;
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == MJH_JNT_HINGE || m->jnt_type[j] == MJH_JNT_SLIDE)
      d->qfrc_passive[da] -= k * (d->qpos[qa] - m->qpos_spring[qa]);
    /* ball/free springs: not used by any reference model */
  }
  for (int i = 0; i < nv; i++) d->qfrc_passive[i] -= m->dof_damping[i] * d->qvel[i];
  if (!(m->opt.disableflags & MJH_DSBL_GRAVITY)) {
    double* jp = d->scr_jac;
    for (int i = 1; i < m->nbody; i++) {
      if (m->body_gravcomp[i] == 0) continue;
      double f[3]; for (int k = 0; k < 3; k++) f[k] = -m->opt.gravity[k] * mass[i] * m->body_gravcomp[i];
      jac_point(d, jp, NULL, d->xipos + 3*i, i);
      for (int q = 0; q < nv; q++) d->qfrc_passive[q] += jp[q]*f[0] + jp[nv+q]*f[1] + jp[2*nv+q]*f[2];
    }
  }
}

/* efc_vel = J qvel; aref = -B vel - K imp (pos - margin) [UPSTREAM mj_referenceConstraint] */
void orc_reference_constraint(orc_data* d) {
  int nv = d->m->nv;
  for (int i = 0; i < d->nefc; i++) {
    d->efc_vel[i] = dotn(d->efc_J + (size_t)i*nv, d->qvel, nv);
    d->efc_aref[i] = -d->efc_KBIP[4*i+1] * d->efc_vel[i] - d->efc_KBIP[4*i] * d->efc_KBIP[4*i+2] * (d->efc_pos[i] - d->efc_margin[i]);
  }
}

/* recursive Newton-Euler [UPSTREAM mj_rne]; flg_acc=0 gives qfrc_bias */
void orc_rne(orc_data* d, int flg_acc, double* result) {
  const mjh_model* m = d->m; int nb = m->nbody;
  double *cacc = d->scr_body6[1], *cfrc = d->scr_body6[2];
  zero(cacc, 6); zero(cfrc, 6);
  if (!(m->opt.disableflags & MJH_DSBL_GRAVITY)) for (int k = 0; k < 3; k++) cacc[3+k] = -m->opt.gravity[k];
  for (int i = 1; i < nb; i++) {
    int bda = m->body_dofadr[i], p = m->body_parentid[i];
    double tmp[6], tmp1[6];
    copyv(cacc + 6*i, cacc + 6*p, 6);
    for (int k = 0; k < m->body_dofnum[i]; k++) for (int q = 0; q < 6; q++) {
      cacc[6*i+q] += d->cdof_dot[6*(bda+k)+q] * d->qvel[bda+k];
      if (flg_acc) cacc[6*i+q] += d->cdof[6*(bda+k)+q] * d->qacc[bda+k];
    }
    mul_inert_vec(cfrc + 6*i, d->cinert + 10*i, cacc + 6*i);
    mul_inert_vec(tmp, d->cinert + 10*i, d->cvel + 6*i);
    cross_force(tmp1, d->cvel + 6*i, tmp);
    for (int q = 0; q < 6; q++) cfrc[6*i+q] += tmp1[q];
  }
  for (int i = nb - 1; i > 0; i--) { int p = m->body_parentid[i]; if (p > 0) for (int q = 0; q < 6; q++) cfrc[6*p+q] += cfrc[6*i+q]; }
  for (int i = 0; i < m->nv; i++) result[i] = dotn(d->cdof + 6*i, cfrc + 6*m->dof_bodyid[i], 6);
}

/* potential + kinetic energy (the `energy` flag is on in world/empty.xml:3; shown by mj_visual.cpp:176) */
void orc_energy(orc_data* d) {
  const mjh_model* m = d->m; const double* mass = p_body_mass(d);
  double pot = 0;
  if (!(m->opt.disableflags & MJH_DSBL_GRAVITY))
    for (int i = 1; i < m->nbody; i++) pot -= mass[i] * dot3(m->opt.gravity, d->xipos + 3*i);
  for (int j = 0; j < m->njnt; j++) if (m->jnt_stiffness[j] != 0 && (m->jnt_type[j] == MJH_JNT_HINGE || m->jnt_type[j] == MJH_JNT_SLIDE)) {
    double dq = d->qpos[m->jnt_qposadr[j]] - m->qpos_spring[m->jnt_qposadr[j]];
    pot += 0.5 * m->jnt_stiffness[j] * dq * dq;
  }
  double* mv = d->scr_nv[5];
  orc_mul_m(d, mv, d->qvel);
  d->energy[0] = pot; d->energy[1] = 0.5 * dotn(mv, d->qvel, m->nv);
}

/* ------------------------------------------------------------------ pipeline */
static int bad(const double* x, int n) { for (int i = 0; i < n; i++) if (!(x[i] == x[i]) || x[i] > MAXVAL || x[i] < -MAXVAL) return 1; return 0; }
static void check_state(orc_data* d) { /* [UPSTREAM mj_checkPos/Vel/Acc]: auto-reset on NaN / huge values */
  if (bad(d->qpos, d->m->nq) || bad(d->qvel, d->m->nv) || bad(d->qacc, d->m->nv)) { double t = d->time; orc_reset(d); d->time = t; d->warn |= 4; }
}

void orc_fwd_position(orc_data* d) {
  orc_kinematics(d); orc_com_pos(d); orc_crb(d); orc_factor_m(d); orc_collision(d); orc_make_constraint(d); orc_project_constraint(d);
}
void orc_fwd_velocity(orc_data* d) { orc_com_vel(d); orc_passive(d); orc_reference_constraint(d); orc_rne(d, 0, d->qfrc_bias); }

/* qacc_smooth = M^-1 (passive - bias + applied) [UPSTREAM mj_fwdAcceleration; nu = 0 in every reference model] */
void orc_fwd_acceleration(orc_data* d) {
  int nv = d->m->nv;
  for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_applied[i];
  /* [UPSTREAM mj_xfrcAccumulate]: Cartesian force / torque on every body at its centre of mass, through the point Jacobian */
  {
    double* jb = NULL;
    for (int b = 1; b < d->m->nbody; b++) {
      const double* f = d->xfrc_applied + 6*b;
      if (f[0] == 0 && f[1] == 0 && f[2] == 0 && f[3] == 0 && f[4] == 0 && f[5] == 0) continue;
      jb = d->scr_jac;
      jac_point(d, jb, jb + 3*nv, d->xipos + 3*b, b);
      for (int i = 0; i < nv; i++) for (int k = 0; k < 3; k++) d->qfrc_smooth[i] += jb[k*nv+i] * f[k] + jb[(3+k)*nv+i] * f[3+k];
    }
  }
  copyv(d->qacc_smooth, d->qfrc_smooth, nv);
  orc_solve_m(d, d->qacc_smooth);
}

/* force response of each row to jar = J a - aref [UPSTREAM mj_constraintUpdate, pyramidal cones] */
static void constraint_update(orc_data* d, const double* jar, double* force) {
  for (int i = 0; i < d->nefc; i++) {
    double D = d->efc_D[i];
    switch (d->efc_type[i]) {
      case MJH_CNSTR_EQUALITY: force[i] = -D * jar[i]; break;
      case MJH_CNSTR_FRICTION_DOF: {
        double f = d->efc_frictionloss[i], R = d->efc_R[i];
        if (jar[i] <= -R*f) force[i] = f; else if (jar[i] >= R*f) force[i] = -f; else force[i] = -D * jar[i];
      } break;
      default: force[i] = jar[i] < 0 ? -D * jar[i] : 0; break;
    }
  }
}

/* Gauss-Seidel visiting order.  Default: plain constraint-row order (mj_solPGS).  The LEGACY orders of the HIP path (HISTORY.md §5;
 * reachable with orc_set_pgs_row_order(0) / mjh_set_pgs_row_order(0)) group the rows into blocks (one per equality / friction-loss /
 * limit row, one per contact = its pyramid rows) and visit them in the greedy "independent pair" order — block i, then the first
 * later unvisited block that shares no kinematic tree with it.  Any permutation is a valid PGS order; the converged solution is the
 * same, the iterates at the sweep cap are not. */
static void block_trees(const orc_data* d, int row, int* t1, int* t2) {
  const mjh_model* m = d->m; int id = d->efc_id[row];
  *t1 = *t2 = -1;
  switch (d->efc_type[row]) {
    case MJH_CNSTR_EQUALITY:
      if (m->eq_type[id] != MJH_EQ_JOINT) { *t1 = m->body_treeid[m->eq_obj1id[id]]; *t2 = m->body_treeid[m->eq_obj2id[id]]; break; }
      *t1 = m->dof_treeid[m->jnt_dofadr[m->eq_obj1id[id]]];
      if (m->eq_obj2id[id] >= 0) *t2 = m->dof_treeid[m->jnt_dofadr[m->eq_obj2id[id]]];
      break;
    case MJH_CNSTR_FRICTION_DOF: *t1 = m->dof_treeid[id]; break;
    case MJH_CNSTR_LIMIT_JOINT: *t1 = m->dof_treeid[m->jnt_dofadr[id]]; break;
    default:
      *t1 = m->body_treeid[m->geom_bodyid[d->contact[id].geom1]];
      *t2 = m->body_treeid[m->geom_bodyid[d->contact[id].geom2]];
  }
}
/* 1 (default): plain constraint-row order, the order mj_solPGS visits the rows in [UPSTREAM] — and the device's default: it runs this
 * very sequence, blocks without a common kinematic tree side by side (they commute exactly: patch_pgs.h, step_kernel.h "list schedule").
 * 0: the device's legacy orders (mjh_set_pgs_row_order(0)) — contact patches (m_patch_order below) or the independent-pair /
 * independent-group order above.  Both are Gauss-Seidel on the same dual problem: they agree at convergence; where the sweep cap ends
 * the iteration first (settled S24 piles run into the default 100 sweeps at tolerance 1e-8) the iterates differ, and
 * tests/test_oracle_pinning.py measures by how much. */
static int g_pgs_row_order = 1;
/* 2: the constraint-row order as the DEVICE walks it by default — the same sequence, list-scheduled (pgs_order below): blocks that
 * share no kinematic tree swap places, nothing else does.  Exists so that a test can show the claim the device relies on in fp64 too:
 * the iterates of 2 equal those of 1 bit for bit (tests/test_oracle_pinning.py). */
void orc_set_pgs_row_order(int mode) { g_pgs_row_order = mode < 0 || mode > 2 ? 1 : mode; }
static int m_group_max(const mjh_model* m) {
  if (m->ntree > 64) return 4;
  for (int t = 0; t < m->ntree; t++) if (m->tree_dofnum[t] > 8) return 4;
  return 16;
}
/* Contact-patch order (the device's patch_pgs.h): models whose trees are all single free bodies about their own centre of mass
 * with body-aligned principal axes (diagonal M), at most 32 dofs, no noslip pass, at most 64 contacts.  -1: by that rule (default); 0 / 1: forced off / on (an engine tells which order it runs: mjh_solver_order). */
static int g_pgs_patch_order = -1;
void orc_set_pgs_patch_order(int mode) { g_pgs_patch_order = mode < 0 ? -1 : (mode != 0); }
static int m_patch_order(const mjh_model* m) {
  if (g_pgs_patch_order >= 0) return g_pgs_patch_order;
  if (m->ntree <= 0 || m->neq != 0 || m->nsensor != 0 || m->nmocap != 0 || m->nv > 32 || m->maxcon > 64 || m->opt.noslip_iterations != 0) return 0;
  for (int t = 0; t < m->ntree; t++) {
    int b = m->tree_bodyid[t];
    if (m->tree_dofnum[t] != 6 || m->body_jntnum[b] != 1 || m->jnt_type[m->body_jntadr[b]] != MJH_JNT_FREE) return 0;
    for (int c = 1; c < m->nbody; c++) if (c != b && m->body_treeid[c] == t) return 0;
    if (m->body_ipos[3*b] != 0 || m->body_ipos[3*b+1] != 0 || m->body_ipos[3*b+2] != 0 || m->body_iquat[4*b] != 1) return 0;
  }
  for (int i = 0; i < m->nv; i++) if (m->dof_frictionloss[i] > 0) return 0;
  return 1;
}
/* Models whose sweeps are sequential on the device whatever the order (more than 32 dofs and a kinematic tree of more than 16:
 * articulated robots — one block at a time on a whole wavefront, or the dense row-space solver) are solved in mj_solPGS's own row
 * order by default: there it costs nothing.  Same rule as engine.hip (derive_device_model: M.pgs_row_order). */
static int m_row_order(const mjh_model* m) {
  if (m->nv <= 32) return 0;
  for (int t = 0; t < m->ntree; t++) if (m->tree_dofnum[t] > 16) return 1;
  return 0;
}
static int pgs_order(const orc_data* d, int* order) {
  int nefc = d->nefc, nblk = 0;
  if (g_pgs_row_order == 1 || (!g_pgs_row_order && m_row_order(d->m))) { for (int i = 0; i < nefc; i++) order[i] = i; return nefc; }
  int* bstart = d->scr_int;                       /* (nefc + 1) * 5 ints; the sequences below take the next (nefc + 1) * 6 */
  int *bnum = bstart + nefc + 1, *bt1 = bnum + nefc + 1, *bt2 = bt1 + nefc + 1, *used = bt2 + nefc + 1;
  for (int i = 0; i < nefc;) {
    int n = 1;
    if (d->efc_type[i] == MJH_CNSTR_CONTACT_PYRAMIDAL)
      while (i + n < nefc && d->efc_type[i+n] == MJH_CNSTR_CONTACT_PYRAMIDAL && d->efc_id[i+n] == d->efc_id[i]) n++;
    bstart[nblk] = i; bnum[nblk] = n; block_trees(d, i, bt1 + nblk, bt2 + nblk); used[nblk] = 0;
    nblk++; i += n;
  }
  int k = 0;
  if (g_pgs_row_order == 2) {
    /* the device's default walk of the constraint order (patch_pgs.h: patch_build; step_kernel.h: "list schedule"; the places per step
     * differ by kernel form — every such schedule gives the same iterates): units = blocks, or
     * (patch models) maximal runs of consecutive contacts of one body pair with at most 16 rows; unit i goes to the first step with
     * a free place after every EARLIER unit that shares a tree with it; the steps are emitted one after the other.  Relative order
     * of any two units with a common tree = constraint order. */
    int* seq = d->scr_int + (size_t)(nefc + 1) * 5;
    int *ufirst = seq, *ucount = ufirst + nefc + 1, *ustep = ucount + nefc + 1, *scount = ustep + nefc + 1, *tlast = scount + nefc + 1;
    int patches = m_patch_order(d->m), cap = patches ? 4 : (nblk > 64 ? m_group_max(d->m) : 2), nu = 0, rows = 0, nstep = 0;
    for (int i = 0; i < nblk; i++) {
      int t1 = bt1[i], t2 = bt2[i];
      if (t1 < 0) { t1 = t2; t2 = -1; }
      if (t2 == t1) t2 = -1;
      if (t2 >= 0 && t2 < t1) { int t = t1; t1 = t2; t2 = t; }
      bt1[i] = t1; bt2[i] = t2;
      int same = patches && i > 0 && bt1[i-1] == t1 && bt2[i-1] == t2;
      if (!same || rows + bnum[i] > 16) { ufirst[nu] = i; ucount[nu] = 0; nu++; rows = 0; }
      ucount[nu-1]++; rows += bnum[i];
    }
    for (int s = 0; s <= nu; s++) scount[s] = 0;
    for (int t = 0; t < d->m->ntree && t <= nefc; t++) tlast[t] = 0;
    if (d->m->ntree > nefc + 1) { for (int i = 0; i < nefc; i++) order[i] = i; return nefc; }   /* (scratch bound; never with the models at hand) */
    for (int u = 0; u < nu; u++) {
      int i = ufirst[u], e = 0;
      if (bt1[i] >= 0 && tlast[bt1[i]] > e) e = tlast[bt1[i]];
      if (bt2[i] >= 0 && tlast[bt2[i]] > e) e = tlast[bt2[i]];
      while (scount[e] >= cap) e++;
      scount[e]++; ustep[u] = e;
      if (bt1[i] >= 0) tlast[bt1[i]] = e + 1;
      if (bt2[i] >= 0) tlast[bt2[i]] = e + 1;
      if (e + 1 > nstep) nstep = e + 1;
    }
    for (int st = 0; st < nstep; st++)
      for (int u = 0; u < nu; u++)
        if (ustep[u] == st)
          for (int i = ufirst[u]; i < ufirst[u] + ucount[u]; i++) for (int r = 0; r < bnum[i]; r++) order[k++] = bstart[i] + r;
    return k;
  }
  if (m_patch_order(d->m)) {
    /* contacts sorted by (couples two bodies first, body pair, constraint order); a patch = a maximal run of contacts of one
     * body pair with at most 16 rows; a step = a patch plus up to three later unvisited patches of the sequence that share no
     * body with the step (first fit); rows in order inside a patch */
    int* seq = d->scr_int + (size_t)(nefc + 1) * 5;
    int *pfirst = seq + nblk + 1, *pcount = pfirst + nblk + 1, *pa = pcount + nblk + 1, *pb = pa + nblk + 1, *pused = pb + nblk + 1;
    long long* key = d->scr_key;
    for (int i = 0; i < nblk; i++) {
      int t1 = bt1[i], t2 = bt2[i];
      if (t1 < 0) { t1 = t2; t2 = -1; }
      if (t2 == t1) t2 = -1;
      int two = t2 >= 0, a = two ? (t1 < t2 ? t1 : t2) : t1, b = two ? (t1 < t2 ? t2 : t1) : 1000;
      bt1[i] = a; bt2[i] = two ? b : -1;
      key[i] = (((long long)(two ? 0 : 1) * 2048 + a) * 2048 + b) * 65536 + i;
      seq[i] = i;
    }
    for (int i = 1; i < nblk; i++) { int v = seq[i], j = i - 1; while (j >= 0 && key[seq[j]] > key[v]) { seq[j+1] = seq[j]; j--; } seq[j+1] = v; }
    int npatch = 0, rows = 0;
    for (int ii = 0; ii < nblk; ii++) {
      int i = seq[ii];
      int same = ii > 0 && bt1[seq[ii-1]] == bt1[i] && bt2[seq[ii-1]] == bt2[i];
      if (!same || rows + bnum[i] > 16) { pfirst[npatch] = ii; pcount[npatch] = 0; pa[npatch] = bt1[i]; pb[npatch] = bt2[i]; pused[npatch] = 0; npatch++; rows = 0; }
      pcount[npatch-1]++; rows += bnum[i];
    }
    for (int p = 0; p < npatch; p++) {
      if (pused[p]) continue;
      int gt[8], ngt = 0, cnt = 0;
      for (int c = p; c < npatch && cnt < 4; c++) {
        if (pused[c]) continue;
        int share = 0;
        for (int q = 0; q < ngt; q++) if (gt[q] == pa[c] || gt[q] == pb[c]) share = 1;
        if (share) continue;
        pused[c] = 1; cnt++;
        gt[ngt++] = pa[c]; if (pb[c] >= 0) gt[ngt++] = pb[c];
        for (int ii = pfirst[c]; ii < pfirst[c] + pcount[c]; ii++) { int i = seq[ii]; for (int r = 0; r < bnum[i]; r++) order[k++] = bstart[i] + r; }
      }
    }
    return k;
  }
  if (nblk > 64) {
    /* many-block models (the device solves them four independent blocks at a time, one per 16-lane row of a wave, on up to
     * four waves): same two-tree-first sequence; a group = a block plus up to GMAX - 1 later unvisited blocks of the
     * sequence, each sharing no tree with any block already in the group (first fit).  GMAX = 16 when the model has at most
     * 64 kinematic trees of at most 8 dofs each (free-body piles: BASELINE config C2), else 4. */
    int gmax = m_group_max(d->m);
    int* seq = d->scr_int + (size_t)(nefc + 1) * 5;
    int ns = 0;
    for (int pass = 0; pass < 2; pass++)
      for (int i = 0; i < nblk; i++) {
        int two = bt1[i] >= 0 && bt2[i] >= 0 && bt1[i] != bt2[i];
        if (two == (pass == 0)) seq[ns++] = i;
      }
    for (int ii = 0; ii < nblk; ii++) {
      int i = seq[ii];
      if (used[i]) continue;
      int gt[32], ngt = 0, cnt = 1;
      used[i] = 1;
      for (int r = 0; r < bnum[i]; r++) order[k++] = bstart[i] + r;
      if (bt1[i] >= 0) gt[ngt++] = bt1[i];
      if (bt2[i] >= 0) gt[ngt++] = bt2[i];
      for (int jj = ii + 1; jj < nblk && cnt < gmax; jj++) {
        int j = seq[jj];
        if (used[j]) continue;
        int share = 0;
        for (int q = 0; q < ngt; q++) if (gt[q] == bt1[j] || gt[q] == bt2[j]) share = 1;
        if (share) continue;
        used[j] = 1;
        for (int r = 0; r < bnum[j]; r++) order[k++] = bstart[j] + r;
        if (bt1[j] >= 0) gt[ngt++] = bt1[j];
        if (bt2[j] >= 0) gt[ngt++] = bt2[j];
        cnt++;
      }
    }
    return k;
  }
  /* visiting sequence: blocks that couple two kinematic trees first (they are the hard ones to pair), then the
   * single-tree blocks, each group in constraint order; then greedy: a block, and the first later unvisited block
   * of the sequence that shares no tree with it */
  int* seq = d->scr_int + (size_t)(nefc + 1) * 5;
  int ns = 0;
  for (int pass = 0; pass < 2; pass++)
    for (int i = 0; i < nblk; i++) {
      int two = bt1[i] >= 0 && bt2[i] >= 0 && bt1[i] != bt2[i];
      if (two == (pass == 0)) seq[ns++] = i;
    }
  for (int ii = 0; ii < nblk; ii++) {
    int i = seq[ii];
    if (used[i]) continue;
    used[i] = 1;
    for (int r = 0; r < bnum[i]; r++) order[k++] = bstart[i] + r;
    for (int jj = ii + 1; jj < nblk; jj++) {
      int j = seq[jj];
      if (used[j]) continue;
      int a1 = bt1[i], a2 = bt2[i], c1 = bt1[j], c2 = bt2[j];
      int share = (a1 >= 0 && (a1 == c1 || a1 == c2)) || (a2 >= 0 && (a2 == c1 || a2 == c2));
      if (share) continue;
      used[j] = 1;
      for (int r = 0; r < bnum[j]; r++) order[k++] = bstart[j] + r;
      break;
    }
  }
  return k;
}

/* warm start + projected Gauss-Seidel on the dual + map back [UPSTREAM mj_fwdConstraint / mj_solPGS] */
void orc_fwd_constraint(orc_data* d) {
  const mjh_model* m = d->m; int nv = m->nv, nefc = d->nefc;
  d->solver_iter = 0;
  if (!nefc) {
    copyv(d->qacc, d->qacc_smooth, nv); copyv(d->qacc_warmstart, d->qacc_smooth, nv); zero(d->qfrc_constraint, nv);
    return;
  }
  double *jar = d->scr_efc[0], *res = d->scr_efc[1];
  for (int i = 0; i < nefc; i++) d->efc_b[i] = dotn(d->efc_J + (size_t)i*nv, d->qacc_smooth, nv) - d->efc_aref[i];
  /* warm start: forces implied by qacc_warmstart, kept only if their dual cost is negative */
  if (!(m->opt.disableflags & MJH_DSBL_WARMSTART)) {
    for (int i = 0; i < nefc; i++) jar[i] = dotn(d->efc_J + (size_t)i*nv, d->qacc_warmstart, nv) - d->efc_aref[i];
    constraint_update(d, jar, d->efc_force);
    double cost = 0;
    for (int i = 0; i < nefc; i++) {
      res[i] = dotn(d->efc_AR + (size_t)i*nefc, d->efc_force, nefc);
      cost += 0.5 * d->efc_force[i] * res[i] + d->efc_force[i] * d->efc_b[i];
    }
    if (cost > 0) zero(d->efc_force, nefc);
  } else zero(d->efc_force, nefc);
  /* PGS sweeps */
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  int iter = 0;
  int* order = d->scr_int + (size_t)(m->maxefc + 1) * 11;   /* behind pgs_order's own scratch */
  pgs_order(d, order);
  while (iter < m->opt.iterations) {
    double improvement = 0;
    for (int k = 0; k < nefc; k++) {
      const int i = order[k];
      double Aii = d->efc_AR[(size_t)i*nefc + i];
      double r = d->efc_b[i] + dotn(d->efc_AR + (size_t)i*nefc, d->efc_force, nefc);
      double old = d->efc_force[i], f = old - r / Aii;
      if (d->efc_type[i] == MJH_CNSTR_FRICTION_DOF) { double fl = d->efc_frictionloss[i]; if (f < -fl) f = -fl; else if (f > fl) f = fl; }
      else if (d->efc_type[i] != MJH_CNSTR_EQUALITY) { if (f < 0) f = 0; }
      double delta = f - old, change = 0.5 * delta*delta * Aii + delta * r;
      if (change > 1e-10) { f = old; change = 0; }
      d->efc_force[i] = f;
      improvement -= change;
    }
    iter++;
    if (improvement * scale < m->opt.tolerance) break;
  }
  /* noslip post-pass [UPSTREAM mj_solNoSlip; option set by model/ontology/scene.xml:2-3]: further Gauss-Seidel sweeps over
   * the friction dimensions only, WITHOUT the regulariser R (so that resting contacts do not creep): dof friction-loss rows
   * are re-solved with A_ii - R_i, and each opposing pair of pyramid edges (j, j+1) of a contact is re-solved along
   * f_j - f_j+1 with f_j + f_j+1 (the normal-force share) held; equality, limit and frictionless rows are untouched. */
  if (m->opt.noslip_iterations > 0) {
    int nit = 0;
    while (nit < m->opt.noslip_iterations) {
      double improvement = 0;
      for (int k = 0; k < nefc; k++) {
        const int i = order[k];
        if (d->efc_type[i] == MJH_CNSTR_FRICTION_DOF) {
          double Aii = d->efc_AR[(size_t)i*nefc + i] - d->efc_R[i];
          if (Aii < MINVAL) continue;
          double r = d->efc_b[i] + dotn(d->efc_AR + (size_t)i*nefc, d->efc_force, nefc) - d->efc_R[i] * d->efc_force[i];
          double old = d->efc_force[i], f = old - r / Aii, fl = d->efc_frictionloss[i];
          if (f < -fl) f = -fl; else if (f > fl) f = fl;
          double delta = f - old;
          d->efc_force[i] = f;
          improvement -= 0.5 * delta*delta * Aii + delta * r;
        } else if (d->efc_type[i] == MJH_CNSTR_CONTACT_PYRAMIDAL && ((i - d->contact[d->efc_id[i]].efc_address) & 1) == 0) {
          const int j = i, q = i + 1;       /* the two opposing edges of one friction direction */
          const double* Aj = d->efc_AR + (size_t)j*nefc; const double* Aq = d->efc_AR + (size_t)q*nefc;
          double rj = d->efc_b[j] + dotn(Aj, d->efc_force, nefc) - d->efc_R[j] * d->efc_force[j];
          double rq = d->efc_b[q] + dotn(Aq, d->efc_force, nefc) - d->efc_R[q] * d->efc_force[q];
          double K1 = (Aj[j] - d->efc_R[j]) + (Aq[q] - d->efc_R[q]) - 2 * Aj[q];
          if (K1 < MINVAL) continue;
          double mid = 0.5 * (d->efc_force[j] + d->efc_force[q]), y0 = 0.5 * (d->efc_force[j] - d->efc_force[q]);
          double y = y0 - (rj - rq) / K1;
          if (y < -mid) y = -mid; else if (y > mid) y = mid;
          double dy = y - y0;
          d->efc_force[j] = mid + y; d->efc_force[q] = mid - y;
          improvement -= 0.5 * dy*dy * K1 + dy * (rj - rq);
        }
      }
      nit++;
      if (improvement * scale < m->opt.noslip_tolerance) break;
    }
    iter += nit;
  }
  d->solver_iter = iter;
  /* qfrc_constraint = J^T f ; qacc = qacc_smooth + M^-1 qfrc_constraint */
  zero(d->qfrc_constraint, nv);
  for (int i = 0; i < nefc; i++) { double f = d->efc_force[i]; if (f != 0) for (int q = 0; q < nv; q++) d->qfrc_constraint[q] += d->efc_J[(size_t)i*nv+q] * f; }
  copyv(d->qacc, d->qfrc_constraint, nv);
  orc_solve_m(d, d->qacc);
  for (int q = 0; q < nv; q++) d->qacc[q] += d->qacc_smooth[q];
  copyv(d->qacc_warmstart, d->qacc, nv);
}

/* semi-implicit Euler with implicit joint damping [UPSTREAM mj_Euler]; the step1/step2 split of
 * mj_main.cpp:83,108 always lands here, whatever integrator="RK4" says in the XMLs */
void orc_euler(orc_data* d) {
  const mjh_model* m = d->m; int nv = m->nv;
  double* qacc = d->scr_nv[0];
  int damped = 0;
  if (!(m->opt.disableflags & MJH_DSBL_EULERDAMP)) for (int i = 0; i < nv; i++) if (m->dof_damping[i] > 0) { damped = 1; break; }
  if (!damped) copyv(qacc, d->qacc, nv);
  else {
    double *MhB = d->scr_nM, *dinv = d->scr_nv[1];
    for (int i = 0; i < nv; i++) qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    copyv(MhB, d->qM, m->nM);
    for (int i = 0; i < nv; i++) MhB[m->dof_Madr[i]] += m->opt.timestep * m->dof_damping[i];
    factor_i(m, MhB, dinv);
    solve_ld(m, qacc, MhB, dinv);
  }
  double h = m->opt.timestep;
  for (int i = 0; i < nv; i++) {
    int bd = m->dof_bodyid[i];
    unsigned rb = (unsigned)bd - (m->nbody > 32 ? (unsigned)(m->nbody - 32) : 0u);
    if (rb < 32u && ((d->slot_mask >> rb) & 1u)) { d->qvel[i] = 0; d->qacc[i] = 0; d->qacc_warmstart[i] = 0; }  /* parked slot */
    else d->qvel[i] += h * qacc[i];
  }
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    switch (m->jnt_type[j]) {
      case MJH_JNT_FREE:
        for (int k = 0; k < 3; k++) d->qpos[qa+k] += h * d->qvel[da+k];
        quat_integrate(d->qpos + qa + 3, d->qvel + da + 3, h); break;
      case MJH_JNT_BALL: quat_integrate(d->qpos + qa, d->qvel + da, h); break;
      default: d->qpos[qa] += h * d->qvel[da];
    }
  }
  d->time += h;
}

/* MjSim::controller, mj_sim.cpp:1055-1077 (installed as mjcb_control, mj_main.cpp:196) */
void orc_controller(orc_data* d) {
  int nv = d->m->nv;
  orc_mul_m(d, d->tau, d->ddq);                                        /* :1057 */
  for (int i = 0; i < nv; i++) if (d->controlled[i]) d->tau[i] += d->qfrc_bias[i];  /* :1058-1063 */
  copyv(d->qfrc_applied, d->tau, nv);                                  /* :1065 */
  for (int i = 0; i < nv; i++) if (fabs(d->dq[i]) > MINVAL) d->qvel[i] = d->dq[i];  /* :1067-1073 */
  zero(d->ddq, nv); zero(d->dq, nv);                                   /* :1075-1076 */
}

/* MjSim::set_odom_vels, mj_sim.cpp:1079-1153 (one robot) */
void orc_set_odom_vels(orc_data* d) {
  double ax = d->odom_angq[0] >= 0 ? d->qpos[d->odom_angq[0]] : 0, ay = d->odom_angq[1] >= 0 ? d->qpos[d->odom_angq[1]] : 0,
         az = d->odom_angq[2] >= 0 ? d->qpos[d->odom_angq[2]] : 0;
  const double* v = d->odom_vel;
  if (d->odom_lin[0] >= 0) d->qvel[d->odom_lin[0]] = v[0]*cos(ay)*cos(az) + v[1]*(sin(ax)*sin(ay)*cos(az) - cos(ax)*sin(az)) + v[2]*(cos(ax)*sin(ay)*cos(az) + sin(ax)*sin(az));
  if (d->odom_lin[1] >= 0) d->qvel[d->odom_lin[1]] = v[0]*cos(ay)*sin(az) + v[1]*(sin(ax)*sin(ay)*sin(az) + cos(ax)*cos(az)) + v[2]*(cos(ax)*sin(ay)*sin(az) - sin(ax)*cos(az));
  if (d->odom_lin[2] >= 0) d->qvel[d->odom_lin[2]] = -v[0]*sin(ay) + v[1]*sin(ax)*cos(ay) + v[2]*cos(ax)*cos(ay);
  for (int k = 0; k < 3; k++) if (d->odom_ang[k] >= 0) d->qvel[d->odom_ang[k]] = v[3+k];
}

/* [UPSTREAM mj_sensorAcc -> mj_rnePostConstraint, force / torque sensors]: cfrc_ext = external forces on every body
 * (xfrc_applied, contact forces, connect / weld forces) as spatial forces about the subtree COM of the body's tree root;
 * cacc with the solved qacc; cfrc_int = cinert cacc + cvel x* (cinert cvel) - cfrc_ext accumulated up the tree = the force the
 * parent exerts on the body's subtree; a sensor reports it at its site, in the site frame */
static void add_ext(orc_data* d, int b, const double* point, const double* force, const double* torque, double sign) {
  if (b <= 0) return;
  const double* com = d->subtree_com + 3*d->m->body_rootid[b];
  double off[3] = {point[0] - com[0], point[1] - com[1], point[2] - com[2]}, t[3];
  cross3(t, off, force);
  for (int k = 0; k < 3; k++) { d->cfrc_ext[6*b+k] += sign * (torque[k] + t[k]); d->cfrc_ext[6*b+3+k] += sign * force[k]; }
}
void orc_sensor_acc(orc_data* d) {
  const mjh_model* m = d->m; int nb = m->nbody, nv = m->nv;
  if (m->nsensor == 0) return;
  /* body velocities of the CURRENT qvel: the reference runs mj_inverse (-> mj_fwdVelocity) between step1 and step2 every step
   * (mj_hw_interface.cpp:61), so d->cvel / cdof_dot always reflect the controller's velocity override when sensors are read */
  orc_com_vel(d);
  zero(d->cfrc_ext, 6*nb);
  for (int b = 1; b < nb; b++) add_ext(d, b, d->xipos + 3*b, d->xfrc_applied + 6*b, d->xfrc_applied + 6*b + 3, 1.0);
  for (int ic = 0; ic < d->ncon; ic++) {
    const orc_contact* c = d->contact + ic;
    if (c->exclude || c->efc_address < 0) continue;
    const double* f = d->efc_force + c->efc_address;
    double cf[3] = {0, 0, 0}, tors = 0;      /* contact-frame force on geom2's body (normal, tangent 1, tangent 2), torque about the normal */
    if (c->dim == 1) cf[0] = f[0];
    else for (int k = 1; k < c->dim; k++) {
      const double mu = c->friction[k-1], fp = f[2*(k-1)], fm = f[2*(k-1)+1];
      cf[0] += fp + fm;
      if (k < 3) cf[k] += mu * (fp - fm); else tors += mu * (fp - fm);
    }
    double F[3], T[3];
    for (int k = 0; k < 3; k++) { F[k] = c->frame[k]*cf[0] + c->frame[3+k]*cf[1] + c->frame[6+k]*cf[2]; T[k] = c->frame[k] * tors; }
    add_ext(d, m->geom_bodyid[c->geom2], c->pos, F, T, 1.0);
    add_ext(d, m->geom_bodyid[c->geom1], c->pos, F, T, -1.0);
  }
  for (int i = 0; i < d->nefc; i++) {          /* connect / weld: rows of one equality are consecutive */
    if (d->efc_type[i] != MJH_CNSTR_EQUALITY || m->eq_type[d->efc_id[i]] == MJH_EQ_JOINT) continue;
    const int e = d->efc_id[i], weld = m->eq_type[e] == MJH_EQ_WELD, b1 = m->eq_obj1id[e], b2 = m->eq_obj2id[e];
    const double* dat = m->eq_data + 11*e;
    const double *a1 = weld ? dat + 3 : dat, *a2 = weld ? dat : dat + 3, zero3[3] = {0, 0, 0};
    double p1[3], p2[3], t[3];
    rotvec(t, d->xmat + 9*b1, a1); for (int k = 0; k < 3; k++) p1[k] = d->xpos[3*b1+k] + t[k];
    rotvec(t, d->xmat + 9*b2, a2); for (int k = 0; k < 3; k++) p2[k] = d->xpos[3*b2+k] + t[k];
    add_ext(d, b1, p1, d->efc_force + i, zero3, 1.0);            /* rows p1 - p2: force +f on body1 at p1, -f on body2 at p2 */
    add_ext(d, b2, p2, d->efc_force + i, zero3, -1.0);
    if (weld) {                                                   /* rows 3..5: cpos' = A (w2 - w1): torque A^T f on body2, -A^T f on body1 */
      const double ts = dat[10];
      double q1i[4] = {d->xquat[4*b1], -d->xquat[4*b1+1], -d->xquat[4*b1+2], -d->xquat[4*b1+3]}, q2r[4], T[3] = {0, 0, 0};
      mulquat(q2r, d->xquat + 4*b2, dat + 6);
      for (int a = 0; a < 3; a++) {
        double w[4] = {0, a == 0, a == 1, a == 2}, u[4], v[4];
        mulquat(u, q1i, w); mulquat(v, u, q2r);
        for (int k = 0; k < 3; k++) T[a] += 0.5 * ts * v[1+k] * d->efc_force[i + 3 + k];
      }
      add_ext(d, b2, p2, zero3, T, 1.0); add_ext(d, b1, p1, zero3, T, -1.0);
    }
    i += weld ? 5 : 2;
  }
  zero(d->cacc, 6); zero(d->cfrc_int, 6);
  if (!(m->opt.disableflags & MJH_DSBL_GRAVITY)) for (int k = 0; k < 3; k++) d->cacc[3+k] = -m->opt.gravity[k];
  for (int i = 1; i < nb; i++) {
    int bda = m->body_dofadr[i], p = m->body_parentid[i];
    double tmp[6], tmp1[6];
    copyv(d->cacc + 6*i, d->cacc + 6*p, 6);
    for (int k = 0; k < m->body_dofnum[i]; k++) for (int q = 0; q < 6; q++)
      d->cacc[6*i+q] += d->cdof_dot[6*(bda+k)+q] * d->qvel[bda+k] + d->cdof[6*(bda+k)+q] * d->qacc[bda+k];
    mul_inert_vec(d->cfrc_int + 6*i, d->cinert + 10*i, d->cacc + 6*i);
    mul_inert_vec(tmp, d->cinert + 10*i, d->cvel + 6*i);
    cross_force(tmp1, d->cvel + 6*i, tmp);
    for (int q = 0; q < 6; q++) d->cfrc_int[6*i+q] += tmp1[q] - d->cfrc_ext[6*i+q];
  }
  for (int i = nb - 1; i > 0; i--) { int p = m->body_parentid[i]; if (p > 0) for (int q = 0; q < 6; q++) d->cfrc_int[6*p+q] += d->cfrc_int[6*i+q]; }
  (void)nv;
  for (int s = 0; s < m->nsensor; s++) {
    const int site = m->sensor_objid[s], b = m->site_bodyid[site];
    const double* com = d->subtree_com + 3*m->body_rootid[b];
    const double* F = d->cfrc_int + 6*b;
    double off[3] = {d->site_xpos[3*site] - com[0], d->site_xpos[3*site+1] - com[1], d->site_xpos[3*site+2] - com[2]}, t[3], tq[3];
    cross3(t, off, F + 3);
    for (int k = 0; k < 3; k++) tq[k] = F[k] - t[k];                      /* torque about the site */
    rotvecT(d->sensordata + m->sensor_adr[s], d->site_xmat + 9*site, m->sensor_type[s] == MJH_SENS_FORCE ? F + 3 : tq);
  }
}

void orc_step1(orc_data* d) {
  check_state(d);
  orc_fwd_position(d); orc_energy(d); orc_fwd_velocity(d);
  orc_controller(d);
}
void orc_step2(orc_data* d) {
  orc_fwd_acceleration(d); orc_fwd_constraint(d); orc_sensor_acc(d); check_state(d); orc_euler(d);
  orc_set_odom_vels(d);
}
void orc_forward(orc_data* d) {
  check_state(d);
  orc_fwd_position(d); orc_energy(d); orc_fwd_velocity(d); orc_controller(d);
  orc_fwd_acceleration(d); orc_fwd_constraint(d); orc_sensor_acc(d);
}

/* mj_inverse as called by MjHWInterface::read (mj_hw_interface.cpp:61) [UPSTREAM mj_inverse]:
 * position stage, velocity stage, analytic constraint force at the CURRENT qacc, RNE with
 * acceleration.  The position stage is identical to step1's (qpos unchanged), so it is reused. */
void orc_inverse(orc_data* d) {
  const mjh_model* m = d->m; int nv = m->nv, nefc = d->nefc;
  orc_fwd_velocity(d); /* mj_invVelocity == mj_fwdVelocity: sees the velocity override of the controller */
  double *jar = d->scr_efc[0], *force = d->scr_efc[2], *qc = d->scr_nv[2];
  for (int i = 0; i < nefc; i++) jar[i] = dotn(d->efc_J + (size_t)i*nv, d->qacc, nv) - d->efc_aref[i];
  constraint_update(d, jar, force);
  zero(qc, nv);
  for (int i = 0; i < nefc; i++) if (force[i] != 0) for (int q = 0; q < nv; q++) qc[q] += d->efc_J[(size_t)i*nv+q] * force[i];
  orc_rne(d, 1, d->qfrc_inverse);
  for (int i = 0; i < nv; i++) d->qfrc_inverse[i] += m->dof_armature[i] * d->qacc[i] - d->qfrc_passive[i] - qc[i];
}

/* loop body of simulate(), mj_main.cpp:82-112, without ROS: step1 -> read() -> write() -> step2 */
void orc_set_pd(orc_data* d, const double* target, double kp, double kd) {
  const int nv = d->m->nv;
  if (!d->pd_target) d->pd_target = dalloc((size_t)nv);
  if (target) copyv(d->pd_target, target, nv);
  d->pd_kp = kp; d->pd_kd = kd;
}
void orc_step(orc_data* d, int nsteps, int with_inverse) {
  const mjh_model* m = d->m;
  for (int s = 0; s < nsteps; s++) {
    if (d->pd_target && (d->pd_kp > 0 || d->pd_kd > 0))      /* controller_manager->update() + write(), mj_main.cpp:99-106 */
      for (int i = 0; i < m->nv; i++) {
        const int j = m->dof_jntid[i];
        if (m->jnt_type[j] == MJH_JNT_HINGE || m->jnt_type[j] == MJH_JNT_SLIDE)
          d->ddq[i] = d->pd_kp * (d->pd_target[i] - d->qpos[m->jnt_qposadr[j]]) - d->pd_kd * d->qvel[i];
      }
    orc_step1(d);
    if (with_inverse) orc_inverse(d);
    orc_step2(d);
  }
}

static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }
/* CPU baseline driver: envs split over OpenMP threads, private orc_data per env, shared read-only model */
void orc_step_many(orc_data** ds, int nenv, int nsteps, int with_inverse) {
  /* one env at a time to whichever thread is free (an env at the 100-sweep cap costs up to 50x a converged one: a static split
   * leaves most threads idle behind the heaviest share — measured), no allocation inside a step */
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
  for (int e = 0; e < nenv; e++) orc_step(ds[e], nsteps, with_inverse);
}
/* The timed variant bench.py's cpu_baseline uses: `warm_steps` untimed steps of every env, then `nsteps` timed ones, on a team of
 * g_threads plain pthreads that NEVER sleep between the phases (spin barriers on C11 atomics), envs handed out one at a time from
 * an atomic counter.  Why not the OpenMP loop above: measured in the VMs this runs in, a sleeping thread takes tens to hundreds
 * of milliseconds to be woken (a halted vCPU has to be rescheduled by the host), so a region entered cold — or one whose
 * threads dozed off at a barrier — shows no speed-up at all (2 threads 0.93x, 256 threads 1.04x) although the steps scale.
 * Returns the seconds of the timed part, clocked from the moment every thread has passed the barrier behind the warm steps. */
#include <pthread.h>
#include <stdatomic.h>
#include <time.h>
typedef struct {
  orc_data** ds; int nenv, warm_steps, nsteps, with_inverse, nthreads;
  atomic_int next_warm, next_timed, arrived[3];
  atomic_int* rounds_done;                               /* per env: timed chunks completed (keeps an env's chunks in order) */
  int chunk;
  double t0, t1;
} orc_team;
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static void spin_barrier(orc_team* t, int k) {
  atomic_fetch_add(&t->arrived[k], 1);
  while (atomic_load(&t->arrived[k]) < t->nthreads) { /* busy wait: no futex, no halt */ }
}
static void* team_main(void* arg) {
  orc_team* t = (orc_team*)arg;
  spin_barrier(t, 0);                                    /* every thread exists and runs */
  for (int e; (e = atomic_fetch_add(&t->next_warm, 1)) < t->nenv;) orc_step(t->ds[e], t->warm_steps, t->with_inverse);
  if (atomic_fetch_add(&t->arrived[1], 1) == t->nthreads - 1) t->t0 = now_s();   /* the last one in starts the clock ... */
  while (atomic_load(&t->arrived[1]) < t->nthreads) {}
  /* timed work in items of (env, `chunk` steps), handed out round by round: an env at the 100-sweep cap costs many times a
   * converged one, and whole-env items would leave the team waiting for whoever drew the heaviest envs */
  const int nround = (t->nsteps + t->chunk - 1) / t->chunk;
  for (int k; (k = atomic_fetch_add(&t->next_timed, 1)) < t->nenv * nround;) {
    const int e = k % t->nenv, r = k / t->nenv;
    while (atomic_load(&t->rounds_done[e]) < r) {}
    const int n = (r + 1) * t->chunk <= t->nsteps ? t->chunk : t->nsteps - r * t->chunk;
    orc_step(t->ds[e], n, t->with_inverse);
    atomic_store(&t->rounds_done[e], r + 1);
  }
  if (atomic_fetch_add(&t->arrived[2], 1) == t->nthreads - 1) t->t1 = now_s();   /* ... and the last one out stops it */
  return NULL;
}
double orc_step_many_timed(orc_data** ds, int nenv, int warm_steps, int nsteps, int with_inverse) {
  orc_team t; memset(&t, 0, sizeof t);
  t.ds = ds; t.nenv = nenv; t.warm_steps = warm_steps; t.nsteps = nsteps; t.with_inverse = with_inverse;
  t.nthreads = g_threads < nenv ? g_threads : nenv;
  if (t.nthreads < 1) t.nthreads = 1;
  t.chunk = getenv("ORC_CHUNK") ? atoi(getenv("ORC_CHUNK")) : 4; if (t.chunk < 1) t.chunk = 1;
  t.rounds_done = (atomic_int*)calloc((size_t)(nenv > 0 ? nenv : 1), sizeof(atomic_int));
  pthread_t* th = (pthread_t*)calloc((size_t)t.nthreads, sizeof(pthread_t));
  int made = 0;
  for (int k = 1; k < t.nthreads; k++) { if (pthread_create(&th[k], NULL, team_main, &t)) break; made++; }
  if (made != t.nthreads - 1) t.nthreads = made + 1;     /* (thread limit hit: the team is what could be created) */
  team_main(&t);
  for (int k = 1; k <= made; k++) pthread_join(th[k], NULL);
  free(th); free(t.rounds_done);
  return t.t1 - t.t0;
}

/* named-field access for the Python test harness */
double* orc_field(orc_data* d, const char* name, int* n) {
  const mjh_model* m = d->m; int nv = m->nv, nb = m->nbody, ne = d->nefc;
#define F(nm, ptr, cnt) if (!strcmp(name, nm)) { *n = (cnt); return (ptr); }
  F("qpos", d->qpos, m->nq) F("qvel", d->qvel, nv) F("qacc", d->qacc, nv) F("qacc_warmstart", d->qacc_warmstart, nv)
  F("qfrc_applied", d->qfrc_applied, nv) F("qfrc_bias", d->qfrc_bias, nv) F("qfrc_passive", d->qfrc_passive, nv)
  F("qfrc_smooth", d->qfrc_smooth, nv) F("qacc_smooth", d->qacc_smooth, nv) F("qfrc_constraint", d->qfrc_constraint, nv)
  F("qfrc_inverse", d->qfrc_inverse, nv) F("xpos", d->xpos, 3*nb) F("xquat", d->xquat, 4*nb) F("xmat", d->xmat, 9*nb)
  F("xipos", d->xipos, 3*nb) F("ximat", d->ximat, 9*nb) F("geom_xpos", d->geom_xpos, 3*m->ngeom) F("geom_xmat", d->geom_xmat, 9*m->ngeom)
  F("subtree_com", d->subtree_com, 3*nb) F("cinert", d->cinert, 10*nb) F("cdof", d->cdof, 6*nv) F("cdof_dot", d->cdof_dot, 6*nv)
  F("cvel", d->cvel, 6*nb) F("qM", d->qM, m->nM) F("qLD", d->qLD, m->nM) F("qLDiagInv", d->qLDiagInv, nv)
  F("efc_J", d->efc_J, ne*nv) F("efc_pos", d->efc_pos, ne) F("efc_R", d->efc_R, ne) F("efc_D", d->efc_D, ne)
  F("efc_aref", d->efc_aref, ne) F("efc_b", d->efc_b, ne) F("efc_force", d->efc_force, ne) F("efc_AR", d->efc_AR, ne*ne)
  F("efc_vel", d->efc_vel, ne) F("efc_diagApprox", d->efc_diagApprox, ne) F("efc_KBIP", d->efc_KBIP, 4*ne)
  F("energy", d->energy, 2) F("time", &d->time, 1) F("ddq", d->ddq, nv) F("dq", d->dq, nv) F("odom_vel", d->odom_vel, 6)
  F("initial_qpos", d->initial_qpos, m->nq) F("xfrc_applied", d->xfrc_applied, 6*nb) F("mocap_pos", d->mocap_pos, 3*m->nmocap)
  F("mocap_quat", d->mocap_quat, 4*m->nmocap) F("sensordata", d->sensordata, m->nsensordata) F("site_xpos", d->site_xpos, 3*m->nsite)
  F("site_xmat", d->site_xmat, 9*m->nsite) F("cfrc_int", d->cfrc_int, 6*nb) F("cfrc_ext", d->cfrc_ext, 6*nb) F("cacc", d->cacc, 6*nb)
#undef F
  *n = 0; return NULL;
}
int orc_int(orc_data* d, const char* name) {
  if (!strcmp(name, "ncon")) return d->ncon;
  if (!strcmp(name, "nefc")) return d->nefc;
  if (!strcmp(name, "solver_iter")) return d->solver_iter;
  if (!strcmp(name, "warn")) return d->warn;
  if (!strcmp(name, "slot_mask")) return (int)d->slot_mask;
  return -1;
}
void orc_set_slot_mask(orc_data* d, unsigned mask) { d->slot_mask = mask; }
int* orc_int_field(orc_data* d, const char* name, int* n) {
  if (!strcmp(name, "controlled")) { *n = d->m->nv; return d->controlled; }
  if (!strcmp(name, "efc_type")) { *n = d->nefc; return d->efc_type; }
  if (!strcmp(name, "efc_id")) { *n = d->nefc; return d->efc_id; }
  if (!strcmp(name, "odom_lin")) { *n = 3; return d->odom_lin; }
  if (!strcmp(name, "odom_ang")) { *n = 3; return d->odom_ang; }
  if (!strcmp(name, "odom_angq")) { *n = 3; return d->odom_angq; }
  *n = 0; return NULL;
}
/* contact k: fills dist, pos[3], frame[9], geom[2], dim */
int orc_get_contact(orc_data* d, int k, double* dist, double* pos, double* frame, int* geom, int* dim) {
  if (k < 0 || k >= d->ncon) return -1;
  orc_contact* c = d->contact + k;
  *dist = c->dist; copyv(pos, c->pos, 3); copyv(frame, c->frame, 9); geom[0] = c->geom1; geom[1] = c->geom2; *dim = c->dim;
  return 0;
}
