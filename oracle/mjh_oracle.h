/* mjh_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * fp64 CPU restatement of the step pipeline the reference drives through
 * MuJoCo 2.3.7 (mj_step1 / mj_inverse / mj_step2 as called at
 * /root/reference/src/mj_main.cpp:83,108 and
 * /root/reference/src/mujoco_sim/mj_hw_interface.cpp:61) plus the wrapper's own
 * controller / odom / ros_control glue (mj_sim.cpp:1055-1153,
 * mj_hw_interface.cpp:59-91).
 *
 * PARITY UNPINNED: the arithmetic lives in the third-party library
 * libmujoco.so 2.3.7 (tarball sha256 3f75e53e…ac39, fetched at build time by the
 * reference, Makefile:3-8) which is absent from /root/reference and from this
 * image, and the reference's tests hold no numeric expectations (SURVEY.md §4,
 * §8-c).  This file restates MuJoCo's published algorithm ("Computation" chapter)
 * from public knowledge; it is anchored by analytic known-answer tests
 * (tests/test_oracle_kat.py) instead of reference vectors.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#ifndef MJH_ORACLE_H_
#define MJH_ORACLE_H_

#include "../include/mjhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_contact {
  double dist, pos[3], frame[9], includemargin, friction[5], solref[2], solimp[5], mu;
  int dim, geom1, geom2, exclude, efc_address;
} orc_contact;

typedef struct orc_data {
  const mjh_model* m;
  double time;
  /* state + generalized vectors */
  double *qpos, *qvel, *qacc, *qacc_warmstart, *qfrc_applied, *qfrc_bias, *qfrc_passive;
  double *qfrc_smooth, *qacc_smooth, *qfrc_constraint, *qfrc_inverse;
  /* position stage */
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat;
  double *subtree_com, *cinert, *crb, *cdof, *cdof_dot, *cvel;
  double *qM, *qLD, *qLDiagInv;
  /* contacts / constraints */
  int ncon, nefc, solver_iter, warn; /* warn bit0 contact overflow, bit1 row overflow, bit2 bad state reset */
  orc_contact* contact;
  int *efc_type, *efc_id;
  double *efc_J, *efc_pos, *efc_margin, *efc_frictionloss, *efc_diagApprox, *efc_R, *efc_D;
  double *efc_KBIP, *efc_vel, *efc_aref, *efc_b, *efc_force, *efc_AR;
  double energy[2];
  /* per-env model overrides (NULL -> shared model) */
  double *geom_size, *geom_rbound, *body_mass, *body_inertia, *body_invweight0, *dof_invweight0;
  /* MjSim::ddq / dq / tau (mj_sim.h:96-100) and controlled-dof mask */
  double *ddq, *dq, *tau; int* controlled;
  int odom_lin[3], odom_ang[3], odom_angq[3]; double odom_vel[6];
  double* initial_qpos;
  unsigned slot_mask; /* bit b = body b is an inactive spawn/destroy slot */
  /* d->xfrc_applied [6*nbody] (force, torque per body at its centre of mass, world frame; mj_sim.cpp:499), d->mocap_pos /
   * mocap_quat [3 / 4 per mocap body] (mj_sim.cpp:903: the *_ref bodies), d->sensordata [nsensordata] (mj_ros.cpp:1933-1966),
   * site frames, and the body accelerations / interaction forces of mj_rnePostConstraint that the sensors read */
  double *xfrc_applied, *mocap_pos, *mocap_quat, *sensordata, *site_xpos, *site_xmat, *cacc, *cfrc_int, *cfrc_ext;
  /* joint-space PD effort controller evaluated in front of every step of orc_step (what ros_control's effort controllers do
   * between read() and write(), mj_main.cpp:86-106): ddq = kp (target - q) - kd qvel on hinge / slide dofs */
  double *pd_target, pd_kp, pd_kd;
  /* scratch */
  double *scr_nv[6], *scr_nM, *scr_efc[3], *scr_B, *scr_body6[3];
  double* scr_jac; int* scr_int; long long* scr_key; int* scr_nz;   /* per-step scratch owned by the data (no malloc inside a step) */
} orc_data;

orc_data* orc_make_data(const mjh_model* m);
void orc_free_data(orc_data* d);
void orc_set_env_param(orc_data* d, int which, const double* values);
void orc_reset(orc_data* d);

/* stages (names follow the MuJoCo stage each one restates) */
void orc_kinematics(orc_data* d);
void orc_com_pos(orc_data* d);
void orc_crb(orc_data* d);
void orc_factor_m(orc_data* d);
void orc_solve_m(const orc_data* d, double* x);          /* x <- M^-1 x */
void orc_mul_m(const orc_data* d, double* res, const double* vec); /* mj_mulM, mj_sim.cpp:1057 */
void orc_collision(orc_data* d);
void orc_make_constraint(orc_data* d);
void orc_project_constraint(orc_data* d);
void orc_com_vel(orc_data* d);
void orc_passive(orc_data* d);
void orc_reference_constraint(orc_data* d);
void orc_rne(orc_data* d, int flg_acc, double* result);
void orc_energy(orc_data* d);
void orc_sensor_acc(orc_data* d);    /* mj_sensorAcc -> mj_rnePostConstraint: force / torque sensors (end of mj_step2's forward part) */

void orc_fwd_position(orc_data* d);
void orc_fwd_velocity(orc_data* d);
void orc_fwd_acceleration(orc_data* d);
void orc_fwd_constraint(orc_data* d);
void orc_euler(orc_data* d);

void orc_controller(orc_data* d);     /* MjSim::controller, mj_sim.cpp:1055-1077 */
void orc_set_odom_vels(orc_data* d);  /* MjSim::set_odom_vels, mj_sim.cpp:1079-1153 */

void orc_step1(orc_data* d);  /* mj_main.cpp:83 (includes the mjcb_control callback) */
void orc_step2(orc_data* d);  /* mj_main.cpp:108-110 (includes set_odom_vels) */
void orc_forward(orc_data* d);/* mj_ros.cpp:608,1421 */
void orc_inverse(orc_data* d);/* mj_hw_interface.cpp:61 */
void orc_step(orc_data* d, int nsteps, int with_inverse); /* the loop body of mj_main.cpp:82-112 */

/* narrow-phase primitive exposed for unit tests: returns count, fills dist[8], pos[24], normal[3] */
int orc_box_box(const double* p1, const double* m1, const double* s1, const double* p2,
                const double* m2, const double* s2, double margin, double* dist, double* pos,
                double* normal);

double* orc_field(orc_data* d, const char* name, int* n);
int orc_int(orc_data* d, const char* name);
int* orc_int_field(orc_data* d, const char* name, int* n);
int orc_get_contact(orc_data* d, int k, double* dist, double* pos, double* frame, int* geom, int* dim);

/* multi-env convenience for the CPU baseline: steps `nenv` independent datas */
double orc_step_many_timed(orc_data** ds, int nenv, int warm_steps, int nsteps, int with_inverse);
void orc_step_many(orc_data** ds, int nenv, int nsteps, int with_inverse);
void orc_set_threads(int n);
void orc_set_slot_mask(orc_data* d, unsigned mask);
/* Gauss-Seidel visiting order of orc_fwd_constraint (process-wide): 0 = the device's independent-pair order (default),
 * 1 = plain constraint-row order as mj_solPGS [UPSTREAM] */
void orc_set_pgs_row_order(int plain);
void orc_set_pgs_patch_order(int mode);   /* -1: by the model rule (default), 0 / 1: independent-pair order / contact-patch order */
void orc_set_pd(orc_data* d, const double* target /* [nv], may be NULL with kp = kd = 0 */, double kp, double kd);

#ifdef __cplusplus
}
#endif
#endif
