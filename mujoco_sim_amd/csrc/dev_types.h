// dev_types.h — device-side model / state / LDS-layout descriptors shared by the
// host API (engine.hip) and the stepping kernel (step_kernel.h).
#pragma once

// ---- device model: every table lives in one int buffer and one float buffer ----
#define MJH_INT_TABLES(X)                                                                          \
  X(body_parentid) X(body_rootid) X(body_jntadr) X(body_jntnum) X(body_dofadr) X(body_dofnum)      \
  X(body_level) X(body_subtreesize) X(body_treeid) X(body_lastdof) X(jnt_type) X(jnt_qposadr)      \
  X(jnt_dofadr) X(jnt_bodyid) X(jnt_limited) X(dof_bodyid) X(dof_jntid) X(dof_parentid)            \
  X(dof_Madr) X(dof_treeid) X(tree_dofadr) X(tree_dofnum) X(geom_type) X(geom_bodyid)              \
  X(geom_condim) X(pair_geom1) X(pair_geom2) X(pair_stageadr) X(eq_obj1id) X(eq_obj2id)            \
  X(eq_active) X(fl_dof) X(gc_body) X(controlled) X(odom)

#define MJH_FLT_TABLES(X)                                                                          \
  X(body_pos) X(body_quat) X(body_ipos) X(body_iquat) X(body_mass) X(body_inertia)                 \
  X(body_gravcomp) X(body_invweight0) X(jnt_pos) X(jnt_axis) X(jnt_stiffness) X(jnt_range)         \
  X(jnt_margin) X(jnt_solref) X(jnt_solimp) X(qpos0) X(qpos_spring) X(dof_armature)                \
  X(dof_damping) X(dof_frictionloss) X(dof_invweight0) X(dof_solref) X(dof_solimp) X(geom_pos)     \
  X(geom_quat) X(geom_size) X(geom_rbound) X(geom_friction) X(geom_solmix) X(geom_solref)          \
  X(geom_solimp) X(geom_margin) X(geom_gap) X(eq_data) X(eq_solref) X(eq_solimp)

struct DModel {
  const int* I;
  const float* F;
#define X(n) int o_##n;
  MJH_INT_TABLES(X)
  MJH_FLT_TABLES(X)
#undef X
  int nq, nv, nbody, njnt, ngeom, neq, npair, nM, ntree, maxcon, maxefc;
  int nqp, nvp;        // padded row strides of the per-env state arrays (floats)
  int maxlevel, nfl, ngc, rowW, nstage, has_damping, has_limits, diagM, maxblk, maxbrow, has_dim4, big, k1_floats, has_convex;
  int iterations, disableflags;
  float timestep, gravity[3], tolerance, impratio, meaninertia;
  // convex mesh assets (read by the CONVEX kernel instances only)
  int o_geom_dataid, o_mesh_vertadr, o_mesh_vertnum, o_mesh_vert;
  // noslip post-pass (EXTRA instances)
  int noslip_iterations; float noslip_tolerance;
  double timestep_d;   // opt.timestep in fp64: per-env time advances by exactly this (time == n * dt after n steps)
  // sites, force / torque sensors, mocap bodies, connect / weld equalities (EXTRA kernel instances only)
  int nsite, nsensor, nmocap, has_weld;
  int group_max;     // Gauss-Seidel groups of many-block models hold up to 4 or 16 mutually independent blocks (16: <= 64 trees of <= 8 dofs)
  int scratch_off;   // many-body layout: LDS scratch of k1_floats floats (the dead position-stage arrays; an own region when sensors keep them alive)
  int o_site_bodyid, o_site_pos, o_site_quat, o_sensor_type, o_sensor_objid, o_body_mocapid, o_eq_type;
  // contact-patch sweep (patch_pgs.h; small free-body models): LDS float offsets of the patch pool (it reuses everything that
  // is dead once the solver starts, from the position-stage arrays to the base-row storage), its size, and of the two live
  // tables next to it: one descriptor per patch, one descriptor per (step, 16-lane row) of the sweep schedule
  int patch, pool, pool_floats, pdesc, pslot;
  // window sweep (window_pgs.h): the fused step of a patch-eligible model in row order as two launches, assemble (PH_PRE) -> mjh_window_kernel
  // (four envs per wavefront, rows in registers); win_nvt: dof slots of a row record (24 or 32)
  int window, win_nvt, win_maxw, win_jsz;     // win_jsz: floats of the base-row (J) pool, which the assemble-only launch keeps in the env's window slice instead of LDS;     // win_maxw: windows of 16 rows an env can hand over (min(WN_MAXW, ceil(maxefc / 16)); rows beyond are dropped with the capacity flag)
  int pgs_row_order;   // 1: Gauss-Seidel visits the constraint rows in their own order, one block after the other (mj_solPGS's order; mjh_set_pgs_row_order)
  // dense row-space solver of the many-body layout (dense_pgs.h): on / off, row capacity (a multiple of 64, <= 256), nv padded to 16
  int dense, dense_cap, dense_nvs;
  int dense_min_iter;   // a cohort takes the dense form while one of its envs swept at least this often in the step its launch order was built from (the host decides per launch)
};

// per-env state in HBM (fp32, env-major rows)
#define PROF_STRIDE 20   // x_prof: 16 stage stamps (shader clock), [16],[17] wall clock start/end, [18] HW_ID|XCC_ID<<32
struct DState {
  float *qpos, *qvel, *qacc, *qacc_ws, *qvel_ref, *qfrc_applied, *ddq, *dq, *qfrc_inverse;
  double* time;  // [nenv] simulation time in fp64 (d->time is mjtNum: ROS stamps, the 10 kHz gate and the RTF logic read it)
  float *initial_qpos, *odom_vel;
  int* stats;  // [nenv*4]: ncon, nefc, solver iterations, flags
  const int* env_order;  // launch order of the envs (longest job first) or null
  unsigned* slot_mask;  // [nenv] bit b = body b inactive (spawn/destroy slots); may be null
  // optional exports (may be null)
  float *x_xpos, *x_xquat, *x_gpos, *x_gmat;
  float *x_bias, *x_passive, *x_smooth, *x_constraint, *x_energy;
  float *x_contacts;  // [n * maxcon * 17]
  float *x_vec, *x_res;  // mulM in/out [n*nvp]
  long long* x_prof;     // [n*16] s_memtime stamps at stage boundaries (debug)
  // per-env model parameter tables (null -> shared model)
  // (one row of p_stride floats per env holds all six, so that an env's parameters cost whole sectors once, not six times)
  const float *p_geom_size, *p_geom_rbound, *p_body_mass, *p_body_inertia, *p_body_invweight0, *p_dof_invweight0;
  int p_stride;
  // optional per-env inputs / outputs of the EXTRA instances (null until used): Cartesian forces on bodies [nenv][xfrc_stride],
  // mocap poses [nenv][7*nmocap] (pos3 quat4), sensor outputs [nenv][3*nsensor]
  float *xfrc_applied, *mocap, *sensordata; int xfrc_stride;
  // in-engine PD law (mjh_set_pd_controller) evaluated by the controller stage of a fused step: targets [nenv][nv] or null, gains
  const float* pd_target; float pd_kp, pd_kd;
  // many-body models (nv > 64): per-env pools that do not fit LDS (contacts, blocks, Jacobians) live here; a negative
  // Lay offset -1-k addresses float k of the env's slice
  float* gscratch; long long gstride;
  int win32;             // window kernel: envs with more than this many rows (default WN32_MIN_ROWS) are swept in 32-row windows, two per wavefront (0: every env in the 16-row form)
  int win64;             // ... with more than this many rows (default WN64_MIN_ROWS, at most 64 WN64_NW) in 64-row windows, one env per wavefront (0: off)
  int probe_slices;      // debug (mjh_debug_solve_probe): mjh_solve_kernel reads the block operands of env0 + blockIdx % probe_slices instead of its own (0: off)
  // window sweep (window_pgs.h): per-env hand-over slice (header, scaled dof vectors, state, window rows, tiles of streamed windows)
  float* wbuf; long long wstride; long long wj_off;     // wj_off: where the assemble-only launch's base-row pool starts inside the env's slice
};

// LDS layout (float offsets into the dynamic shared array; negative: offset into the env's global scratch slice)
// per-body / per-dof working arrays: always LDS
#define MJH_LDS_SMALL(X)                                                                           \
  X(qpos) X(qvel) X(qvref) X(ws) X(qacc) X(smooth) X(asmooth) X(passive) X(bias) X(applied)        \
  X(tmpv) X(tmpv2) X(xpos) X(xquat) X(xmat) X(xipos) X(ximat) X(com) X(cinert) X(crb) X(cvel)      \
  X(cacc) X(cfrc) X(cfrcsub) X(xanchor) X(xaxis) X(cdof) X(cdofdot) X(qM) X(qLD) X(qLDinv)          \
  X(gpos) X(gmat) X(zero) X(dofpar) X(dofMadr) X(anc) X(p_gsize) X(p_rbound) X(p_mass) X(p_inertia) X(site) X(fext)
// contact / block / Jacobian pools: LDS, or (many-body layout, NROW = 8 kernels) the env's slice of global memory
#define MJH_LDS_POOLS(X) X(con) X(blki) X(blkf) X(blkq) X(bv) X(phi) X(sched) X(order) X(J) X(B) X(ext)
#define MJH_LDS_ARRAYS(X) MJH_LDS_SMALL(X) MJH_LDS_POOLS(X)

struct Lay {
#define X(n) int n;
  MJH_LDS_ARRAYS(X)
#undef X
  int total;  // floats
  // many-body layout, three-launch step: float offsets into the env's global scratch slice of the hand-over vectors
  // (initial acceleration, 1/M_dd, velocity after the controller, smooth force, solved acceleration) and of 8 ints of
  // meta data (nblk, nfixblk, nefc, ncon, flags, solver iterations)
  int g_a0, g_minv, g_qvel, g_smooth, g_qacc, g_meta, g_qM;
  int g_qLD, g_anc;   // dense solver: the factor L of M (nM floats) and the ancestor lists (nM ints), handed over by the assemble launch
  int g_dense;   // dense solver: AR' [cap x cap] | B rows [cap x nvs] | J rows [cap x nvs] | f, lo, hi, AR_qq, t, row map [6 x cap]
};

struct DConst { DModel M; Lay L; };

// kernel phases
enum { PH_STEP1 = 1, PH_INV = 2, PH_STEP2 = 4, PH_NOINT = 8, PH_FKONLY = 16, PH_MULM = 32, PH_RESET = 64,
       // many-body layout only: the fused step as three launches  assemble (PH_PRE) -> mjh_solve_kernel -> integrate (PH_POST)
       PH_PRE = 128, PH_POST = 256 };
// export flags
enum { XF_BODY = 1, XF_GEOM = 2, XF_CON = 4, XF_FORCE = 8, XF_PROF = 16,
       XF_NOSTORE = 32,     // read-only launch: nothing of the env's state, statistics or time is written back
       XF_DENSE = 64,
       XF_SPLIT2 = 256,     // window kernel behind a step2-only launch (split API): what mj_checkAcc's reset clears besides the state (qfrc_applied) is cleared as the fused kernel's step2 does
       XF_DEFER = 128 };    // window chain, assemble launch of the split API (mjh_step1 [+ mjh_inverse]): EVERY env is handed over to the window kernel (unconstrained ones too: nothing is integrated in this launch); qpos / qvel / qvel_ref / qfrc_applied are stored as mj_step1 leaves them     // assemble launch of a cohort whose solve runs the dense row-space solver (dense_pgs.h): no per-block solver matrices

#define CON_STRIDE 16  // dist, pos3, frame9, [13] geom1 | geom2 << 12 | dim << 24, [14] includemargin, [15] pad
#define CON_GEOMS 13
#define CON_MARGIN 14
#define CON_G1(c) (__float_as_int((c)[CON_GEOMS]) & 0xfff)
#define CON_G2(c) ((__float_as_int((c)[CON_GEOMS]) >> 12) & 0xfff)
#define CON_DIM(c) (__float_as_int((c)[CON_GEOMS]) >> 24)
// constraint blocks (HISTORY.md §5): header int4 + 16 parameter floats (s_blkf) + 16 solver-matrix floats (s_blkq) per block
//   hd.x = kind | nrows<<4 | nbase<<8 | clamp<<12 | jadr<<16 ; hd.y = id | rtype<<24 | side<<28 ; hd.z = a1 | n1<<16 ; hd.w = a2 | n2<<16
//   floats: [0..3] R, frictionloss, KI, Bc ; [4..7] aref per BASE row (row r = n +- k has aref_n +- aref_k) ;
//           [8..13] force per row ; [14..15] lo, hi ; s_blkq[0..15]: A_c = J_base M^-1 J_base^T (upper triangle), converted in
//           place into the row-space solver data Q (step_kernel.h: pgs_rows) when the solver starts ; condim-4 models add 12 floats per block in s_ext
#define BLKI_STRIDE 4
#define BLKF_STRIDE 16
#define BLKQ_STRIDE 16
#define BF_AREF 4
#define BF_F 8
#define BF_LO 14   // [14],[15]: projection interval lo, hi of the block's rows
enum { BK_SINGLE = 0, BK_PYR3 = 3, BK_PYR4 = 4 };
enum { RT_EQ = 0, RT_FL = 1, RT_LIMIT = 2, RT_CONTACT = 3, RT_WELD = 4 };   // RT_WELD: one row of a connect / weld equality, id = eq | row << 16
