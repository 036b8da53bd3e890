/* mjhip.h — C ABI of the MI355X-native many-environment rigid-body stepper.
 *
 * This is the drop-in boundary for the hot path of HoangGiang93/mujoco_sim: the
 * reference has no plugin interface; its boundary IS the handful of MuJoCo C
 * calls made by the step driver plus the raw mjModel/mjData fields the wrapper
 * dereferences.  Every entry point below names the reference
 * call site (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - plain C linkage, plain pointers and sizes, no exceptions across the ABI
 *   - return 0 on success, negative mjh_error on failure; text in mjh_last_error()
 *   - host I/O is double (mjtNum / ros_control handles are double,
 *     include/mujoco_sim/mj_hw_interface.h:58-65); device arithmetic is fp32
 *   - one stepping thread per engine (the reference serialises on `mtx`,
 *     src/mj_main.cpp:82,112)
 *   - "env" = one independent copy of the simulated world (the reference has
 *     exactly one: the globals `m`,`d`, include/mujoco_sim/mj_model.h:29-30)
 */
#ifndef MJHIP_H_
#define MJHIP_H_

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ enums */

enum mjh_error {
  MJH_OK = 0,
  MJH_ERR_ARG = -1,        /* bad argument / index out of range            */
  MJH_ERR_NO_DEVICE = -2,  /* no HIP device / HIP runtime failure          */
  MJH_ERR_CAPACITY = -3,   /* model exceeds what the kernels support       */
  MJH_ERR_STATE = -4,      /* call sequence violation (step2 before step1) */
  MJH_ERR_UNSUPPORTED = -5 /* feature not implemented                      */
};

/* joint types: values follow MuJoCo's mjtJoint so that jnt_type tables read by
 * the wrapper (mj_sim.cpp:503-513, mj_ros.cpp:2164-2194) keep their meaning */
enum mjh_joint { MJH_JNT_FREE = 0, MJH_JNT_BALL = 1, MJH_JNT_SLIDE = 2, MJH_JNT_HINGE = 3 };

/* geom types: values follow mjtGeom (read by mj_ros.cpp:1968-2094) */
enum mjh_geom {
  MJH_GEOM_PLANE = 0, MJH_GEOM_HFIELD = 1, MJH_GEOM_SPHERE = 2, MJH_GEOM_CAPSULE = 3,
  MJH_GEOM_ELLIPSOID = 4, MJH_GEOM_CYLINDER = 5, MJH_GEOM_BOX = 6, MJH_GEOM_MESH = 7
};

enum mjh_eq { MJH_EQ_CONNECT = 0, MJH_EQ_WELD = 1, MJH_EQ_JOINT = 2 };
/* sensor types: values follow mjtSensor; the reference publishes exactly these two (MjSim::init_sensors, mj_sim.cpp:973-1014) */
enum mjh_sensor { MJH_SENS_FORCE = 4, MJH_SENS_TORQUE = 5 };

/* constraint row types (order of assembly: equality, friction loss, limit, contact) */
enum mjh_cnstr {
  MJH_CNSTR_EQUALITY = 0, MJH_CNSTR_FRICTION_DOF = 1, MJH_CNSTR_LIMIT_JOINT = 3,
  MJH_CNSTR_CONTACT_FRICTIONLESS = 5, MJH_CNSTR_CONTACT_PYRAMIDAL = 6
};

/* disable-flag bits (subset of mjtDisableBit that this path honours) */
enum mjh_disable {
  MJH_DSBL_CONSTRAINT = 1 << 0, MJH_DSBL_EQUALITY = 1 << 1, MJH_DSBL_FRICTIONLOSS = 1 << 2,
  MJH_DSBL_LIMIT = 1 << 3, MJH_DSBL_CONTACT = 1 << 4, MJH_DSBL_PASSIVE = 1 << 5,
  MJH_DSBL_GRAVITY = 1 << 6, MJH_DSBL_WARMSTART = 1 << 8, MJH_DSBL_FILTERPARENT = 1 << 9,
  MJH_DSBL_REFSAFE = 1 << 11, MJH_DSBL_EULERDAMP = 1 << 13
};

/* per-env parameter tables that may override the shared model (S24 draws box
 * half-extents per env) */
enum mjh_env_param {
  MJH_EP_GEOM_SIZE = 0,       /* 3*ngeom  */
  MJH_EP_GEOM_RBOUND = 1,     /* ngeom    */
  MJH_EP_BODY_MASS = 2,       /* nbody    */
  MJH_EP_BODY_INERTIA = 3,    /* 3*nbody  */
  MJH_EP_BODY_INVWEIGHT0 = 4, /* 2*nbody  */
  MJH_EP_DOF_INVWEIGHT0 = 5,  /* nv       */
  MJH_EP_COUNT = 6
};

/* ------------------------------------------------------------ model (IR) */

typedef struct mjh_option {
  double timestep;          /* model/world/empty.xml:2 -> 0.005                */
  double gravity[3];
  int iterations;           /* main solver sweeps (MuJoCo default 100)         */
  double tolerance;         /* scaled-improvement threshold (default 1e-8)     */
  double impratio;          /* default 1                                       */
  int noslip_iterations;    /* model/ontology/scene.xml:2-3; 0 = off (default)  */
  int disableflags;         /* mjh_disable bits                                */
  double noslip_tolerance;  /* model/ontology/scene.xml:3 (MuJoCo default 1e-6) */
} mjh_option;

/* Compiled, topology-sorted model: the subset of mjModel this path reads
 * (SURVEY.md §8-b B1 lists the fields the wrapper itself touches).  All arrays
 * are owned by the model object; bodies are sorted parents-first, body 0 is
 * the world. */
typedef struct mjh_model {
  /* sizes */
  int nq, nv, nbody, njnt, ngeom, neq, npair, nM, ntree, nexclude;
  int maxcon;  /* contact capacity per env (contacts beyond it are dropped + flagged) */
  int maxefc;  /* constraint-row capacity per env                                     */
  int nmesh, nmeshvert; /* convex mesh assets: count, total vertices                    */
  mjh_option opt;
  double meaninertia; /* stat.meaninertia: mean diag(M(qpos0)) — scales solver tolerance */

  /* bodies */
  int *body_parentid, *body_rootid, *body_weldid, *body_jntadr, *body_jntnum;
  int *body_dofadr, *body_dofnum, *body_treeid, *body_level, *body_geomadr, *body_geomnum;
  double *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_inertia;
  double *body_gravcomp, *body_invweight0; /* [2*nbody] translational, rotational */
  /* joints */
  int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  double *jnt_pos, *jnt_axis, *jnt_stiffness, *jnt_range, *jnt_margin;
  double *jnt_solref, *jnt_solimp; /* [2*njnt], [5*njnt] limit parameters */
  double *qpos0, *qpos_spring;
  /* dofs */
  int *dof_bodyid, *dof_jntid, *dof_parentid, *dof_Madr, *dof_treeid;
  double *dof_armature, *dof_damping, *dof_frictionloss, *dof_invweight0;
  double *dof_solref, *dof_solimp; /* friction-loss parameters */
  /* trees (a tree = a child of the world with all its descendants; contiguous dofs) */
  int *tree_dofadr, *tree_dofnum, *tree_bodyid;
  /* geoms */
  int *geom_type, *geom_bodyid, *geom_condim, *geom_contype, *geom_conaffinity, *geom_priority;
  double *geom_pos, *geom_quat, *geom_size, *geom_rbound, *geom_friction;
  double *geom_solmix, *geom_solref, *geom_solimp, *geom_margin, *geom_gap;
  /* static candidate pair list: geom1 < geom2 by (type, then id) ordering rule of
   * the narrow phase; filters (same/weld body, parent-child, contype/conaffinity,
   * <exclude>) applied at compile time */
  int *pair_geom1, *pair_geom2;
  /* equality constraints */
  int *eq_type, *eq_obj1id, *eq_obj2id, *eq_active;
  double *eq_data, *eq_solref, *eq_solimp; /* [11*neq], [2*neq], [5*neq] */
  /* convex mesh assets (mjModel mesh_vert / geom_dataid; pr2.xml:5-22): the support-relevant vertices of each
   * mesh in the mesh geom's own frame (origin = centre of mass, axes = principal axes) */
  int *geom_dataid;                /* [ngeom] mesh id of a mesh geom, -1 otherwise */
  int *mesh_vertadr, *mesh_vertnum; /* [nmesh] */
  double *mesh_vert;                /* [3*nmeshvert] */
  /* name tables: resolved ONCE on the host (the reference calls mj_name2id per
   * joint per step: mj_hw_interface.cpp:64,79; mj_sim.cpp:1060,1083-1146) */
  char **body_names, **jnt_names, **geom_names;
  /* ---- appended in round 2 (older fields keep their offsets) ----
   * sites and force / torque sensors (m->nsensor, sensor_type, sensor_objid, sensor_adr, nsensordata: mj_sim.cpp:973-1014,
   * mj_ros.cpp:1933-1966); a sensor measures the interaction force (torque) between its site's body and that body's parent,
   * in the site frame, 3 numbers each */
  int nsite, nsensor, nsensordata, nmocap;
  int* site_bodyid; double *site_pos, *site_quat;      /* [nsite], [3*nsite], [4*nsite]: site frame in its body */
  int *sensor_type, *sensor_objid, *sensor_adr;        /* [nsensor]: mjh_sensor, site id, first index in sensordata */
  char **site_names, **sensor_names;
  /* mocap bodies (the `_ref` bodies MjSim::init_references welds robot links to, mj_sim.cpp:847-960): static children of the
   * world whose pose is an INPUT per environment (mjh_set_mocap_pose); body_mocapid[b] = index or -1; their initial pose is
   * body_pos / body_quat.  eq_type CONNECT / WELD: eq_obj1id / eq_obj2id are BODY ids (0 = world) and eq_data holds
   *   connect: [0..2] anchor in body1's frame, [3..5] the same point in body2's frame at qpos0
   *   weld:    [0..2] anchor in body2's frame, [3..5] the same point in body1's frame at qpos0,
   *            [6..9] relative orientation quat(body2)^-1 * quat(body1) at qpos0, [10] torquescale */
  int* body_mocapid;
} mjh_model;

/* ------------------------------------------------- model builder (host) */
/* Replaces the model-ingest boundary mj_loadXML (include/mujoco_sim/mj_util.h:190)
 * for programmatic scenes; an MJCF-subset loader is SURVEY.md §8-f F1. */
typedef struct mjh_builder mjh_builder;

mjh_builder* mjh_builder_create(void);
void mjh_builder_destroy(mjh_builder*);
void mjh_builder_set_option(mjh_builder*, const mjh_option*);
void mjh_builder_get_option(const mjh_builder*, mjh_option*);
void mjh_builder_set_capacity(mjh_builder*, int maxcon, int maxefc);
/* <compiler boundmass boundinertia>: lower bounds on the mass / principal inertias of every body except the world */
void mjh_builder_set_bounds(mjh_builder*, double boundmass, double boundinertia);
void mjh_builder_set_balanceinertia(mjh_builder*, int on);   /* <compiler balanceinertia> (mujoco_compile.cpp:157-160) */
/* returns body id (>0) ; parent 0 = world.  mass<=0 -> inertia inferred from geoms (density 1000) */
int mjh_builder_add_body(mjh_builder*, const char* name, int parent, const double pos[3],
                         const double quat[4], double gravcomp);
int mjh_builder_set_inertial(mjh_builder*, int body, double mass, const double ipos[3],
                             const double iquat[4], const double diaginertia[3]);
/* range==NULL -> unlimited.  Sentinels of add_geom / add_mesh_geom: friction NULL, condim / contype / conaffinity < 0 and
 * density < 0 mean "MuJoCo's default" (1 0.005 0.0001, 3, 1, 1, 1000); an explicit density 0 is a massless geom */
int mjh_builder_add_joint(mjh_builder*, const char* name, int body, int type, const double pos[3],
                          const double axis[3], const double range[2], double damping,
                          double stiffness, double armature, double frictionloss, double ref);
int mjh_builder_add_geom(mjh_builder*, const char* name, int body, int type, const double size[3],
                         const double pos[3], const double quat[4], const double friction[3],
                         int condim, int contype, int conaffinity, double density);
/* convex mesh asset (the <asset><mesh> of the reference's robots, pr2.xml:5-22): a vertex cloud, optionally with
 * triangles (volume, centre of mass and principal axes from the faces as mj_loadXML derives them; from the
 * bounding box otherwise).  Collides as its convex hull, like MuJoCo: the builder keeps the vertices that are
 * extreme along a dense set of directions.  scale may be NULL.  Returns the mesh id (>= 0) or a negative code. */
int mjh_builder_add_mesh(mjh_builder*, const double* vert, int nvert, const int* face, int nface, const double scale[3]);
int mjh_builder_add_mesh_stl(mjh_builder*, const char* path, const double scale[3]);   /* binary or ASCII STL, Wavefront OBJ (by extension) */
/* mesh geom: pos/quat place the MESH FILE's frame in the body, as <geom type="mesh" pos quat> does */
int mjh_builder_add_mesh_geom(mjh_builder*, const char* name, int body, int mesh, const double pos[3], const double quat[4],
                              const double friction[3], int condim, int contype, int conaffinity, double density);
int mjh_builder_add_exclude(mjh_builder*, int body1, int body2);
int mjh_builder_add_eq_joint(mjh_builder*, int joint1, int joint2, const double polycoef[5]);
/* <equality><connect body1 body2 anchor> / <weld body1 body2 anchor torquescale> (body2 = 0: the world).  connect: `anchor`
 * in body1's frame; weld: in body2's frame (NULL = its origin), the relative pose is the one at qpos0.  MjSim::init_references
 * emits <weld body1="X" body2="X_ref" torquescale="0.9"/> against a mocap clone of X (mj_sim.cpp:933-938). */
int mjh_builder_add_eq_connect(mjh_builder*, int body1, int body2, const double anchor[3]);
int mjh_builder_add_eq_weld(mjh_builder*, int body1, int body2, const double anchor[3], double torquescale);
/* <body mocap="true">: a child of the world without joints whose pose is set per environment at run time */
int mjh_builder_set_mocap(mjh_builder*, int body);
/* <site> and <sensor><force site=.../><torque site=.../> */
int mjh_builder_add_site(mjh_builder*, const char* name, int body, const double pos[3], const double quat[4]);
int mjh_builder_add_sensor(mjh_builder*, const char* name, int type /* mjh_sensor */, int site);
/* compile: derives inertias, qpos0, invweight0, meaninertia, rbound, pair list */
mjh_model* mjh_builder_compile(mjh_builder*);
void mjh_model_destroy(mjh_model*);
/* Sub-wave packing for small models (nv of a few, a handful of bodies: the C1 / C3 / C5 scenes): `copies` independent
 * instances of every moving body tree in ONE model, sharing the static geometry (world / welded-to-world geoms) and
 * never colliding with each other.  An engine created from the result steps `copies` environments per wavefront:
 * row w of every state array holds the qpos / qvel / ... of environments w*copies .. w*copies+copies-1 back to back
 * (copy c at offset c*nq, c*nv).  What the instances of one wavefront share: the solver's termination test (sweeps
 * continue until the slowest instance has converged), `time`, the statistics row and the bad-state reset.  Contact
 * and row capacities scale with `copies`.  Returns a new model (mjh_model_destroy) or NULL. */
mjh_model* mjh_model_replicate(const mjh_model* m, int copies);
int mjh_name2id(const mjh_model*, int objtype /*0 body,1 joint,2 geom,3 site,4 sensor*/, const char* name);
const char* mjh_id2name(const mjh_model*, int objtype, int id);

/* MJCF-subset loader: replaces load_XML -> mj_loadXML (include/mujoco_sim/mj_util.h:185-193) for the element
 * subset the reference models use on the step path (see csrc/mjcf_loader.cpp).  Returns NULL + mjh_last_error()
 * on failure; mjh_load_note() lists what was skipped (mesh geoms, <include>, default classes, ...). */
mjh_model* mjh_load_mjcf_string(const char* xml);
mjh_model* mjh_load_mjcf_file(const char* path);
/* one model from several files: the reference composes a world file and robot files into one MJCF before mj_loadXML
 * (MjSim::init, mj_sim.cpp:573-710).  <option> comes from the first file; names must be unique across files. */
mjh_model* mjh_load_mjcf_files(const char* const* paths, int n);
const char* mjh_load_note(void);
/* The loader options of ONE call (what MjSim::init_tmp reads from rosparams and writes into the files before mj_loadXML): the
 * mjh_load_set_* functions below keep per-THREAD settings for all later loads; this variant takes them as an argument and
 * leaves the thread's settings untouched. */
typedef struct mjh_load_options {
  double boundmass, boundinertia;   /* floor for <compiler boundmass boundinertia>: the reference writes 1e-6 / 1e-6 (mj_sim.cpp:584-590) */
  int robot_gravcomp;               /* ~disable_gravity: -1 keep the files' values, 0 / 1 written on every robot body (mj_sim.cpp:301-310) */
  int load_meshes;                  /* 1: mesh assets are read and collide as hulls; 0: mesh geoms are dropped (and noted) */
  unsigned odom_joints;             /* ~add_odom_joints mask: bits 0..5 = lin x y z, ang x y z (mj_sim.cpp:337-415) */
  int nrobot_pose;                  /* ~pose_init entries (mj_sim.cpp:312-335): root body names and x y z roll pitch yaw each */
  const char* const* robot_pose_body; const double* robot_pose;
  int parent_child_exclude;         /* disable_parent_child_collision_level (mujoco_sim.launch:7; mujoco_compile.cpp:250-290): see mjh_load_set_parent_child_exclude; 0 adds nothing */
} mjh_load_options;
void mjh_load_default_options(mjh_load_options*);
mjh_model* mjh_load_mjcf_files_opt(const char* const* paths, int n, const mjh_load_options* options);
/* 1 (default): <asset><mesh> files are read and mesh geoms collide as convex hulls; 0: mesh geoms are dropped (and
 * listed in mjh_load_note), which leaves the primitive collision geometry only */
void mjh_load_set_mesh_mode(int mode);
/* rosparam ~disable_gravity of the reference (robot.yaml:19, default true): MjSim::init_tmp rewrites gravcomp on EVERY
 * body of a robot file — 1 if set, 0 otherwise (mj_sim.cpp:301-310).  mode 1 / 0 does the same to the bodies of the files
 * after the first one in mjh_load_mjcf_files; -1 (default) keeps what the files say. */
void mjh_load_set_robot_gravcomp(int mode);
/* rosparam ~add_odom_joints (robot.yaml; mj_sim.cpp:337-415): the root body of every robot file gets the joints
 * "<root>_lin_odom_{x,y,z}_joint" (slide) / "<root>_ang_odom_{x,y,z}_joint" (hinge) selected by mask bits 0..5
 * (lin x y z, ang x y z), with the reference's rule that a planar linear axis comes along when the other one and the yaw
 * (or, for z, the pitch) are selected.  Resolve their dofs with mjh_name2id and pass them to mjh_set_odom_dofs. */
void mjh_load_set_odom_joints(unsigned mask);
/* launch argument disable_parent_child_collision_level of the reference (mujoco_sim.launch:7, default 1): its mujoco_compile writes
 * <exclude> pairs between every body and its first `level` ancestors into the compiled robot file (mujoco_compile.cpp:250-290).
 * Per-thread setting for all later loads; 0 (default) adds nothing (adjacent bodies are filtered by the model compiler anyway,
 * as MuJoCo's filterparent does).  Applied to robot files only — the files after the first of mjh_load_mjcf_files, or the one file of a
 * single-file load —, to the bodies that file adds; the walk ends at the file's top level (the reference names the robot's geom-less
 * wrapper body there, mujoco_compile.cpp:272-276: no collision is excluded by that pair). */
void mjh_load_set_parent_child_exclude(int level);
/* rosparam ~pose_init (mj_sim.cpp:312-335): position and roll / pitch / yaw (radians, tf2 setRPY) written onto the root body
 * of a robot file, by body name; pose NULL removes the entry, root_body NULL removes all */
void mjh_load_set_robot_pose(const char* root_body, const double pose[6]);
/* per-thread floor for <compiler boundmass boundinertia> of every file loaded afterwards: the reference writes
 * 1e-6 / 1e-6 into each file before mj_loadXML (mj_sim.cpp:584-590) */
void mjh_load_set_bounds(double boundmass, double boundinertia);

/* ------------------------------------------------------ scene builders */
/* SURVEY.md §8-d configs, as programmatic models.  `seed_base+env` seeds PCG32. */
mjh_model* mjh_scene_s24(void); /* 4 free boxes in a walled pen on the empty.xml floor  */
mjh_model* mjh_scene_s24_pen(double pen_half, int maxcon); /* the same scene with another pen (S24 itself: 0.175, 40); bench.py `s24d`: a narrow pen, ~30 contacts */
/* per-env S24 randomisation: fills qpos0[nenv*nq] and the per-env parameter tables
 * (each out pointer may be NULL).  Box half-extents U[0.05,0.125]^3.             */
int mjh_scene_s24_randomize(const mjh_model*, int env0, int nenv, unsigned seed_base,
                            double* qpos, double* geom_size, double* geom_rbound,
                            double* body_mass, double* body_inertia,
                            double* body_invweight0, double* dof_invweight0);
/* per-env randomisation of any scene of free boxes (C2): half-extents U[0.05,0.125]^3 with the mass /
 * inertia / inverse weights that follow, lattice position = the model's qpos0 + U[-jitter,jitter]^3, uniformly random
 * orientation; same table layout and seeding as mjh_scene_s24_randomize */
int mjh_scene_boxes_randomize(const mjh_model*, int env0, int nenv, unsigned seed_base, double jitter,
                              double* qpos, double* geom_size, double* geom_rbound,
                              double* body_mass, double* body_inertia,
                              double* body_invweight0, double* dof_invweight0);
mjh_model* mjh_scene_pendulum(void);          /* model/test/pendulum.xml restated (C1/C5) */
mjh_model* mjh_scene_arm7(int gravcomp);      /* 7-hinge Panda-like chain with limits (C3) */
mjh_model* mjh_scene_boxpile(int nbox);       /* nbox free boxes over the floor (C2 family) */

/* --------------------------------------------------------------- engine */
typedef struct mjh_engine mjh_engine;

/* Create an engine for `nenv` environments on HIP device `device`.
 * Replaces mj_makeData (mj_sim.cpp:816,835; mj_ros.cpp:571) + init_malloc
 * (mj_sim.cpp:563-571: ddq/dq/tau buffers).  `stream` is a hipStream_t (may be
 * NULL for the default stream); all kernels are launched on it. */
int mjh_create(const mjh_model* model, int nenv, int device, void* stream, mjh_engine** out);
void mjh_destroy(mjh_engine*); /* mj_deleteData/mj_deleteModel, mj_main.cpp:232-233 */

/* mj_step1 (mj_main.cpp:83): position + velocity stages, then the control
 * callback MjSim::controller (mj_sim.cpp:1055-1077) as a built-in device stage.
 * The launch itself is deferred to the next entry point: the reference calls
 * MjHWInterface::read() = mj_inverse right behind mj_step1 (mj_main.cpp:83-94), and
 * mjh_step1 + mjh_inverse then go out as ONE launch sharing the position and velocity
 * stages (the literal loop: 5.3 -> 5.9 M env-steps/s on S24); any other entry point
 * first issues the plain step1 launch, so the order of effects is the order of the
 * calls.  MJH_LAZY_STEP1=0 launches immediately. */
int mjh_step1(mjh_engine*);
/* mj_step2 (mj_main.cpp:108) then MjSim::set_odom_vels (mj_main.cpp:110). */
int mjh_step2(mjh_engine*);
/* n fused steps: step1 -> [inverse if with_inverse] -> step2, one launch per step
 * window; with_inverse reproduces the per-step mj_inverse of MjHWInterface::read
 * (mj_hw_interface.cpp:61). */
int mjh_step(mjh_engine*, int nsteps, int with_inverse);
/* mj_inverse (mj_hw_interface.cpp:61): fills qfrc_inverse. */
int mjh_inverse(mjh_engine*);
/* mj_forward (mj_ros.cpp:608,1421). */
int mjh_forward(mjh_engine*);
/* mj_mulM (mj_sim.cpp:1057): res = M(q)*vec for envs [env0, env0+n); host doubles [n*nv]. */
int mjh_mulM(mjh_engine*, int env0, int n, const double* vec, double* res);
int mjh_synchronize(mjh_engine*);

/* MjHWInterface::write (mj_hw_interface.cpp:73-91): ddq = effort command
 * (interpreted as desired acceleration), dq = velocity command; [n*nv] each,
 * either may be NULL.  Consumed (and zeroed, mj_sim.cpp:1075-1076) by the next step1.
 * The buffers are read before the call returns; a call of at most 4096 values goes through a
 * host-mapped staging ring and does not wait for the device (stream-ordered in front of the
 * range's next step launch), larger ones copy and wait. */
int mjh_set_cmd(mjh_engine*, int env0, int n, const double* ddq, const double* dq);
/* In-engine joint-space PD effort controller for ALL environments.  The reference closes this loop on the host for its one
 * environment: read() -> controller_manager->update() -> write() (mj_main.cpp:86-106) with ros_control effort controllers
 * (PID p 200 d 50: model/ontology/box/box.yaml:5-13) whose output MjSim::controller treats as a desired acceleration
 * (mj_sim.cpp:1057).  With thousands of environments the same law runs on the device in front of every step:
 * ddq[d] = kp (target[d] - qpos[d]) - kd qvel[d] on every hinge / slide dof (other dofs untouched), consumed by the controller
 * stage of mj_step1 exactly like a command written with mjh_set_cmd.  kp = kd = 0 switches it off.  Targets default to qpos0;
 * mjh_set_pd_target takes [n*nv] values indexed by dof. */
int mjh_set_pd_controller(mjh_engine*, double kp, double kd);
int mjh_set_pd_target(mjh_engine*, int env0, int n, const double* target);
/* which dofs are "controlled" (MjSim::controlled_joints, mj_sim.cpp:1058-1063): mask[nv] */
int mjh_set_controlled_dofs(mjh_engine*, const int* mask);
/* MjSim::set_odom_vels (mj_sim.cpp:1079-1153): dof ids of the 6 odom joints of one
 * robot (-1 = absent) and the commanded twist per env [n*6] */
int mjh_set_odom_dofs(mjh_engine*, const int lin_dof[3], const int ang_dof[3],
                      const int ang_qpos[3]);
int mjh_set_odom_vel(mjh_engine*, int env0, int n, const double* twist);

/* MjHWInterface::read (mj_hw_interface.cpp:62-70): waits for the steps queued on the range's
 * stream (one cohort's, if the range lies inside a cohort); up to 16384 values are packed by one
 * small kernel into host-mapped memory (one synchronisation), larger ranges take strided copies */
int mjh_get_joint_state(mjh_engine*, int env0, int n, double* qpos, double* qvel,
                        double* qfrc_inverse);
/* d->xpos / d->xquat readers (mj_ros.cpp:2100-2147) */
int mjh_get_body_state(mjh_engine*, int env0, int n, double* xpos, double* xquat);
/* d->geom_xpos / geom_xmat readers (mj_ros.cpp:1968-2094) */
int mjh_get_geom_state(mjh_engine*, int env0, int n, double* geom_xpos, double* geom_xmat);
/* full state: time, qpos, qvel, qacc_warmstart (add_old_state, mj_sim.cpp:465-558).  `time` is kept in fp64 on the device
 * (d->time is mjtNum: the ROS stamps, the 10 kHz controller gate and the real-time factor of mj_main.cpp:85,115-163 read
 * it): after n steps it equals n * opt.timestep to fp64 round-off, however long the simulation runs; only the packed
 * fp32 publish slice (mjh_export_state_device) rounds it */
int mjh_get_state(mjh_engine*, int env0, int n, double* time, double* qpos, double* qvel,
                  double* qacc_warmstart);
int mjh_set_state(mjh_engine*, int env0, int n, const double* time, const double* qpos,
                  const double* qvel, const double* qacc_warmstart);
/* (qacc and qacc_warmstart are one array on the device: after every solve qacc_warmstart = qacc, as mj_advance leaves them;
 * writing a warm start through mjh_set_state therefore also sets the qacc a following mjh_inverse reads)
 * other per-env vectors by name: "qacc","qfrc_bias","qfrc_applied","qfrc_passive",
 * "qfrc_constraint","qfrc_inverse","qacc_smooth","energy"(2) */
int mjh_get_field(mjh_engine*, const char* name, int env0, int n, double* out);
/* per-env solver statistics: ncon, nefc, solver iterations, flags (bit0 contact overflow,
 * bit1 row overflow, bit2 NaN reset) — int[n*4] */
int mjh_get_stats(mjh_engine*, int env0, int n, int* out);
/* contacts of ONE env (debug/parity): dist[maxcon], pos[3*maxcon], frame[9*maxcon], geom[2*maxcon]; returns ncon or <0.
 * Read-only: runs the position stage of that env into a scratch buffer; state, statistics, warm start and time are untouched
 * (it may be called between mjh_step1 and mjh_step2) */
int mjh_get_contacts(mjh_engine*, int env, double* dist, double* pos, double* frame, int* geom);

/* d->xfrc_applied (mj_sim.cpp:499 carries it across a model change): Cartesian force (3) and torque (3) per body, world
 * frame, applied at the body's centre of mass; [n * 6*nbody].  Enters qfrc_smooth of every step until changed. */
int mjh_set_xfrc_applied(mjh_engine*, int env0, int n, const double* xfrc);
int mjh_get_xfrc_applied(mjh_engine*, int env0, int n, double* xfrc);
/* d->sensordata (mj_ros.cpp:1933-1966 reads sensordata[sensor_adr[i] .. +3] of every force / torque sensor): [n * nsensordata],
 * computed at the end of mj_step2's forward part (mj_sensorAcc), i.e. the values of the LAST step */
int mjh_get_sensordata(mjh_engine*, int env0, int n, double* sensordata);
/* pose of mocap body `mocapid` in envs [env0, env0+n): pos[n*3], quat[n*4] (either may be NULL) — d->mocap_pos / mocap_quat,
 * what the reference's `~receive` mode drives (mj_sim.cpp:847-960) */
int mjh_set_mocap_pose(mjh_engine*, int env0, int n, int mocapid, const double* pos, const double* quat);
/* per-env model parameters (enum mjh_env_param) */
int mjh_set_env_param(mjh_engine*, int which, int env0, int n, const double* values);
/* MjRos::reset_robot (mj_ros.cpp:569-609): back to qpos0 (or the per-env initial
 * qpos set by mjh_set_initial_qpos), zero qvel/qacc/warmstart/time */
/* State transplant after a model change (reference: add_old_state(), mj_sim.cpp:465-558, run when spawn/destroy or a
 * robot reload recompiles the model): for every body of `from` whose NAME also exists in `to`, copy its joints' state
 * of envs [0, min(nenv)) — time, qvel, qacc_warmstart, qfrc_applied and qacc per dof if the dof counts match, qpos if the
 * joint counts match.  qpos_mode 0 is literal: the reference copies body_jntnum scalars of qpos (one per joint, so a free
 * body keeps only its x and otherwise takes the pose baked into the new model, mj_sim.cpp:613-623); qpos_mode 1 copies
 * every qpos scalar of the body's joints.  Returns the number of bodies matched, or a negative code. */
int mjh_transplant_state(mjh_engine* from, mjh_engine* to, int qpos_mode);
int mjh_set_initial_qpos(mjh_engine*, int env0, int n, const double* qpos);
int mjh_reset(mjh_engine*, const int* env_ids, int n);

/* spawn / destroy as per-env slot toggling (reference: spawn_objects / destroy_objects services,
 * mj_ros.cpp:859-1507, which re-compile the whole model): an inactive slot does not collide and is frozen.
 * Toggleable are the last 32 bodies of the model (all of them, except the world, in a model of up to 32 bodies):
 * declare the object pool after the robot. */
int mjh_set_slot_active(mjh_engine*, int env0, int n, int body, int active);
/* the services take LISTS (mujoco_msgs SpawnObject.objects[] / DestroyObject.names[], mj_ros.cpp:859-904,1430-1507): one call,
 * one upload and one small kernel for n (env, body) pairs — activate + pose (pos[n*3], quat[n*4] or NULL) + twist (vel[n*6] or
 * NULL: linear world, angular body frame, as qvel of a free joint) / deactivate */
int mjh_spawn_objects(mjh_engine*, int n, const int* env, const int* body, const double* pos, const double* quat, const double* vel);
int mjh_destroy_objects(mjh_engine*, int n, const int* env, const int* body);
/* initial pose / twist of a spawned free body (mj_ros.cpp:1406-1412) */
int mjh_set_body_pose(mjh_engine*, int env, int body, const double pos[3], const double quat[4], const double vel[6]);

/* zero-copy export for the single ROS state topic: packs time(1)+qpos(nq)+qvel(nv)
 * fp32 per env into a caller-provided DEVICE buffer [nenv*(1+nq+nv)] on the engine's
 * stream (feeds the RCCL all-gather). */
int mjh_export_state_device(mjh_engine*, void* d_out);
int mjh_state_stride(const mjh_engine*);

/* Pinned host mirror of envs [env0, env0+n) for the publisher threads: the reference's ROS layer
 * reads d->qpos / qvel / qfrc_inverse (joint states, mj_ros.cpp:2164-2194), d->xpos / xquat (tf, object states,
 * :2096-2149) and d->geom_xpos / geom_xmat (markers, :1968-2094) at each topic's own rate.  mjh_mirror_update() enqueues
 * an asynchronous refresh of the selected parts behind the steps queued so far and returns at once; mjh_mirror_wait()
 * blocks until the latest refresh has landed; mjh_mirror_field() returns the fp32 rows (env-major, `row_width` floats per
 * env) inside the pinned block. */
typedef struct mjh_mirror mjh_mirror;
enum { MJH_MIRROR_JOINTS = 1, MJH_MIRROR_BODIES = 2, MJH_MIRROR_GEOMS = 4 };
enum { MJH_MIRROR_TIME = 0, MJH_MIRROR_QPOS = 1, MJH_MIRROR_QVEL = 2, MJH_MIRROR_QFRC_INVERSE = 3, MJH_MIRROR_XPOS = 4,
       MJH_MIRROR_XQUAT = 5, MJH_MIRROR_GEOM_XPOS = 6, MJH_MIRROR_GEOM_XMAT = 7 };
int mjh_mirror_create(mjh_engine*, int env0, int n, mjh_mirror** out);
void mjh_mirror_destroy(mjh_mirror*);
int mjh_mirror_update(mjh_mirror*, int what);
int mjh_mirror_wait(mjh_mirror*);
const float* mjh_mirror_field(const mjh_mirror*, int which, int* row_width);
/* the mirrored simulation time of the n environments in fp64 (field MJH_MIRROR_TIME holds the same doubles: row_width 2 floats) */
const double* mjh_mirror_time(const mjh_mirror*);

/* debug: mean shader-clock ticks from kernel start to each of the 16 stage boundaries of one fused step */
int mjh_debug_stage_cycles(mjh_engine*, int with_inverse, double* out16);
/* debug: raw stamps of one step launch, out[nenv*20] indexed by launch position: [0..15] shader-clock stage stamps,
 * [16],[17] 100 MHz wall clock at start/end, [18] HW_ID | XCC_ID<<32, [19] env id (tools/timeline.py) */
int mjh_debug_stage_raw(mjh_engine*, int with_inverse, long long* out);
/* debug: the solve launch of a free-body model's many-body chain (C2) timed on the pools of the last step — out_ms[0] as it is,
 * out_ms[1] with every wave reading the block operands of one of `slices` environments (an L2-resident working set): what the
 * operand stream from MALL / HBM costs the sweeps.  out_iter[2]: mean sweeps of the two runs. */
int mjh_debug_solve_probe(mjh_engine*, int slices, int reps, double* out_ms, double* out_iter);
/* debug: one step launch that returns at stage boundary `stage` (1..14) and stores nothing (tools/stage_valu.sh:
 * per-stage hardware-counter differences) */
int mjh_debug_stop_at(mjh_engine*, int stage, int with_inverse);

/* ---- launch scheduling (no reference counterpart: the reference steps one mjData on one CPU thread,
 * mj_main.cpp:82-112).  Environments are independent, so mjh_step() may split them into `n` cohorts
 * (contiguous env ranges, 1..8), each stepped on its own HIP stream: the low-occupancy tail of one cohort's
 * step kernel then overlaps the bulk of another's, across consecutive mjh_step calls too.  The split entry
 * points of the reference's loop (mjh_step1 / mjh_inverse / mjh_step2) are issued per cohort as well, and the
 * two calls that loop makes for ONE robot in between — mjh_get_joint_state and mjh_set_cmd (MjHWInterface::read /
 * write) — on an env range inside one cohort are ordered on that cohort's stream only: the host waits for that
 * cohort, the others keep stepping.  The caller's stream forks into the cohort streams inside these calls and is
 * joined again by the next call of any other entry point (and by a ranged call that spans cohorts), so results and
 * ordering seen through this API do not depend on `n` (tests/test_gpu_round3.py: bitwise).  Within a launch the
 * envs are dispatched longest-solver-job first (order rebuilt on the device every MJH_ORDER_EVERY-th step, default 8).
 * Default n: 1 below 1024 envs, else 2 for the fused step and 3 (from 1536 envs) for the three-launch step of the many-body layout
 * (MJH_COHORTS overrides it for engines created afterwards). */
/* m->opt.timestep of a running engine: simulate() doubles it while the simulation lags the wall clock by > 1 ms (up to
 * max_time_step) and halves it back otherwise (mj_main.cpp:150-163) */
int mjh_set_timestep(mjh_engine*, double dt);
double mjh_get_timestep(const mjh_engine*);

/* Memory layout of engines created afterwards (process-wide; no reference counterpart).  0 (default): per-env
 * contact / block / Jacobian pools in LDS while they are small (<= 24 KB per env), otherwise in a per-env slice of global
 * memory with the step issued as three launches (assemble, solve, integrate); 1: LDS whenever the working set fits one
 * CU's 160 KiB; 2: global pools whenever possible.  Results do not depend on the layout beyond fp32 rounding. */
void mjh_set_layout_policy(int policy);
/* Engines created afterwards (process-wide; default 1, environment MJH_WINDOW): mjh_step of a small free-body model (the models that take
 * the contact-patch sweep: every kinematic tree a single free body, nv <= 32) in the default row order runs as two launches — the step
 * kernel up to the constraint rows, then mjh_window_kernel (csrc/window_pgs.h): projected Gauss-Seidel over windows of 16 consecutive
 * rows, FOUR environments per wavefront with the rows in registers, followed by mj_Euler.  0: the one-launch fused step with the
 * contact-patch sweep (same order, same iterates up to fp32 rounding).  mjh_window_solver() reports what an engine runs. */
void mjh_set_window_solver(int on);
int mjh_window_solver(const mjh_engine*);
int mjh_set_cohorts(mjh_engine*, int n);
int mjh_get_cohorts(const mjh_engine*);
/* mjh_step(e, n) of an articulated model in the LDS-resident layout runs up to `n` steps per launch and cohort with the environment's
 * state resident in LDS between them (the in-kernel step loop; default 8, 1 = one launch per step; bitwise the same results).
 * mjh_get_steps_per_launch: what the engine's mjh_step uses — 1 for free-body models and the many-body chain, whose steps are
 * chains of launches (src/mj_main.cpp:83-108 is one step of ONE world; the batch steps n times before the host looks again). */
int mjh_set_steps_per_launch(mjh_engine*, int n);
int mjh_get_steps_per_launch(const mjh_engine*);
/* Launch chains as graphs.  Where a step is a chain of launches per cohort — the window chain of small free-body models (assemble ->
 * window kernel) and the many-body layout (assemble -> [dense build -> dense solve] -> solve -> integrate) — mjh_step captures the
 * chain once per cohort and variant (hipStreamBeginCapture on the cohort's stream) and queues every further cohort-step with ONE
 * hipGraphLaunch; the graph is captured again whenever something its kernels take by value changes.  Same kernels, same arguments, same
 * order: results are bitwise those of the separate launches.  mode (also MJH_CHAIN_GRAPH): 1 (default) the many-body chain only, 2 the
 * window chain as well (measured slower on S24: two plain launches stay the default there), 0 every launch on its own.
 * mjh_launches_per_step: queue entries the LAST mjh_step issued per cohort-step — 1 when it replayed a captured graph, else the chain's
 * kernel launches (2 window chain, 3 / 5 many-body block / dense chain, 1 fused kernel): plain launches are also what a step falls back to
 * on the NULL stream, on a stream that is already being captured and after a failed capture / instantiate (latched per cohort and variant).
 * Before the first step (and after mjh_set_cohorts): what the next step intends.  A retired graph exec is destroyed only after the
 * stream it was last launched on has drained. */
void mjh_set_chain_graph(int mode);
int mjh_launches_per_step(const mjh_engine*);
/* HIP-event timing of the step-kernel launches on the stream they run on: enable (on = 1: every launch, on = N > 1: every
 * N-th launch — the event pairs cost stream time of their own, visible in launch-bound configs), step, then read the mean
 * duration [ms] and the number of launches timed since the last read (bench.py's roofline leg). */
int mjh_set_launch_timing(mjh_engine*, int on);
int mjh_get_launch_timing(mjh_engine*, double* mean_ms, int* count);

/* ---- multi-GPU: the environments of ONE simulation sharded over the GPUs of a node, driven by ONE host thread.
 * The reference is a single C++ node with a single publisher set (src/mj_main.cpp:167-236, src/mujoco_sim/mj_ros.cpp:554-564).
 * Environments are independent, so a group is one engine + one stream per device over contiguous env ranges
 * [k*N/G + min(k, N%G), ...) (shares differ by at most one), model tables replicated and NO collective in the step; the only
 * exchange is mjh_group_publish(): every device packs its slice (time | qpos | qvel per env, fp32, mjh_export_state_device)
 * and one RCCL ncclAllGather over xGMI leaves the full env-ordered state on EVERY device — issued at the state topic's rate,
 * not per step.  RCCL is resolved at run time (dlopen); without it, or when a device is listed twice, the
 * gather uses peer copies.  Everything else (commands, getters, slots, mirrors) goes through the per-device engines with
 * LOCAL env ids: mjh_group_engine() / mjh_group_locate().  All calls are asynchronous like their mjh_* counterparts. */
typedef struct mjh_group mjh_group;
int mjh_group_create(const mjh_model* model, int nenv_total, const int* devices /* NULL: 0..ndev-1 */, int ndev, mjh_group** out);
void mjh_group_destroy(mjh_group*);
int mjh_group_ndev(const mjh_group*);
int mjh_group_nenv(const mjh_group*);
mjh_engine* mjh_group_engine(mjh_group*, int rank);
int mjh_group_env_range(const mjh_group*, int rank, int* env0, int* n);
int mjh_group_locate(const mjh_group*, int env, int* rank, int* local_env);
int mjh_group_step(mjh_group*, int nsteps, int with_inverse);   /* mjh_step on every device */
int mjh_group_step1(mjh_group*);                                /* mj_main.cpp:83 on every device */
int mjh_group_inverse(mjh_group*);
int mjh_group_step2(mjh_group*);                                /* mj_main.cpp:108 */
int mjh_group_reset(mjh_group*);
int mjh_group_synchronize(mjh_group*);
/* pack + all-gather; host_out (may be NULL) receives device 0's copy: [nenv_total * mjh_group_state_stride()] floats, env order */
int mjh_group_publish(mjh_group*, float* host_out);
/* the gathered state on device `rank`: valid once the publish's exchange — which runs on a communication stream of its own,
 * beside the steps — has finished: after mjh_group_wait_publish(rank, stream) in that stream's order, after
 * mjh_group_synchronize(), or after a publish with host_out */
const float* mjh_group_state_device(const mjh_group*, int rank);
int mjh_group_wait_publish(mjh_group*, int rank, void* stream /* NULL: the device's engine stream */);
/* ... and the consumer's release: `stream` has finished reading the gathered state of device `rank` up to this point of its order.
 * The NEXT publish overwrites that buffer; its exchange waits for the release.  A consumer that reads asynchronously without
 * releasing must synchronise before the next mjh_group_publish. */
int mjh_group_release_publish(mjh_group*, int rank, void* stream /* NULL: the device's engine stream */);
int mjh_group_state_stride(const mjh_group*);
int mjh_group_uses_rccl(const mjh_group*);
/* HIP events on device 0's COMMUNICATION stream around the exchange of every publish (the all-gather, or the peer copies): mean duration in
 * milliseconds and the number of publishes since the last call */
int mjh_group_set_publish_timing(mjh_group*, int on);
int mjh_group_get_publish_timing(mjh_group*, double* mean_ms, int* count);
void mjh_group_set_transport(int mode);   /* groups created afterwards: 0 = RCCL when available and the devices are distinct (default), 1 = peer copies,
                                             2 = RCCL also when a device is listed twice (a real RCCL refuses that at ncclCommInitAll and the group falls back to peer
                                             copies; the recording stand-in of tests/nccl_stub, named by MJH_RCCL_LIB, accepts it: N ranks on one device) */
/* Test hook, no device needed: the RCCL call sequence of `publishes` exchanges with `ndev` ranks exactly as mjh_group_publish issues it
 * (per_thread != 0: a host thread per rank enqueues its own ncclAllGather; 0: one grouped call inside ncclGroupStart / ncclGroupEnd), between
 * ncclCommInitAll and ncclCommDestroy, against the library MJH_RCCL_LIB names (default: the process's RCCL); buffers and streams are
 * made-up addresses that are only passed through. */
int mjh_debug_rccl_exchange(int ndev, int per_thread, unsigned long slot_floats, int publishes);
/* Host threads of a group (groups created afterwards; default 1, environment MJH_GROUP_THREADS): 1 = one persistent host thread per device
 * issues that device's launches, exports and its rank of the all-gather, every mjh_group_* call posts one job per device and waits for
 * them — the host time of a call is that of ONE device (the reference steps on one thread, mj_main.cpp:203: with eight devices behind
 * it that thread's launch calls alone would be as long as a light scene's step); 0 = the caller's thread issues everything, one device
 * after the other.  Same launches in the same per-device order either way: results are bitwise the same.  mjh_group_host_threads:
 * the number of such threads of a group (0: none). */
void mjh_group_set_host_threads(int on);
int mjh_group_host_threads(const mjh_group* g);

/* ROS-free harness of the host loop (csrc/host_sim.cpp: simulate() + MjhHWInterface, mirrors of
 * mj_main.cpp:76-164 and mj_hw_interface.cpp:59-110) with an in-process PD effort controller on
 * every hinge/slide joint of env `env`; returns final joint positions / efforts and the real-time factor */
int mjh_host_run_pd(mjh_engine*, int env, const double* target, double kp, double kd, long nsteps,
                    double* out_qpos, double* out_effort, double* out_rtf);

/* the same harness over a multi-GPU group: the ROS surface attached to GLOBAL env `env`, every shard stepped each step
 * (mjh_group_step1 / inverse / step2), the state slice of all environments all-gathered every `publish_every` steps; out_state
 * (may be NULL, [nenv_total * stride] floats) receives the last published slice */
int mjh_host_run_pd_group(mjh_group*, int env, const double* target, double kp, double kd, long nsteps, int publish_every,
                          double* out_qpos, double* out_effort, float* out_state);
/* simulate() with its real-time pacing switched ON (mj_main.cpp:115-163): the wall-clock spin that holds the real-time factor
 * at <= 1 and the adaptive timestep (doubled while the simulation lags the wall clock by > 1 ms, up to max_time_step; halved
 * back otherwise).  out_stats6 = {sim_time, wall_time, rtf, final_dt, steps, dt_changes} */
int mjh_host_run_realtime(mjh_engine*, int env, const double* target, double kp, double kd, long nsteps, double max_time_step,
                          double* out_stats6);

/* introspection */
int mjh_nenv(const mjh_engine*);
const mjh_model* mjh_engine_model(const mjh_engine*);
int mjh_lds_bytes(const mjh_engine*);  /* dynamic LDS per env (= per workgroup) */
/* Gauss-Seidel visiting order of the engine's solver sweeps: 2 = the constraint rows of efc_* in their own order, the order
 * mj_solPGS (engine_solver.c, reached through mj_step2, mj_main.cpp:108) walks — the DEFAULT; 1 / 0 = the legacy orders that regroup
 * the rows to expose more independent work (1: contact patches sorted by body pair, small free-body models; 0: independent pairs /
 * groups of constraint blocks).  Any order is a valid PGS iteration and they agree at convergence; stopped at the iteration cap they do
 * not (BASELINE.md section 3 has the size of that effect), which is why the reference's order is the default. */
int mjh_solver_order(const mjh_engine*);
/* How the engine walks that order: 1 = row order with a precedence-preserving list schedule (default) — a constraint block (or contact
 * patch) starts once every EARLIER block that shares a kinematic tree with it is done; blocks without a common tree touch disjoint
 * dofs, their updates commute exactly, so up to 2 / 4 / 16 of them run side by side in one wavefront and the iterates (and the sweep
 * count: the convergence test sums fixed-point integers) are bit-identical to the strictly sequential sweep; 2 = that sequential sweep
 * itself, one block after the other (the reference the tests compare 1 against); 0 = a legacy reordering schedule (mjh_solver_order
 * says which). */
int mjh_pgs_schedule(const mjh_engine*);
/* 1: the engine's sweeps take the contact-patch form (csrc/patch_pgs.h: small free-body models — up to 16 rows between the same two
 * bodies are one unit, up to four units side by side), 0: a block form (one constraint block = one equality / limit / friction-loss
 * row or the pyramid rows of one contact) */
int mjh_patch_sweep(const mjh_engine*);
/* Gauss-Seidel order / schedule of engines created afterwards (process-wide; no reference counterpart): 1 (default), 2 or 0 as
 * mjh_pgs_schedule reports them.  Articulated models whose sweeps are sequential in any case (more than 32 dofs and a kinematic tree
 * of more than 16) run row order under 0 as well. */
void mjh_set_pgs_row_order(int mode);
/* 1: mjh_step solves the environments of this engine whose rows fit with the dense row-space solver (AR = J M^-1 J^T on the matrix
 * cores, column sweeps: csrc/dense_pgs.h) — articulated models in the many-body layout; same rows, same visiting order, same results up
 * to fp32 rounding as the block solver (MJH_DENSE=0 turns it off) */
int mjh_dense_solver(const mjh_engine*);
int mjh_query_lds_bytes(const mjh_model*); /* same figure without a device: capacity planning (160 KiB per CU) */
int mjh_debug_lds_layout(const mjh_model*, char* out, int cap);   /* the LDS layout as text ("name offset" per array, float units; then totals): capacity planning, tools/lds_layout.py */
int mjh_query_lds_bytes_assemble(const mjh_model*);   /* LDS per env of the assemble-only launch of the window chain (small free-body models; 0: not taken) */
const char* mjh_last_error(void);
const char* mjh_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MJHIP_H_ */
