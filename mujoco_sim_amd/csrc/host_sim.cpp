// host_sim.cpp — see host_sim.h.  Host C++ only; every physics call goes through the C ABI.
#include "host_sim.h"

#include <chrono>
#include <cmath>
#include <cstring>
#include <deque>
#include <thread>

static const double kMinVal = 1e-15;  // mjMINVAL (mj_hw_interface.cpp:81, mj_sim.cpp:1069)

int MjhSim::sync_controlled() {
  std::vector<int> mask(model->nv > 0 ? model->nv : 1, 0);
  for (const std::string& j : controlled_joints) {
    const int id = mjh_name2id(model, 1, j.c_str());
    if (id < 0) continue;
    mask[model->jnt_dofadr[id]] = 1;  // hinge / slide joints: one dof (the reference indexes jnt_dofadr the same way)
  }
  if (group) {   // the controlled-joint set is a property of the model, not of one environment: every shard gets it
    for (int k = 0; k < mjh_group_ndev(group); k++) { const int rc = mjh_set_controlled_dofs(mjh_group_engine(group, k), mask.data()); if (rc) return rc; }
    return MJH_OK;
  }
  return mjh_set_controlled_dofs(engine, mask.data());
}

int MjhSim::attach_group(mjh_group* g, int global_env) {
  int rank = 0, local = 0;
  const int rc = mjh_group_locate(g, global_env, &rank, &local);
  if (rc) return rc;
  group = g; engine = mjh_group_engine(g, rank); env = local; model = mjh_engine_model(engine);
  published.assign((size_t)mjh_group_nenv(g) * mjh_group_state_stride(g), 0.0f);
  return MJH_OK;
}

int MjhSim::sync_odom(const std::string& robot) {
  static const char* lin[3] = {"_lin_odom_x_joint", "_lin_odom_y_joint", "_lin_odom_z_joint"};
  static const char* ang[3] = {"_ang_odom_x_joint", "_ang_odom_y_joint", "_ang_odom_z_joint"};
  int l[3], a[3], aq[3];
  for (int k = 0; k < 3; k++) {
    const int jl = mjh_name2id(model, 1, (robot + lin[k]).c_str()), ja = mjh_name2id(model, 1, (robot + ang[k]).c_str());
    l[k] = jl >= 0 ? model->jnt_dofadr[jl] : -1;
    a[k] = ja >= 0 ? model->jnt_dofadr[ja] : -1;
    aq[k] = ja >= 0 ? model->jnt_qposadr[ja] : -1;
  }
  if (group) {
    for (int k = 0; k < mjh_group_ndev(group); k++) { const int rc = mjh_set_odom_dofs(mjh_group_engine(group, k), l, a, aq); if (rc) return rc; }
    return MJH_OK;
  }
  return mjh_set_odom_dofs(engine, l, a, aq);
}

int MjhSim::push_odom_vels(const std::string& robot) {
  static const char* names[6] = {"_lin_odom_x_joint", "_lin_odom_y_joint", "_lin_odom_z_joint",
                                 "_ang_odom_x_joint", "_ang_odom_y_joint", "_ang_odom_z_joint"};
  double twist[6];
  for (int k = 0; k < 6; k++) { auto it = odom_vels.find(robot + names[k]); twist[k] = it == odom_vels.end() ? 0.0 : it->second; }
  return mjh_set_odom_vel(engine, env, 1, twist);
}

MjhHWInterface::MjhHWInterface(MjhSim* sim, const std::string& robot) : sim_(sim) {
  joint_names = sim->joint_names[robot];
  const size_t n = joint_names.size();
  joint_positions.assign(n, 0.0); joint_velocities.assign(n, 0.0); joint_efforts.assign(n, 0.0);
  joint_velocities_command.assign(n, 0.0); joint_efforts_command.assign(n, 0.0);
  for (const std::string& j : joint_names) {  // resolved once; the reference calls mj_name2id per joint per step
    const int id = mjh_name2id(sim->model, 1, j.c_str());
    qpos_id_.push_back(id >= 0 ? sim->model->jnt_qposadr[id] : -1);
    dof_id_.push_back(id >= 0 ? sim->model->jnt_dofadr[id] : -1);
  }
  qpos_.assign(sim->model->nq, 0.0); qvel_.assign(sim->model->nv, 0.0); qfrc_.assign(sim->model->nv, 0.0);
  ddq_.assign(sim->model->nv, 0.0); dq_.assign(sim->model->nv, 0.0);
}

void MjhHWInterface::read() {
  if (sim_->group) mjh_group_inverse(sim_->group); else mjh_inverse(sim_->engine);   // mj_inverse: every shard, then the attached env is read
  mjh_get_joint_state(sim_->engine, sim_->env, 1, qpos_.data(), qvel_.data(), qfrc_.data());
  for (size_t i = 0; i < joint_names.size(); i++) {
    if (dof_id_[i] < 0) continue;
    joint_positions[i] = qpos_[qpos_id_[i]];
    joint_velocities[i] = qvel_[dof_id_[i]];
    joint_efforts[i] = qfrc_[dof_id_[i]];
  }
}

void MjhHWInterface::write() {
  std::fill(ddq_.begin(), ddq_.end(), 0.0); std::fill(dq_.begin(), dq_.end(), 0.0);
  for (size_t i = 0; i < joint_names.size(); i++) {
    if (dof_id_[i] < 0 || sim_->controlled_joints.find(joint_names[i]) == sim_->controlled_joints.end()) continue;
    if (std::fabs(joint_velocities_command[i]) > kMinVal) dq_[dof_id_[i]] = joint_velocities_command[i];
    else ddq_[dof_id_[i]] = joint_efforts_command[i];
  }
  mjh_set_cmd(sim_->engine, sim_->env, 1, ddq_.data(), dq_.data());
}

void MjhHWInterface::doSwitch(const std::vector<std::string>& stopped) {
  for (const std::string& name : stopped)
    for (size_t i = 0; i < joint_names.size(); i++) if (joint_names[i] == name) joint_efforts_command[i] = 0.0;
}

SimulateStats simulate(MjhSim* sim, std::vector<MjhHWInterface*>& hw, const std::function<void(double, double)>& update,
                       long nsteps, bool real_time) {
  using clk = std::chrono::steady_clock;
  SimulateStats st;
  const double time_step = sim->model->opt.timestep;   // the configured step; dt is what the engine currently integrates with
  double dt = mjh_get_timestep(sim->engine) > 0 ? mjh_get_timestep(sim->engine) : time_step;
  double sim_time = 0, last_sim_time = 0;
  const auto t0 = clk::now();
  std::deque<double> win_sim, win_wall;
  const size_t num_step = (size_t)std::ceil(1.0 / time_step);
  for (long s = 0; s < nsteps; s++) {
    const double sim_period = sim_time - last_sim_time;
    if (sim->group) mjh_group_step1(sim->group); else mjh_step1(sim->engine);   // mj_main.cpp:83 (+ controller callback :49-52); every shard
    if (sim_period >= 1.0 / 10000.0 || s == 0) {             // :85 controller update at <= 10 kHz
      last_sim_time = sim_time;
      for (MjhHWInterface* h : hw) h->read();                // :93
      if (update) update(sim_time, sim_period);              // :99 controller_manager->update
    }
    for (MjhHWInterface* h : hw) h->write();                 // :105
    if (sim->group) mjh_group_step2(sim->group); else mjh_step2(sim->engine);   // :108, set_odom_vels :110 runs inside
    if (sim->group && sim->publish_every > 0 && (s + 1) % sim->publish_every == 0)   // state topic rate: pack + RCCL all-gather
      mjh_group_publish(sim->group, sim->published.data());
    sim_time += dt;
    const double wall = std::chrono::duration<double>(clk::now() - t0).count();
    if (real_time) {                                         // :127-131 spin until wall-clock >= sim-time
      while (std::chrono::duration<double>(clk::now() - t0).count() - sim_time < -1e-6) std::this_thread::yield();
    }
    win_sim.push_front(sim_time); win_wall.push_front(wall);  // :115-147 real-time factor over a sliding 1 s window
    if (win_sim.size() > num_step) { st.rtf = (sim_time - win_sim.back()) / (wall - win_wall.back() + 1e-12); win_sim.pop_back(); win_wall.pop_back(); }
    if (real_time) {                                         // :150-163 change the timestep when out of sync
      const double error_time = std::chrono::duration<double>(clk::now() - t0).count() - sim_time;
      double ndt = dt;
      if (error_time > 1e-3) { if (dt < sim->max_time_step) ndt = dt * 2; }
      else if (dt > time_step) ndt = dt / 2;
      if (ndt != dt) {
        dt = ndt; st.dt_changes++;
        if (sim->group) { for (int k = 0; k < mjh_group_ndev(sim->group); k++) mjh_set_timestep(mjh_group_engine(sim->group, k), dt); }
        else mjh_set_timestep(sim->engine, dt);
      }
    }
  }
  st.final_dt = dt;
  if (sim->group) mjh_group_synchronize(sim->group); else mjh_synchronize(sim->engine);
  st.sim_time = sim_time; st.steps = nsteps;
  st.wall_time = std::chrono::duration<double>(clk::now() - t0).count();
  if (st.rtf == 0 && st.wall_time > 0) st.rtf = sim_time / st.wall_time;
  return st;
}

// ---- C shim used by the tests: runs simulate() with an in-process PD "effort controller"
// (ros_control PID with i = 0: p 200, d 50 in model/ontology/box/box.yaml:5-13) on every joint of the model.
static int run_pd(MjhSim& sim, const double* target, double kp, double kd, long nsteps, bool real_time, double* out_qpos, double* out_effort, SimulateStats* out_st);
extern "C" int mjh_host_run_pd(mjh_engine* engine, int env, const double* target, double kp, double kd, long nsteps,
                               double* out_qpos, double* out_effort, double* out_rtf) {
  if (!engine) return MJH_ERR_ARG;
  MjhSim sim; sim.engine = engine; sim.model = mjh_engine_model(engine); sim.env = env;
  SimulateStats st;
  const int rc = run_pd(sim, target, kp, kd, nsteps, false, out_qpos, out_effort, &st);
  if (out_rtf) *out_rtf = st.rtf;
  return rc;
}
// the same loop over a multi-GPU group: the ROS surface attached to GLOBAL env `env`, every shard stepped, the state of all
// environments all-gathered every `publish_every` steps; out_state (optional) receives the last published slice
extern "C" int mjh_host_run_pd_group(mjh_group* group, int env, const double* target, double kp, double kd, long nsteps, int publish_every,
                                     double* out_qpos, double* out_effort, float* out_state) {
  if (!group) return MJH_ERR_ARG;
  MjhSim sim;
  int rc = sim.attach_group(group, env);
  if (rc) return rc;
  sim.publish_every = publish_every;
  SimulateStats st;
  rc = run_pd(sim, target, kp, kd, nsteps, false, out_qpos, out_effort, &st);
  if (!rc && out_state) std::memcpy(out_state, sim.published.data(), sim.published.size() * sizeof(float));
  return rc;
}
// real-time pacing of simulate() under test (mj_main.cpp:115-163): `max_time_step` bounds the adaptive timestep; returns the
// loop's statistics [sim_time, wall_time, rtf, final_dt, steps, dt_changes]
extern "C" int mjh_host_run_realtime(mjh_engine* engine, int env, const double* target, double kp, double kd, long nsteps, double max_time_step,
                                     double* out_stats6) {
  if (!engine) return MJH_ERR_ARG;
  MjhSim sim; sim.engine = engine; sim.model = mjh_engine_model(engine); sim.env = env; sim.max_time_step = max_time_step;
  SimulateStats st;
  const int rc = run_pd(sim, target, kp, kd, nsteps, true, nullptr, nullptr, &st);
  if (out_stats6) { out_stats6[0] = st.sim_time; out_stats6[1] = st.wall_time; out_stats6[2] = st.rtf; out_stats6[3] = st.final_dt; out_stats6[4] = (double)st.steps; out_stats6[5] = (double)st.dt_changes; }
  return rc;
}
static int run_pd(MjhSim& sim, const double* target, double kp, double kd, long nsteps, bool real_time, double* out_qpos, double* out_effort, SimulateStats* out_st) {
  const mjh_model* m = sim.model;
  std::vector<std::string> names;
  for (int j = 0; j < m->njnt; j++) if (m->jnt_type[j] == MJH_JNT_HINGE || m->jnt_type[j] == MJH_JNT_SLIDE) names.push_back(m->jnt_names[j]);
  sim.joint_names["robot"] = names; sim.robot_names.insert("robot");
  for (const std::string& n : names) sim.controlled_joints.insert(n);
  int rc = sim.sync_controlled();
  if (rc) return rc;
  MjhHWInterface hwi(&sim, "robot");
  std::vector<MjhHWInterface*> hw{&hwi};
  auto update = [&](double, double) {
    for (size_t i = 0; i < names.size(); i++)
      hwi.joint_efforts_command[i] = kp * (target[i] - hwi.joint_positions[i]) - kd * hwi.joint_velocities[i];
  };
  SimulateStats st = simulate(&sim, hw, update, nsteps, real_time);
  hwi.read();
  for (size_t i = 0; i < names.size(); i++) { if (out_qpos) out_qpos[i] = hwi.joint_positions[i]; if (out_effort) out_effort[i] = hwi.joint_efforts[i]; }
  if (out_st) *out_st = st;
  return MJH_OK;
}
