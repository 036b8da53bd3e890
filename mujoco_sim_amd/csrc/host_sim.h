// host_sim.h — ROS-free C++ mirror of the reference's host side of the hot path, written against the
// C ABI (include/mjhip.h).  Same names, argument meaning and call order as the reference so that the
// maintainer-side diff is mechanical:
//   MjhHWInterface::read / write / doSwitch   <- MjHWInterface (src/mujoco_sim/mj_hw_interface.cpp:59-110)
//   MjhSim::controlled_joints / odom_vels      <- MjSim statics (include/mujoco_sim/mj_sim.h:84-100)
//   simulate()                                 <- simulate() loop body (src/mj_main.cpp:76-164)
// ROS types (hardware_interface::RobotHW, controller_manager) are not available in this image; the
// controller_manager->update(time, period) call becomes a std::function hook with the same signature.
#pragma once
#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/mjhip.h"

struct MjhSim {
  // MjSim::joint_names[robot] (mj_sim.h:86): joints exported to ros_control, per robot
  std::map<std::string, std::vector<std::string>> joint_names;
  // MjSim::controlled_joints (mj_sim.h:88): filled from controller_manager/list_controllers (mj_ros.cpp:634-668)
  std::set<std::string> controlled_joints;
  // MjSim::odom_vels (mj_sim.h:90): "<robot>_lin_odom_x_joint" ... -> commanded velocity (mj_ros.cpp:193-206)
  std::map<std::string, double> odom_vels;
  std::set<std::string> robot_names;
  double max_time_step = 0.005;  // robot.yaml:56

  mjh_engine* engine = nullptr;
  const mjh_model* model = nullptr;
  int env = 0;  // which environment the single ROS surface is attached to (LOCAL id on `engine`)
  // multi-GPU (include/mjhip.h "multi-GPU"): when set, simulate() steps every device of the group; `engine` / `env` are the
  // shard and local id of the attached environment (attach_group), and every `publish_every` steps the state slice of ALL
  // environments is all-gathered (RCCL) into `published` — what the single state / clock publisher set reads (mj_ros.cpp:554-564)
  mjh_group* group = nullptr;
  int publish_every = 0;
  std::vector<float> published;
  int attach_group(mjh_group* g, int global_env);

  // push controlled_joints / odom joints to the engine (name -> id resolved ONCE, not per step)
  int sync_controlled();
  int sync_odom(const std::string& robot);
  int push_odom_vels(const std::string& robot);
};

class MjhHWInterface {
 public:
  MjhHWInterface(MjhSim* sim, const std::string& robot);
  void read();   // mj_inverse + gather qpos/qvel/qfrc_inverse per joint (mj_hw_interface.cpp:59-71)
  void write();  // velocity command (if |v| > mjMINVAL) else effort command, controlled joints only (:73-91)
  // doSwitch (:93-110): zero the effort command of every joint claimed by a stopped controller
  void doSwitch(const std::vector<std::string>& stopped_joint_names);

  std::vector<std::string> joint_names;
  std::vector<double> joint_positions, joint_velocities, joint_efforts;
  std::vector<double> joint_velocities_command, joint_efforts_command;

 private:
  MjhSim* sim_;
  std::vector<int> qpos_id_, dof_id_;
  std::vector<double> qpos_, qvel_, qfrc_, ddq_, dq_;
};

struct SimulateStats { double sim_time = 0, wall_time = 0, rtf = 0, final_dt = 0; long steps = 0, dt_changes = 0; };
// One pass of the reference loop per step: step1 -> read -> controller update -> write -> step2 (+odom).
// `update(sim_time, sim_period)` stands in for controller_manager->update (mj_main.cpp:99).
// real_time = true reproduces the wall-clock spin of mj_main.cpp:127-131.
SimulateStats simulate(MjhSim* sim, std::vector<MjhHWInterface*>& hw, const std::function<void(double, double)>& update,
                       long nsteps, bool real_time);
