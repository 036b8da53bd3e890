// ThreadSanitizer harness for the host-side model code (no HIP): several threads compile models at the same time — MJCF strings
// with per-call loader options, the built-in scenes, replication, name look-ups — the way a multi-robot host process would
// (the reference guards its own loading with one mutex, mj_sim.cpp; this library keeps its loader state per thread).
// Built and run by tests/test_host_tsan.py:  g++ -fsanitize=thread  csrc/{model_builder,mjcf_loader,scenes}.cpp  this file.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "mjhip.h"

static const char* ARM = R"(<mujoco><option timestep="0.002"/><worldbody>
  <geom name="floor" type="plane" size="0 0 0.05"/>
  <body name="base" pos="0 0 0.5"><joint name="j1" type="hinge" axis="0 0 1" range="-1 1" limited="true"/>
    <geom type="capsule" size="0.04 0.2" pos="0.2 0 0" quat="0.707 0 0.707 0"/>
    <body name="fore" pos="0.4 0 0"><joint name="j2" type="hinge" axis="0 1 0" damping="0.1"/>
      <geom type="capsule" size="0.03 0.15" pos="0.15 0 0" quat="0.707 0 0.707 0"/></body></body>
  <body name="ball" pos="0.3 0 1.0"><freejoint/><geom type="sphere" size="0.05"/></body>
</worldbody></mujoco>)";

static std::atomic<int> failures{0};
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "thread %d: check failed: %s (%s)\n", id, #c, mjh_last_error()); failures++; return; } } while (0)

static void worker(int id) {
  for (int rep = 0; rep < 8; rep++) {
    mjh_model* a = mjh_load_mjcf_string(ARM);
    CHECK(a && a->nv == 8 && a->nbody == 4);
    CHECK(mjh_name2id(a, 1, "j2") == 1 && mjh_name2id(a, 0, "ball") == 3);
    mjh_load_options opt; mjh_load_default_options(&opt);
    opt.robot_gravcomp = id & 1; opt.load_meshes = 0;      // (per-call options live on the caller's stack, the loader's state per thread)
    (void)opt;
    mjh_model* s = mjh_scene_s24();
    CHECK(s && s->nv == 24);
    mjh_model* p = mjh_scene_pendulum();
    CHECK(p && p->nv == 9);
    mjh_model* p3 = mjh_model_replicate(p, 2 + (id % 3));
    CHECK(p3 && p3->nv == 9 * (2 + (id % 3)));
    // an error on this thread must not show up on another one
    mjh_model* bad = mjh_load_mjcf_string(id & 1 ? "<mujoco><worldbody><body><geom type=\"sphere\"/></body></worldbody>" : "not xml at all");
    CHECK(bad == nullptr && std::strlen(mjh_last_error()) > 0);
    mjh_model_destroy(p3); mjh_model_destroy(p); mjh_model_destroy(s); mjh_model_destroy(a);
  }
}

int main() {
  std::vector<std::thread> th;
  for (int i = 0; i < 6; i++) th.emplace_back(worker, i);
  for (auto& t : th) t.join();
  std::printf("host threads: %d failures\n", failures.load());
  return failures.load() ? 1 : 0;
}
