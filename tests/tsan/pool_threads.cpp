// ThreadSanitizer driver of csrc/host_pool.h (the group host's thread per device): 8 workers, 20 000 fork-join rounds of jobs that write
// per-worker state the caller reads back, error propagation, and pools created and destroyed while idle and right after a job.
#include <cstdio>
#include <string>
#include <vector>

#include "../../mujoco_sim_amd/csrc/host_pool.h"

static thread_local std::string t_err;
static const char* last_error() { return t_err.c_str(); }

int main() {
  int failures = 0;
  {
    const int n = 8;
    std::vector<long long> acc(n, 0);
    std::vector<int> inited(n, 0);
    HostPool pool(n, [&](int k) { inited[k] = 1; }, last_error);
    for (int round = 0; round < 20000; round++) {
      std::string err;
      const int rc = pool.run([&](int k) -> int { acc[k] += round + k; if (round == 777 && k == 3) { t_err = "boom from 3"; return -7; } return 0; }, &err);
      if (round == 777) { if (rc != -7 || err != "boom from 3") failures++; }
      else if (rc != 0) failures++;
    }
    for (int k = 0; k < n; k++) {
      long long want = 0; for (int r = 0; r < 20000; r++) want += r + k;
      if (acc[k] != want || !inited[k]) failures++;
    }
  }
  for (int rep = 0; rep < 50; rep++) {      // life cycle: destroyed idle, and right after a job
    HostPool p(3, nullptr, last_error);
    if (rep & 1) { int x[3] = {0, 0, 0}; p.run([&](int k) -> int { x[k] = k + 1; return 0; }, nullptr); if (x[0] != 1 || x[2] != 3) failures++; }
  }
  std::printf("%d failures\n", failures);
  return failures ? 1 : 0;
}
