// group.hip — the environments of ONE simulation sharded over the GPUs of a node, driven by ONE caller thread through the C ABI (behind it: a
// host thread per device issues that device's launches — host_pool.h)
// (include/mjhip.h "multi-GPU").  The reference is one C++ node with one publisher set (src/mj_main.cpp:167-236,
// src/mujoco_sim/mj_ros.cpp:554-564); with many environments the stepper shards them — contiguous env ranges, one engine and one
// stream per device, model tables replicated, NO collective in the step (environments are independent) — and the only exchange
// is the all-gather of the published state slice (time | qpos | qvel per env, fp32) that feeds the single state / clock
// publisher: RCCL ncclAllGather over xGMI, issued at the publish rate, not per step.
// RCCL is resolved with dlopen (no link-time dependency: the library loads on hosts without it); without it the gather falls
// back to peer copies (hipMemcpyPeerAsync), which is also what a gather to ONE consumer would use.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mjhip.h"
#include "host_pool.h"

void mjh_set_error(const std::string& s);  // model_builder.cpp

#define GCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { mjh_set_error(std::string(#call) + ": " + hipGetErrorString(e_)); return MJH_ERR_NO_DEVICE; } } while (0)

// The handful of RCCL declarations the all-gather needs, stated here instead of #include <rccl/rccl.h>: the library is a RUN-TIME
// dependency only (dlopen below), so a ROCm install without the RCCL headers still builds libmjhip.so — single-GPU engine
// and peer-copy group included.  Values are NCCL's public ABI (ncclSuccess = 0, ncclFloat32 = 7).
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
static const ncclResult_t ncclSuccess = 0;
static const ncclDataType_t ncclFloat = 7;

namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;      // optional: after a partial failure of a per-thread exchange
  bool ok() const { return lib && CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd; }
};
Rccl* load_rccl() {
  static Rccl r; static bool tried = false;
  if (tried) return r.ok() ? &r : nullptr;
  tried = true;
  // a process that already holds RCCL (torch bundles its own) must not get a second copy: look the symbols up globally first
  // MJH_RCCL_LIB names the library instead (tests: tests/nccl_stub — a recording stand-in that lets the grouped-call and the
  // per-thread sequence run with N ranks where there is one device or none)
  const char* forced = getenv("MJH_RCCL_LIB");
  const char* names[] = {forced && *forced ? forced : nullptr, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (int ni = 0; ni < (forced && *forced ? 1 : 4); ni++) {
    const char* n = names[ni];
    void* h = n ? dlopen(n, RTLD_NOW | RTLD_GLOBAL) : dlopen(nullptr, RTLD_NOW);
    if (!h) continue;
    if (!dlsym(h, "ncclAllGather")) { if (n) dlclose(h); continue; }
    r.lib = h;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    r.CommAbort = (decltype(r.CommAbort))dlsym(h, "ncclCommAbort");
    if (r.ok()) return &r;
  }
  return nullptr;
}
int g_transport = getenv("MJH_GROUP_TRANSPORT") ? atoi(getenv("MJH_GROUP_TRANSPORT")) : 0;   // 0: RCCL when it can be loaded (and the devices are distinct), 1: peer copies, 2: RCCL also for repeated devices (only a stand-in library accepts that: tests)
}  // namespace

struct mjh_group {
  const mjh_model* model = nullptr;
  int nenv = 0, ndev = 0, stride = 0;
  size_t slot = 0;                                   // floats per rank in the gathered buffer: max_k n_k * stride
  std::vector<int> dev, env0, n;
  std::vector<mjh_engine*> eng;
  std::vector<hipStream_t> stream;
  std::vector<hipStream_t> comm;                     // per device: the exchange runs here, beside the steps queued on stream[k]
  std::vector<hipEvent_t> gathered;                  // comm[k] has finished the last publish (exchange + compaction)
  std::vector<hipEvent_t> ready;                     // send buffer of device k packed (peer-copy transport)
  std::vector<hipEvent_t> consumed;                  // stream k has finished reading every rank's send buffer (peer-copy transport)
  std::vector<hipEvent_t> released;                  // a consumer of device k's gathered buffer has finished reading it (mjh_group_release_publish)
  std::vector<char> has_release;
  bool published = false;                            // consumed[] have been recorded at least once
  hipEvent_t t0 = nullptr, t1 = nullptr;             // device 0's stream around the exchange (mjh_group_publish_timing)
  bool timing = false, t_pending = false; double t_sum_ms = 0; int t_count = 0;
  std::vector<float*> send, recv, packed;            // per device: own slice | every rank's slot | env-ordered, contiguous
  float* host = nullptr;                             // pinned staging of the gathered state (rank 0's copy)
  Rccl* rccl = nullptr; std::vector<ncclComm_t> ncomm;
  bool padded = false;
  // one host thread per device (ndev > 1; MJH_GROUP_THREADS=0 / mjh_group_set_host_threads(0): the caller's thread issues everything, one
  // device after the other): every entry point posts one job per device and waits for them
  std::unique_ptr<HostPool> pool;
  bool rccl_per_thread = true;     // threaded host: every device's thread enqueues its own ncclAllGather (NCCL's one-thread-per-device use) instead of one grouped call
};

extern "C" void mjh_group_set_transport(int mode) { g_transport = mode == 1 ? 1 : (mode == 2 ? 2 : 0); }
static int g_host_threads = getenv("MJH_GROUP_THREADS") ? atoi(getenv("MJH_GROUP_THREADS")) : 1;
extern "C" void mjh_group_set_host_threads(int on) { g_host_threads = on ? 1 : 0; }      // groups created afterwards
extern "C" int mjh_group_host_threads(const mjh_group* g) { return g && g->pool ? g->pool->size() : 0; }
// fn(k) for every device: on the device's own host thread when the group has them, else one after the other on the caller's
static int for_devices(mjh_group* g, const std::function<int(int)>& fn) {
  if (g->pool) {
    std::string err;
    const int rc = g->pool->run(fn, &err);
    if (rc) mjh_set_error(err);
    return rc;
  }
  for (int k = 0; k < g->ndev; k++) { const int rc = fn(k); if (rc) return rc; }
  return MJH_OK;
}


// The exchange of one publish over RCCL: every rank's all-gather of `slot` floats on its communication stream, then after(k) per device.
//  * pool != nullptr (a host thread per device): every device's thread enqueues its own rank (the library's one-thread-per-device use).
//    Nothing that can fail stands between a thread and its ncclAllGather — the device was selected by the thread's init job and by stage A —,
//    so no rank is left waiting for a partner that returned early; if a rank's call itself fails, the communicators are aborted (the other
//    ranks' collectives would otherwise stay enqueued without a partner) and the error is returned.
//  * pool == nullptr: ONE grouped call on the caller's thread, ncclGroupStart .. ncclGroupEnd around the ranks; every exit path passes GroupEnd.
// set_device = false and after = nullptr: the call sequence alone (mjh_debug_rccl_exchange: no HIP call at all).
static int rccl_exchange(Rccl* R, int ndev, const int* dev, float* const* send, float* const* recv, size_t slot, ncclComm_t* ncomm, hipStream_t* comm,
                         HostPool* pool, bool set_device, const std::function<int(int)>& after) {
  auto errtext = [&](ncclResult_t r) { return std::string("ncclAllGather: ") + (R->GetErrorString ? R->GetErrorString(r) : "failed"); };
  if (pool) {
    std::vector<ncclResult_t> res((size_t)ndev, ncclSuccess);
    std::string err;
    int rc = pool->run([&](int k) -> int {
      if (set_device) (void)hipSetDevice(dev[k]);
      res[(size_t)k] = R->AllGather(send[k], recv[k], slot, ncclFloat, ncomm[k], comm[k]);
      if (res[(size_t)k] != ncclSuccess) { mjh_set_error(errtext(res[(size_t)k])); return MJH_ERR_NO_DEVICE; }
      return after ? after(k) : MJH_OK;
    }, &err);
    if (rc) {
      bool partial = false;
      for (int k = 0; k < ndev; k++) if (res[(size_t)k] != ncclSuccess) partial = true;
      if (partial && R->CommAbort) for (int k = 0; k < ndev; k++) if (ncomm[k]) { (void)R->CommAbort(ncomm[k]); ncomm[k] = nullptr; }
      mjh_set_error(err);
    }
    return rc;
  }
  ncclResult_t r = R->GroupStart();
  hipError_t he = hipSuccess;
  if (r == ncclSuccess) {
    for (int k = 0; k < ndev && r == ncclSuccess && he == hipSuccess; k++) {
      if (set_device) he = hipSetDevice(dev[k]);
      if (he == hipSuccess) r = R->AllGather(send[k], recv[k], slot, ncclFloat, ncomm[k], comm[k]);
    }
    const ncclResult_t r2 = R->GroupEnd();
    if (r == ncclSuccess) r = r2;
  }
  if (he != hipSuccess) { mjh_set_error(std::string("hipSetDevice (all-gather): ") + hipGetErrorString(he)); return MJH_ERR_NO_DEVICE; }
  if (r != ncclSuccess) { mjh_set_error(errtext(r)); return MJH_ERR_NO_DEVICE; }
  for (int k = 0; k < ndev && after; k++) {
    if (set_device) { const hipError_t e2 = hipSetDevice(dev[k]); if (e2 != hipSuccess) { mjh_set_error(std::string("hipSetDevice: ") + hipGetErrorString(e2)); return MJH_ERR_NO_DEVICE; } }
    const int rc = after(k); if (rc) return rc;
  }
  return MJH_OK;
}
// The RCCL call sequence of one publish with `ndev` ranks and no device at all (tests on the CPU box, with MJH_RCCL_LIB = a recording
// stand-in): ncclCommInitAll, `publishes` exchanges of `slot` floats per rank exactly as mjh_group_publish issues them — per_thread != 0: a
// host thread per rank, else one grouped call —, ncclCommDestroy.  Buffers and streams are distinct made-up addresses, never dereferenced here.
extern "C" int mjh_debug_rccl_exchange(int ndev, int per_thread, unsigned long slot, int publishes) {
  if (ndev <= 0 || ndev > 64) { mjh_set_error("mjh_debug_rccl_exchange: bad rank count"); return MJH_ERR_ARG; }
  Rccl* R = load_rccl();
  if (!R) { mjh_set_error("mjh_debug_rccl_exchange: no RCCL library (MJH_RCCL_LIB)"); return MJH_ERR_NO_DEVICE; }
  std::vector<int> dev((size_t)ndev); std::vector<ncclComm_t> nc((size_t)ndev, nullptr);
  std::vector<float*> send((size_t)ndev), recv((size_t)ndev); std::vector<hipStream_t> st((size_t)ndev);
  for (int k = 0; k < ndev; k++) {
    dev[(size_t)k] = k; send[(size_t)k] = (float*)(uintptr_t)(0x10000000ull + 0x100000ull * (unsigned)k); recv[(size_t)k] = (float*)(uintptr_t)(0x20000000ull + 0x100000ull * (unsigned)k);
    st[(size_t)k] = (hipStream_t)(uintptr_t)(0x1000u + (unsigned)k);
  }
  ncclResult_t r = R->CommInitAll(nc.data(), ndev, dev.data());
  if (r != ncclSuccess) { mjh_set_error(std::string("ncclCommInitAll: ") + (R->GetErrorString ? R->GetErrorString(r) : "failed")); return MJH_ERR_NO_DEVICE; }
  std::unique_ptr<HostPool> pool;
  if (per_thread) pool.reset(new HostPool(ndev, nullptr, mjh_last_error));
  int rc = MJH_OK;
  for (int p = 0; p < publishes && !rc; p++)
    rc = rccl_exchange(R, ndev, dev.data(), send.data(), recv.data(), (size_t)slot, nc.data(), st.data(), pool.get(), false, nullptr);
  pool.reset();
  for (int k = 0; k < ndev; k++) if (nc[(size_t)k]) (void)R->CommDestroy(nc[(size_t)k]);
  return rc;
}

extern "C" void mjh_group_destroy(mjh_group* g) {
  if (!g) return;
  g->pool.reset();
  for (int k = 0; k < (int)g->eng.size(); k++) {
    (void)hipSetDevice(g->dev[k]);
    if (g->eng[k]) mjh_destroy(g->eng[k]);
    if (k < (int)g->ncomm.size() && g->ncomm[k] && g->rccl) (void)g->rccl->CommDestroy(g->ncomm[k]);
    if (k < (int)g->send.size() && g->send[k]) (void)hipFree(g->send[k]);
    if (k < (int)g->recv.size() && g->recv[k]) (void)hipFree(g->recv[k]);
    if (g->padded && k < (int)g->packed.size() && g->packed[k]) (void)hipFree(g->packed[k]);
    if (k < (int)g->ready.size() && g->ready[k]) (void)hipEventDestroy(g->ready[k]);
    if (k < (int)g->consumed.size() && g->consumed[k]) (void)hipEventDestroy(g->consumed[k]);
    if (k < (int)g->gathered.size() && g->gathered[k]) (void)hipEventDestroy(g->gathered[k]);
    if (k < (int)g->released.size() && g->released[k]) (void)hipEventDestroy(g->released[k]);
    if (k < (int)g->comm.size() && g->comm[k]) (void)hipStreamDestroy(g->comm[k]);
    if (k < (int)g->stream.size() && g->stream[k]) (void)hipStreamDestroy(g->stream[k]);
  }
  if (g->t0) (void)hipEventDestroy(g->t0);
  if (g->t1) (void)hipEventDestroy(g->t1);
  if (g->host) (void)hipHostFree(g->host);
  delete g;
}

extern "C" int mjh_group_create(const mjh_model* model, int nenv_total, const int* devices, int ndev, mjh_group** out) {
  if (!model || nenv_total <= 0 || ndev <= 0 || ndev > nenv_total || !out) { mjh_set_error("mjh_group_create: bad argument"); return MJH_ERR_ARG; }
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) { mjh_set_error("mjh_group_create: no HIP device visible (no CPU fallback)"); return MJH_ERR_NO_DEVICE; }
  mjh_group* g = new mjh_group();
  g->model = model; g->nenv = nenv_total; g->ndev = ndev;
  g->dev.resize(ndev); g->env0.resize(ndev); g->n.resize(ndev);
  g->eng.assign(ndev, nullptr); g->stream.assign(ndev, nullptr); g->ready.assign(ndev, nullptr); g->consumed.assign(ndev, nullptr); g->comm.assign(ndev, nullptr); g->gathered.assign(ndev, nullptr); g->released.assign(ndev, nullptr); g->has_release.assign(ndev, 0);
  g->send.assign(ndev, nullptr); g->recv.assign(ndev, nullptr); g->packed.assign(ndev, nullptr);
  bool distinct = true;
  for (int k = 0; k < ndev; k++) {
    g->dev[k] = devices ? devices[k] : k;
    if (g->dev[k] < 0 || g->dev[k] >= have) { mjh_set_error("mjh_group_create: bad device index"); mjh_group_destroy(g); return MJH_ERR_ARG; }
    for (int j = 0; j < k; j++) distinct &= g->dev[j] != g->dev[k];
    // contiguous env ranges, sizes differ by at most one (same rule as mujoco_sim_amd/shard.py: env_range)
    const int base = nenv_total / ndev, rem = nenv_total % ndev;
    g->env0[k] = k * base + std::min(k, rem); g->n[k] = base + (k < rem ? 1 : 0);
  }
  g->padded = nenv_total % ndev != 0;
#define GFAIL(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { mjh_set_error(std::string(#call) + ": " + hipGetErrorString(e_)); mjh_group_destroy(g); return MJH_ERR_NO_DEVICE; } } while (0)
  for (int k = 0; k < ndev; k++) {
    GFAIL(hipSetDevice(g->dev[k]));
    GFAIL(hipStreamCreateWithFlags(&g->stream[k], hipStreamNonBlocking));
    GFAIL(hipEventCreateWithFlags(&g->ready[k], hipEventDisableTiming));
    GFAIL(hipEventCreateWithFlags(&g->consumed[k], hipEventDisableTiming));
    GFAIL(hipEventCreateWithFlags(&g->gathered[k], hipEventDisableTiming));
    GFAIL(hipEventCreateWithFlags(&g->released[k], hipEventDisableTiming));
    GFAIL(hipStreamCreateWithFlags(&g->comm[k], hipStreamNonBlocking));
    const int rc = mjh_create(model, g->n[k], g->dev[k], g->stream[k], &g->eng[k]);
    if (rc) { mjh_group_destroy(g); return rc; }
  }
  g->stride = mjh_state_stride(g->eng[0]);
  g->slot = (size_t)g->n[0] * g->stride;             // n[0] is the largest share
  for (int k = 0; k < ndev; k++) {
    GFAIL(hipSetDevice(g->dev[k]));
    GFAIL(hipMalloc((void**)&g->send[k], g->slot * sizeof(float)));
    GFAIL(hipMemsetAsync(g->send[k], 0, g->slot * sizeof(float), g->stream[k]));
    GFAIL(hipMalloc((void**)&g->recv[k], g->slot * ndev * sizeof(float)));
    if (g->padded) GFAIL(hipMalloc((void**)&g->packed[k], (size_t)nenv_total * g->stride * sizeof(float)));
    else g->packed[k] = g->recv[k];
    GFAIL(hipStreamSynchronize(g->stream[k]));
  }
  GFAIL(hipHostMalloc((void**)&g->host, (size_t)nenv_total * g->stride * sizeof(float), hipHostMallocDefault));
  GFAIL(hipSetDevice(g->dev[0]));
  GFAIL(hipEventCreate(&g->t0)); GFAIL(hipEventCreate(&g->t1));
#undef GFAIL
  if ((g_transport == 0 && distinct) || g_transport == 2) {
    g->rccl = load_rccl();
    if (g->rccl) {
      g->ncomm.assign(ndev, nullptr);
      const ncclResult_t r = g->rccl->CommInitAll(g->ncomm.data(), ndev, g->dev.data());
      if (r != ncclSuccess) {
        mjh_set_error(std::string("ncclCommInitAll: ") + (g->rccl->GetErrorString ? g->rccl->GetErrorString(r) : "failed"));
        g->ncomm.clear(); g->rccl = nullptr;           // keep going on peer copies; mjh_group_uses_rccl() says so
      }
    }
  }
  if (ndev > 1 && g_host_threads) {
    std::vector<int> devs = g->dev;
    g->pool.reset(new HostPool(ndev, [devs](int k) { (void)hipSetDevice(devs[k]); }, mjh_last_error));
    if (const char* v = getenv("MJH_GROUP_RCCL_PER_THREAD")) g->rccl_per_thread = atoi(v) != 0;
  }
  *out = g;
  return MJH_OK;
}

extern "C" int mjh_group_ndev(const mjh_group* g) { return g ? g->ndev : 0; }
extern "C" int mjh_group_nenv(const mjh_group* g) { return g ? g->nenv : 0; }
extern "C" int mjh_group_uses_rccl(const mjh_group* g) { return g && g->rccl ? 1 : 0; }
extern "C" mjh_engine* mjh_group_engine(mjh_group* g, int k) { return g && k >= 0 && k < g->ndev ? g->eng[k] : nullptr; }
extern "C" int mjh_group_env_range(const mjh_group* g, int k, int* env0, int* n) {
  if (!g || k < 0 || k >= g->ndev) { mjh_set_error("mjh_group_env_range: bad rank"); return MJH_ERR_ARG; }
  if (env0) *env0 = g->env0[k];
  if (n) *n = g->n[k];
  return MJH_OK;
}
extern "C" int mjh_group_locate(const mjh_group* g, int env, int* rank, int* local) {
  if (!g || env < 0 || env >= g->nenv) { mjh_set_error("mjh_group_locate: env out of range"); return MJH_ERR_ARG; }
  for (int k = 0; k < g->ndev; k++) if (env < g->env0[k] + g->n[k]) { if (rank) *rank = k; if (local) *local = env - g->env0[k]; return MJH_OK; }
  return MJH_ERR_ARG;
}

// one call per device, all asynchronous: the devices step concurrently, the host thread never waits here
#define FOR_ALL(expr) return for_devices(g, [&](int k) -> int { mjh_engine* e = g->eng[k]; return (expr); })
extern "C" int mjh_group_step(mjh_group* g, int nsteps, int with_inverse) { if (!g) return MJH_ERR_ARG; FOR_ALL(mjh_step(e, nsteps, with_inverse)); }
extern "C" int mjh_group_step1(mjh_group* g) { if (!g) return MJH_ERR_ARG; FOR_ALL(mjh_step1(e)); }
extern "C" int mjh_group_step2(mjh_group* g) { if (!g) return MJH_ERR_ARG; FOR_ALL(mjh_step2(e)); }
extern "C" int mjh_group_inverse(mjh_group* g) { if (!g) return MJH_ERR_ARG; FOR_ALL(mjh_inverse(e)); }
extern "C" int mjh_group_reset(mjh_group* g) { if (!g) return MJH_ERR_ARG; FOR_ALL(mjh_reset(e, nullptr, 0)); }
extern "C" int mjh_group_synchronize(mjh_group* g) {
  if (!g) return MJH_ERR_ARG;
  return for_devices(g, [&](int k) -> int {
    const int rc = mjh_synchronize(g->eng[k]); if (rc) return rc;
    GCHK(hipStreamSynchronize(g->comm[k]));                            // (the last publish's exchange as well)
    return MJH_OK;
  });
}
#undef FOR_ALL

// Publish: every device packs its slice (time | qpos | qvel per env) behind the steps queued so far, then ONE all-gather
// leaves the full, env-ordered state on every device; host_out (optional, [nenv * stride] floats) receives device 0's copy.
static void collect_timing(mjh_group* g) {
  if (!g->t_pending) return;
  float ms = 0;
  if (hipEventSynchronize(g->t1) == hipSuccess && hipEventElapsedTime(&ms, g->t0, g->t1) == hipSuccess) { g->t_sum_ms += ms; g->t_count++; }
  g->t_pending = false;
}
extern "C" int mjh_group_publish(mjh_group* g, float* host_out) {
  if (!g) { mjh_set_error("null group"); return MJH_ERR_ARG; }
  const size_t slot_bytes = g->slot * sizeof(float);
  if (g->timing) collect_timing(g);
  // The exchange runs on a communication stream per device, BESIDE the steps: the engine's stream only waits for the previous
  // publish before it overwrites the send buffer (three steps later at 60 Hz: long done), and never for the current one — a
  // collective on the stepping stream would sit between two steps of every cohort and drain the pipeline (measured on one
  // device: 7.0 M against 9.1 M env-steps/s on S24).
  // stage A, per device: the engine's stream waits for the previous publish, packs the send buffer, hands over to the communication stream
  int rc = for_devices(g, [&](int k) -> int {
    GCHK(hipSetDevice(g->dev[k]));
    if (g->published) {
      GCHK(hipStreamWaitEvent(g->stream[k], g->gathered[k], 0));          // (own collective has read send[k] / written recv[k])
      // peer-copy transport: every consumer must be done reading the previous copy of send[k] (a lagging device would otherwise
      // gather a torn slice, or one from a later step)
      if (!g->rccl) for (int r = 0; r < g->ndev; r++) if (r != k) GCHK(hipStreamWaitEvent(g->stream[k], g->consumed[r], 0));
    }
    const int rce = mjh_export_state_device(g->eng[k], g->send[k]);      // (re-selects device k)
    if (rce) return rce;
    GCHK(hipEventRecord(g->ready[k], g->stream[k]));
    GCHK(hipStreamWaitEvent(g->comm[k], g->ready[k], 0));
    // a consumer that is still reading the previous gathered state of this device (mjh_group_release_publish): the exchange below
    // overwrites that buffer
    if (g->has_release[k]) { GCHK(hipStreamWaitEvent(g->comm[k], g->released[k], 0)); g->has_release[k] = 0; }
    if (g->timing && k == 0) GCHK(hipEventRecord(g->t0, g->comm[0]));
    return MJH_OK;
  });
  if (rc) return rc;
  // stage B: the exchange (every ready[] event has been recorded: stage A is complete on every device), then the compaction
  auto finish = [&](int k) -> int {
    if (g->timing && k == 0) { GCHK(hipEventRecord(g->t1, g->comm[0])); g->t_pending = true; }
    if (g->padded)     // uneven shares: the ranks' slots carry padding behind the smaller shares; close the gaps
      for (int r = 0; r < g->ndev; r++)
        GCHK(hipMemcpyAsync(g->packed[k] + (size_t)g->env0[r] * g->stride, g->recv[k] + (size_t)r * g->slot, (size_t)g->n[r] * g->stride * sizeof(float),
                            hipMemcpyDeviceToDevice, g->comm[k]));
    GCHK(hipEventRecord(g->gathered[k], g->comm[k]));
    return MJH_OK;
  };
  if (g->rccl) {
    rc = rccl_exchange(g->rccl, g->ndev, g->dev.data(), g->send.data(), g->recv.data(), g->slot, g->ncomm.data(), g->comm.data(),
                       g->rccl_per_thread ? g->pool.get() : nullptr, true, [&](int k) -> int { return finish(k); });
    if (rc) return rc;
  } else {
    rc = for_devices(g, [&](int k) -> int {
      GCHK(hipSetDevice(g->dev[k]));
      for (int r = 0; r < g->ndev; r++) {
        GCHK(hipStreamWaitEvent(g->comm[k], g->ready[r], 0));
        GCHK(hipMemcpyPeerAsync(g->recv[k] + (size_t)r * g->slot, g->dev[k], g->send[r], g->dev[r], slot_bytes, g->comm[k]));
      }
      GCHK(hipEventRecord(g->consumed[k], g->comm[k]));
      return finish(k);
    });
    if (rc) return rc;
  }
  g->published = true;
  if (host_out) {
    GCHK(hipSetDevice(g->dev[0]));
    const size_t bytes = (size_t)g->nenv * g->stride * sizeof(float);
    GCHK(hipMemcpyAsync(g->host, g->packed[0], bytes, hipMemcpyDeviceToHost, g->comm[0]));
    GCHK(hipStreamSynchronize(g->comm[0]));
    std::memcpy(host_out, g->host, bytes);
  }
  return MJH_OK;
}
// makes `stream` (a stream of device `rank`; NULL: that device's engine stream) wait for the last publish: from then on, in that
// stream's order, mjh_group_state_device(rank) holds the gathered state
extern "C" int mjh_group_wait_publish(mjh_group* g, int rank, void* stream) {
  if (!g || rank < 0 || rank >= g->ndev) { mjh_set_error("mjh_group_wait_publish: bad rank"); return MJH_ERR_ARG; }
  if (!g->published) return MJH_OK;
  GCHK(hipSetDevice(g->dev[rank]));
  GCHK(hipStreamWaitEvent(stream ? (hipStream_t)stream : g->stream[rank], g->gathered[rank], 0));
  return MJH_OK;
}
// the consumer's side of the hand-over: `stream` (a stream of device `rank`) has read the gathered state up to this point of its order;
// the next publish's exchange on that device waits for it before it overwrites the buffer.  A consumer that reads asynchronously and
// does not call this must finish (synchronise) before the next mjh_group_publish.
extern "C" int mjh_group_release_publish(mjh_group* g, int rank, void* stream) {
  if (!g || rank < 0 || rank >= g->ndev) { mjh_set_error("mjh_group_release_publish: bad rank"); return MJH_ERR_ARG; }
  GCHK(hipSetDevice(g->dev[rank]));
  GCHK(hipEventRecord(g->released[rank], stream ? (hipStream_t)stream : g->stream[rank]));
  g->has_release[rank] = 1;
  return MJH_OK;
}
// HIP events on device 0's communication stream around the exchange (all-gather or peer copies) of every publish: the collective's own time
extern "C" int mjh_group_set_publish_timing(mjh_group* g, int on) {
  if (!g) { mjh_set_error("null group"); return MJH_ERR_ARG; }
  g->timing = on != 0; g->t_pending = false; g->t_sum_ms = 0; g->t_count = 0;
  return MJH_OK;
}
extern "C" int mjh_group_get_publish_timing(mjh_group* g, double* mean_ms, int* count) {
  if (!g) { mjh_set_error("null group"); return MJH_ERR_ARG; }
  collect_timing(g);
  if (mean_ms) *mean_ms = g->t_count ? g->t_sum_ms / g->t_count : 0.0;
  if (count) *count = g->t_count;
  g->t_sum_ms = 0; g->t_count = 0;
  return MJH_OK;
}
extern "C" const float* mjh_group_state_device(const mjh_group* g, int k) { return g && k >= 0 && k < g->ndev ? g->packed[k] : nullptr; }
extern "C" int mjh_group_state_stride(const mjh_group* g) { return g ? g->stride : 0; }
