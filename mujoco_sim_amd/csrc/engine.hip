// engine.hip — host side of the C ABI (include/mjhip.h): device model upload, per-env state in
// HBM (env-major fp32 rows padded to 128 B), kernel launches on the caller's HIP stream, and
// double<->float marshalling for the mjData-style getters/setters the reference's ROS layer uses.
// gfx950 only; there is no CPU fallback — every entry point fails loudly without a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mjhip.h"
#include "step_kernel.h"

void mjh_set_error(const std::string& s);  // model_builder.cpp
hipError_t mjh_launch_window(hipStream_t st, int nvt, int grid, size_t lds, const DConst* dC, const DState& S, int env0, int n, int nl, int wxf, int n32, int n64);   // window.hip
hipError_t mjh_dense_attributes(size_t build_lds, size_t solve_lds);   // dense.hip (kernels of dense_pgs.h)
hipError_t mjh_launch_dense(hipStream_t st, int n, size_t build_lds, size_t solve_lds, const DConst* dC, const DState& S, int env0);
#define DN_CAP_MAX 256          // row capacity of the dense solver (dense_pgs.h)

#define HIPCHK(call)                                                                             \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      mjh_set_error(std::string(#call) + ": " + hipGetErrorString(e_));                          \
      return MJH_ERR_NO_DEVICE;                                                                  \
    }                                                                                            \
  } while (0)

// restores the engine's device-state descriptor when a scope that patched it (export pointers, launch order) is left on
// ANY path, early error returns included
struct StateGuard {
  DState* where; DState saved;
  explicit StateGuard(DState* w) : where(w), saved(*w) {}
  ~StateGuard() { *where = saved; }
  StateGuard(const StateGuard&) = delete; StateGuard& operator=(const StateGuard&) = delete;
};
struct DevBuf {   // temporary device allocation freed on every path
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
};

#define MJH_MAX_COHORTS 8
#define MJH_IO_SLOTS 8             // host-mapped staging of mjh_set_cmd: ring of slots ...
#define MJH_IO_PUT_FLOATS 4096     // ... of this many floats each (larger calls take the copy path)
#define MJH_IO_GET_FLOATS 16384    // host-mapped staging of mjh_get_joint_state
// HIP streams that share a hardware queue serialise; the engine uses up to five at once (three cohorts, the caller's stream, the
// export stream) and RCCL adds its own.  The runtime reads GPU_MAX_HW_QUEUES (default 4) when it initialises, i.e. at the first
// HIP call of the process: asked for here, when the library is loaded, unless the host has set it itself.
__attribute__((constructor)) static void mjh_library_init() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }
struct mjh_engine {
  const mjh_model* model = nullptr;
  int nenv = 0, device = 0;
  hipStream_t stream = nullptr;
  DModel M{};
  DState S{};
  Lay L{};
  int lds_bytes = 0, lds_bytes_pre = 0;   // dynamic LDS per env of the step kernel; ... of its assemble-only instance (no patch pool: more envs per CU)
  size_t dense_lds = 0, dense_solve_lds = 0;   // dynamic LDS of mjh_dense_build_kernel / mjh_dense_solve_kernel
  // dense solver on / off per cohort: mjh_order_kernel leaves "an env of the cohort swept long" in a host-mapped word (four slots per
  // cohort, one per rebuild of the launch order); the host adopts the word of TWO rebuilds ago after waiting for that kernel's event
  // (long finished: no stall, and the decision depends on the step count only, not on timing: runs stay reproducible)
  // per-step host traffic of the reference's loop (MjHWInterface::read / write: mjh_get_joint_state / mjh_set_cmd on a few envs): a
  // host-mapped staging area the device reads and writes directly — one small kernel per call instead of two or three pageable 2D
  // copies; the write side is a ring of slots, each fenced by an event, so mjh_set_cmd does not wait for the device
  float* h_io = nullptr; float* d_io = nullptr; hipEvent_t io_ev[MJH_IO_SLOTS] = {}; bool io_used[MJH_IO_SLOTS] = {}; unsigned io_next = 0;
  int* h_wn = nullptr; int* d_wn = nullptr; int cur_cohort = -1;   // window models: largest row count per cohort (host-mapped, written by mjh_order_kernel); cohort of the launch being issued
  int* h_dense = nullptr; int* d_dense = nullptr; hipEvent_t ev_dense[MJH_MAX_COHORTS][4] = {}; unsigned dense_epoch[MJH_MAX_COHORTS] = {}; bool dense_now[MJH_MAX_COHORTS];
  int* dI = nullptr; float* dF = nullptr; DConst* dC = nullptr;
  std::vector<int> hI;  // host copy of the int tables (controlled / odom are patched in place)
  int o_controlled = 0, o_odom = 0;
  std::vector<void*> allocs;
  float* scratch = nullptr; size_t scratch_floats = 0;  // export staging
  float* p_tables[MJH_EP_COUNT] = {nullptr};
  int* d_order = nullptr;   // LPT launch order (mjh_order_kernel)
  bool lpt = true;
  bool split3 = true;         // many-body layout: three-launch step (MJH_SPLIT3=0: fused kernel)
  bool order_valid = false;   // d_order holds a full-range permutation (split API); mjh_step sorts per cohort
  // one captured graph per cohort and variant of its step chain (mjh_step: assemble -> [dense build -> dense solve] -> solve -> integrate of the
  // many-body layout; assemble -> window kernel of the window chain): the chain is queued with ONE hipGraphLaunch per cohort-step
  struct ChainGraph { hipGraphExec_t exec = nullptr; unsigned char S[sizeof(DState)]; int g0 = -1, n = -1, key = -1; bool nocap = false; };   // nocap: capture / instantiate failed once for this slot — plain launches from then on
  int last_launches = 0;      // queue entries the last mjh_step issued per cohort-step (1: a captured graph; else the chain's kernel launches)
  ChainGraph cgraph[MJH_MAX_COHORTS][8];
  int steps_per_launch = 8;   // mjh_step(n): steps one launch of a loop-capable kernel instance runs (mjh_set_steps_per_launch; 1: one launch per step)
  long order_age = 0; int order_G = 0; int last_chunk = 1;   // last_chunk: steps of the previous launch (the sort is renewed when a multiple of MJH_ORDER_EVERY was crossed)   // mjh_step renews its per-cohort sorts every MJH_ORDER_EVERY-th step; order_G: cohort count they were made for (-1: none)
  // Cohorts: mjh_step() splits the envs into ncohort contiguous groups, each stepped on its own stream, so that the
  // low-occupancy tail of one cohort's step kernel overlaps the next cohort's (or its own next step's) bulk.  The
  // caller's stream forks into the cohort streams at mjh_step and joins them again at the next other API call.
  int ncohort = 1;
  hipStream_t cstream[MJH_MAX_COHORTS] = {nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[MJH_MAX_COHORTS] = {nullptr};
  bool forked = false;
  // optional per-launch timing of the step kernels (mjh_set_launch_timing)
  bool timing = false; int timing_stride = 1; long timing_count = 0;   // every timing_stride-th step launch is bracketed
  std::vector<std::pair<hipEvent_t, hipEvent_t>> tev; size_t tev_used = 0;
  bool step1_done = false;
  bool handover = false;        // window chain: the last step1 [+ inverse] launch left the hand-over mjh_step2 sweeps (dropped by any call that may change what it was built from)
  bool step1_pending = false;   // mjh_step1 has been called, its launch is deferred to the next entry point (fused with mjh_inverse if that is the one)
  // in-engine joint-space PD effort controller (mjh_set_pd_controller): ddq written on the device in front of every step
  float pd_kp = 0, pd_kd = 0; float* pd_target = nullptr; bool pd_on = false;
};

static int pad32(int n) { return ((n + 31) / 32) * 32; }

template <class T> static int dev_alloc(mjh_engine* e, T** p, size_t n, bool zero = true) {
  void* q = nullptr;
  HIPCHK(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
  if (zero) HIPCHK(hipMemsetAsync(q, 0, std::max<size_t>(n, 1) * sizeof(T), e->stream));
  e->allocs.push_back(q);
  *p = (T*)q;
  return MJH_OK;
}

// 1 (default): mjh_step queues the many-body layout's launch chain of a cohort-step as one captured graph; 2: the window chain as well (S24:
// 11.42 against 11.63 M env-steps/s with two plain launches — the graph's own launch costs more than it saves there); 0: never
static int g_chain_graph = getenv("MJH_CHAIN_GRAPH") ? atoi(getenv("MJH_CHAIN_GRAPH")) : 1;
extern "C" void mjh_set_chain_graph(int mode) { g_chain_graph = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }

static int ensure_scratch(mjh_engine* e, size_t floats) {
  if (floats <= e->scratch_floats) return MJH_OK;
  if (e->scratch) { HIPCHK(hipStreamSynchronize(e->stream)); HIPCHK(hipFree(e->scratch)); }
  HIPCHK(hipMalloc((void**)&e->scratch, floats * sizeof(float)));
  e->scratch_floats = floats;
  return MJH_OK;
}

// models that need the EXTRA kernel instances: generic convex narrow phase (cylinder-x, capsule-box, ellipsoid, mesh) or noslip sweeps
// ... or sites / sensors / mocap bodies / connect-weld equalities; and (engine state) Cartesian forces on bodies in use
static bool extra_instance(const DModel& M) { return M.has_convex || M.noslip_iterations > 0 || M.nsensor > 0 || M.nmocap > 0 || M.has_weld; }

// windows of the window kernel's LDS tier for the next launch of cohort e->cur_cohort
static int window_tier(const mjh_engine* e) {
  static const int nl_env = getenv("MJH_WN_NL") ? atoi(getenv("MJH_WN_NL")) : -1;      // (experiments: force the number of LDS-tier windows)
  // ... and none at all while no env of the cohort comes near the register-resident windows' rows: mjh_order_kernel leaves the cohort's
  // largest row count in a host-mapped word whenever it renews the launch order; read unsynchronised — the tiers hold the same values,
  // the choice changes where a window waits, not what is computed (S24: 9.13 -> 9.35 M env-steps/s; S24D needs the tier)
  const int nwreg = e->M.win_nvt == 24 ? WN_NW24 : WN_NW32;
  const int seen = (e->h_wn && e->cur_cohort >= 0) ? *(volatile int*)(e->h_wn + e->cur_cohort) : (1 << 20);
  const int nl_full = std::min(WN_MAXW, (int)((40 * 1024 - 1024) / (4 * WN_XREC(e->M.win_nvt) * 16 * sizeof(float))));
  // (with the 32-row section on, the 16-row form only meets envs of at most WN32_MIN_ROWS rows — or more than 128, the tier's clients)
  const bool sec32 = e->S.win32 > 0 && e->M.win_nvt == 24 && e->S.win32 <= 16 * nwreg;
  const bool tier = sec32 ? seen > 32 * WN32_NW : seen + 16 > 16 * nwreg;
  // (models whose rows can exceed 256 — win_maxw > 16 — always get the tier: the 64-row section keeps the tiles of its last windows there,
  //  and which envs take that section is decided on the device)
  return nl_env >= 0 ? nl_env : ((tier || e->M.win_maxw > 16) ? nl_full : 0);
}
// wmode (window chain of small free-body models): 0 the whole step; 1 the assemble launch only, every env handed over — the split API's
// mjh_step1 [+ mjh_inverse] doing the work mjh_step2 would otherwise repeat (the rows only exist in LDS); 2 the window kernel only
static int launch_on(mjh_engine* e, hipStream_t st, int env0, int n, int nsteps, int ph, int xflags, int wmode = 0, int tier_nl = -1) {
  if (n <= 0) return MJH_OK;
  // window sweep (window_pgs.h): every launch that runs mj_step2 to the end (the fused step and the split API's step2) of a small
  // free-body model = assemble launch (PH_PRE), then four envs per wavefront through the sweeps and the integration
  const bool window = e->M.window && e->S.wbuf && (ph & PH_STEP2) && !(ph & (PH_NOINT | PH_PRE | PH_POST)) && !(xflags & ~(XF_FORCE | XF_DEFER));
  if (window) ph |= PH_PRE;
  if (wmode && !window) { mjh_set_error("internal: window-chain launch mode on a launch that is not one"); return MJH_ERR_STATE; }
  if (wmode == 1) xflags |= XF_DEFER;
  if (wmode != 2) {
#define MJH_LAUNCH2(NR, DG, CX) hipLaunchKernelGGL((mjh_step_kernel<NR, DG, CX>), dim3(n), dim3(64), (size_t)e->lds_bytes, st, e->dC, e->S, env0, nsteps, ph, xflags)
#define MJH_LAUNCH(NR, DG) do { if (extra_instance(e->M) || e->S.xfrc_applied) MJH_LAUNCH2(NR, DG, true); else MJH_LAUNCH2(NR, DG, false); } while (0)
  const int nr = e->M.big ? 8 : (e->M.nv <= 16 ? 1 : (e->M.nv <= 32 ? 2 : 4));   // 8: many-body layout, running acceleration in LDS
  static const bool slim = !(getenv("MJH_WINDOW_SLIM") && atoi(getenv("MJH_WINDOW_SLIM")) == 0);
  if (window && slim) {
    // the assemble-only instance (WPRE): the step kernel without any sweep of its own — 128 VGPRs instead of 236, so that its waves
    // fit beside the window kernel's on a SIMD
    const bool cx = extra_instance(e->M) || e->S.xfrc_applied;
    static const bool slim_lds = !(getenv("MJH_WINDOW_SLIM_LDS") && atoi(getenv("MJH_WINDOW_SLIM_LDS")) == 0);
    static const int wpad = getenv("MJH_WPRE_LDS_PAD") ? std::max(0, atoi(getenv("MJH_WPRE_LDS_PAD"))) : 0;      // (occupancy experiments: bytes of unused LDS per assemble-only workgroup)
    const size_t wlds = (size_t)((slim_lds && e->lds_bytes_pre > 0) ? e->lds_bytes_pre : e->lds_bytes) + (size_t)wpad;
    // (models of up to 64 contacts keep the base-row pool in LDS — instance 1 —, larger ones in the env's window slice — instance 2: derive_device_model)
#define MJH_LAUNCHW(NR, CX, GJ) hipLaunchKernelGGL((mjh_step_kernel<NR, true, CX, GJ>), dim3(n), dim3(64), wlds, st, e->dC, e->S, env0, nsteps, ph, xflags)
#define MJH_LAUNCHW2(NR, CX) do { if (e->M.patch) MJH_LAUNCHW(NR, CX, 1); else MJH_LAUNCHW(NR, CX, 2); } while (0)
    if (nr == 1) { if (cx) MJH_LAUNCHW2(1, true); else MJH_LAUNCHW2(1, false); } else { if (cx) MJH_LAUNCHW2(2, true); else MJH_LAUNCHW2(2, false); }
#undef MJH_LAUNCHW2
#undef MJH_LAUNCHW
  } else
  if (e->M.diagM) { if (nr == 1) MJH_LAUNCH(1, true); else if (nr == 2) MJH_LAUNCH(2, true); else if (nr == 4) MJH_LAUNCH(4, true); else MJH_LAUNCH(8, true); }
  else { if (nr == 1) MJH_LAUNCH(1, false); else if (nr == 2) MJH_LAUNCH(2, false); else if (nr == 4) MJH_LAUNCH(4, false); else MJH_LAUNCH(8, false); }
#undef MJH_LAUNCH2
#undef MJH_LAUNCH
  HIPCHK(hipGetLastError());
  }
  if (window && wmode != 1) {
    // LDS tier: windows beyond the register-resident ones, as many as leave four waves per CU (40 KB per wave)
    const int nl = tier_nl >= 0 ? tier_nl : window_tier(e);
    const size_t lds = (size_t)4 * nl * WN_XREC(e->M.win_nvt) * 16 * sizeof(float);
    // 24-dof models: a first section of wavefronts sweeps the envs with many rows in 32-row windows, two per wavefront (they scan the
    // same launch order and take the envs the assemble launch marked; almost all of them exit at once)
    const int n32 = (e->S.win32 && e->M.win_nvt == 24) ? (n + 1) / 2 : 0;
    const int wxf = (xflags & ~XF_DEFER) | ((ph & PH_STEP1) ? 0 : XF_SPLIT2);
    const int n64 = (e->S.win64 && e->M.win_nvt == 24 && e->S.win64 < 16 * e->M.win_maxw) ? n : 0;      // the 64-row section: one wavefront per env of the launch order, almost all of them exit at once
    HIPCHK(mjh_launch_window(st, e->M.win_nvt, n64 + (e->M.win_nvt == 24 ? n32 : 0) + (n + 3) / 4, lds, e->dC, e->S, env0, n, nl, wxf, n32, n64));
  }
  return MJH_OK;
}

static int launch(mjh_engine* e, int env0, int n, int nsteps, int ph, int xflags, int wmode = 0) { return launch_on(e, e->stream, env0, n, nsteps, ph, xflags, wmode); }

static int fork_cohorts(mjh_engine* e) {
  if (e->forked || e->ncohort <= 1) return MJH_OK;
  HIPCHK(hipEventRecord(e->ev_fork, e->stream));
  for (int g = 0; g < e->ncohort; g++) HIPCHK(hipStreamWaitEvent(e->cstream[g], e->ev_fork, 0));
  e->forked = true;
  return MJH_OK;
}
static int join_cohorts(mjh_engine* e) {
  if (!e->forked) return MJH_OK;
  for (int g = 0; g < e->ncohort; g++) {
    HIPCHK(hipEventRecord(e->ev_join[g], e->cstream[g]));
    HIPCHK(hipStreamWaitEvent(e->stream, e->ev_join[g], 0));
  }
  e->forked = false;
  return MJH_OK;
}
static int set_cohorts(mjh_engine* e, int n) {
  n = std::max(1, std::min(n, MJH_MAX_COHORTS));
  int rc = join_cohorts(e);
  if (rc) return rc;
  if (n > 1) {
    if (!e->ev_fork) HIPCHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    for (int g = 0; g < n; g++) {
      if (!e->cstream[g]) HIPCHK(hipStreamCreateWithFlags(&e->cstream[g], hipStreamNonBlocking));
      if (!e->ev_join[g]) HIPCHK(hipEventCreateWithFlags(&e->ev_join[g], hipEventDisableTiming));
    }
  }
  e->ncohort = n;
  return MJH_OK;
}

static int pair_cap(int t1, int t2) {
  if (t1 > t2) std::swap(t1, t2);
  if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_BOX) return 4;
  if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_CAPSULE) return 2;
  if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_CYLINDER) return 4;
  if (t1 == MJH_GEOM_PLANE && t2 == MJH_GEOM_MESH) return 4;
  if (t1 == MJH_GEOM_BOX && t2 == MJH_GEOM_BOX) return 8;
  return 1;
}

// Host-only derivation of the device model: packed tables, derived topology tables, capacities and the LDS
// layout.  Needs no HIP device (mjh_query_lds_bytes uses it for capacity planning and in the CPU tests).
struct HostPack { DModel M{}; Lay L{}; std::vector<int> I; std::vector<float> F; int o_controlled = 0, o_odom = 0, lds_bytes = 0, lds_bytes_pre = 0; long long gstride = 0; };
// Gauss-Seidel order of engines created afterwards (mjhip.h): 1 = mj_solPGS's own row order
#define MJH_WINDOW_MAXCON 128    // contact capacity up to which a small free-body model steps through the window chain
static int g_window_solver = getenv("MJH_WINDOW") ? (atoi(getenv("MJH_WINDOW")) != 0) : 1;
extern "C" void mjh_set_window_solver(int on) { g_window_solver = on ? 1 : 0; }
static int g_pgs_row_order = 1;
extern "C" void mjh_set_pgs_row_order(int mode) { g_pgs_row_order = mode < 0 || mode > 2 ? 1 : mode; }
static void derive_device_model(const mjh_model* m, HostPack& hp, bool force_big = false, bool allow_patch = true) {
  DModel& M = hp.M; std::vector<int>& I = hp.I; std::vector<float>& F = hp.F;
  // ---- derived integer tables
  const int nb = m->nbody, nv = m->nv, nj = m->njnt, ng = m->ngeom;
  std::vector<int> subtreesize(nb, 1), lastdof(nb, -1), stageadr(m->npair, 0), fl_dof, gc_body, controlled(std::max(nv, 1), 0), odom(10, -1);
  odom[9] = 0;
  for (int b = nb - 1; b > 0; b--) subtreesize[m->body_parentid[b]] += subtreesize[b];
  for (int b = 1; b < nb; b++) lastdof[b] = m->body_dofnum[b] ? m->body_dofadr[b] + m->body_dofnum[b] - 1 : lastdof[m->body_parentid[b]];
  int nstage = 0, maxlevel = 0, rowW = 1;
  for (int b = 0; b < nb; b++) maxlevel = std::max(maxlevel, m->body_level[b]);
  auto treenum = [&](int b) { int t = m->body_treeid[b]; return t >= 0 ? m->tree_dofnum[t] : 0; };
  for (int i = 0; i < m->npair; i++) {
    int g1 = m->pair_geom1[i], g2 = m->pair_geom2[i];
    stageadr[i] = nstage; nstage += pair_cap(m->geom_type[g1], m->geom_type[g2]);
    int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
    int w = treenum(b1) + ((m->body_treeid[b1] != m->body_treeid[b2]) ? treenum(b2) : 0);
    rowW = std::max(rowW, w);
  }
  for (int t = 0; t < m->ntree; t++) rowW = std::max(rowW, m->tree_dofnum[t]);
  int neqrow = 0; bool has_weld = false;
  for (int q = 0; q < m->neq; q++) {
    if (m->eq_type[q] != MJH_EQ_JOINT) {      // connect / weld between bodies: 3 / 6 single-row blocks over the bodies' trees
      has_weld = true; neqrow += m->eq_type[q] == MJH_EQ_WELD ? 6 : 3;
      rowW = std::max(rowW, treenum(m->eq_obj1id[q]) + ((m->body_treeid[m->eq_obj1id[q]] != m->body_treeid[m->eq_obj2id[q]]) ? treenum(m->eq_obj2id[q]) : 0));
      continue;
    }
    neqrow++;
    int j1 = m->eq_obj1id[q], j2 = m->eq_obj2id[q];
    int t1 = m->dof_treeid[m->jnt_dofadr[j1]], w = m->tree_dofnum[t1];
    if (j2 >= 0) { int t2 = m->dof_treeid[m->jnt_dofadr[j2]]; if (t2 != t1) w += m->tree_dofnum[t2]; }
    rowW = std::max(rowW, w);
  }
  rowW = ((rowW + 3) / 4) * 4;
  // every tree a single free body about its own COM with principal axes = body axes  =>  M is diagonal
  bool diagM = m->ntree > 0 && m->neq == 0 && m->nsensor == 0 && m->nmocap == 0;   // (also implies: no limit / equality rows, every block is a contact)
  for (int t = 0; t < m->ntree && diagM; t++) {
    const int b = m->tree_bodyid[t];
    diagM = m->tree_dofnum[t] == 6 && m->body_jntnum[b] == 1 && m->jnt_type[m->body_jntadr[b]] == MJH_JNT_FREE && subtreesize[b] == 1 &&
            m->body_ipos[3*b] == 0 && m->body_ipos[3*b+1] == 0 && m->body_ipos[3*b+2] == 0 && m->body_iquat[4*b] == 1;
  }
  bool has_damping = false, has_limits = false;
  for (int d = 0; d < nv; d++) { if (m->dof_frictionloss[d] > 0) { fl_dof.push_back(d); diagM = false; } if (m->dof_damping[d] > 0) has_damping = true; }
  for (int j = 0; j < nj; j++) if (m->jnt_limited[j]) has_limits = true;
  for (int b = 1; b < nb; b++) if (m->body_gravcomp[b] != 0) gc_body.push_back(b);

  // ---- pack tables
  auto addI = [&](const int* p, size_t n) { int o = (int)I.size(); I.insert(I.end(), p, p + n); while (I.size() % 4) I.push_back(0); return o; };
  auto addF = [&](const double* p, size_t n) { int o = (int)F.size(); for (size_t i = 0; i < n; i++) F.push_back((float)p[i]); while (F.size() % 4) F.push_back(0); return o; };
#define PI(name, n) M.o_##name = addI(m->name, (size_t)(n))
#define PF(name, n) M.o_##name = addF(m->name, (size_t)(n))
  PI(body_parentid, nb); PI(body_rootid, nb); PI(body_jntadr, nb); PI(body_jntnum, nb); PI(body_dofadr, nb); PI(body_dofnum, nb);
  PI(body_level, nb); M.o_body_subtreesize = addI(subtreesize.data(), nb); PI(body_treeid, nb); M.o_body_lastdof = addI(lastdof.data(), nb);
  PI(jnt_type, nj); PI(jnt_qposadr, nj); PI(jnt_dofadr, nj); PI(jnt_bodyid, nj); PI(jnt_limited, nj);
  PI(dof_bodyid, nv); PI(dof_jntid, nv); PI(dof_parentid, nv); PI(dof_Madr, nv); PI(dof_treeid, nv);
  PI(tree_dofadr, m->ntree); PI(tree_dofnum, m->ntree);
  PI(geom_type, ng); PI(geom_bodyid, ng); PI(geom_condim, ng);
  PI(pair_geom1, m->npair); PI(pair_geom2, m->npair); M.o_pair_stageadr = addI(stageadr.data(), m->npair);
  PI(eq_obj1id, m->neq); PI(eq_obj2id, m->neq); PI(eq_active, m->neq);
  M.o_fl_dof = addI(fl_dof.data(), fl_dof.size()); M.o_gc_body = addI(gc_body.data(), gc_body.size());
  M.o_controlled = hp.o_controlled = addI(controlled.data(), controlled.size());
  M.o_odom = hp.o_odom = addI(odom.data(), odom.size());
  PF(body_pos, 3*nb); PF(body_quat, 4*nb); PF(body_ipos, 3*nb); PF(body_iquat, 4*nb); PF(body_mass, nb); PF(body_inertia, 3*nb);
  PF(body_gravcomp, nb); PF(body_invweight0, 2*nb);
  PF(jnt_pos, 3*nj); PF(jnt_axis, 3*nj); PF(jnt_stiffness, nj); PF(jnt_range, 2*nj); PF(jnt_margin, nj); PF(jnt_solref, 2*nj); PF(jnt_solimp, 5*nj);
  PF(qpos0, m->nq); PF(qpos_spring, m->nq);
  PF(dof_armature, nv); PF(dof_damping, nv); PF(dof_frictionloss, nv); PF(dof_invweight0, nv); PF(dof_solref, 2*nv); PF(dof_solimp, 5*nv);
  PF(geom_pos, 3*ng); PF(geom_quat, 4*ng); PF(geom_size, 3*ng); PF(geom_rbound, ng); PF(geom_friction, 3*ng); PF(geom_solmix, ng);
  PF(geom_solref, 2*ng); PF(geom_solimp, 5*ng); PF(geom_margin, ng); PF(geom_gap, ng);
  PF(eq_data, 11*m->neq); PF(eq_solref, 2*m->neq); PF(eq_solimp, 5*m->neq);
  PI(geom_dataid, ng); PI(mesh_vertadr, m->nmesh); PI(mesh_vertnum, m->nmesh); PF(mesh_vert, 3 * (size_t)m->nmeshvert);
  PI(eq_type, m->neq);
  if (m->nsite > 0) { PI(site_bodyid, m->nsite); PF(site_pos, 3 * (size_t)m->nsite); PF(site_quat, 4 * (size_t)m->nsite); }
  if (m->nsensor > 0) { PI(sensor_type, m->nsensor); PI(sensor_objid, m->nsensor); }
  {
    std::vector<int> mocapid(nb, -1);
    if (m->body_mocapid) for (int b = 0; b < nb; b++) mocapid[b] = m->body_mocapid[b];
    M.o_body_mocapid = addI(mocapid.data(), nb);
  }
  M.nsite = m->nsite; M.nsensor = m->nsensor; M.nmocap = m->nmocap; M.has_weld = has_weld ? 1 : 0;
  // groups of up to 16 mutually independent blocks (four waves x four 16-lane rows in mjh_solve_kernel) when a tree bitmask fits
  // one 64-bit word and every block fits a 16-lane row; else 4 (same rule as the oracle's m_group_max)
  M.group_max = m->ntree <= 64 ? 16 : 4;
  for (int t = 0; t < m->ntree; t++) if (m->tree_dofnum[t] > 8) M.group_max = 4;
#undef PI
#undef PF
  M.nq = m->nq; M.nv = nv; M.nbody = nb; M.njnt = nj; M.ngeom = ng; M.neq = m->neq; M.npair = m->npair; M.nM = m->nM; M.ntree = m->ntree;
  M.maxcon = std::max(m->maxcon, 1); M.maxefc = std::max(m->maxefc, 1);
  // per-env state record: the rows a step reads (qpos, qvel, qacc_warmstart, ddq, dq) lie back to back in one record per
  // env, padded to whole 128-byte lines (S24: 28 + 4 x 24 floats = 496 B -> 4 lines; as five separately padded rows it
  // was 5 lines), and every per-env array uses the record stride (the kernel knows only (pointer, stride) pairs)
  M.nqp = M.nvp = pad32(m->nq + 4 * std::max(nv, 1));
  M.maxlevel = maxlevel; M.nfl = (int)fl_dof.size(); M.ngc = (int)gc_body.size(); M.rowW = rowW; M.nstage = nstage;
  M.has_damping = has_damping; M.has_limits = has_limits; M.diagM = diagM;
  M.has_dim4 = 0;
  for (int g = 0; g < ng; g++) if (m->geom_condim[g] == 4) M.has_dim4 = 1;
  M.has_convex = 0;   // some pair needs the generic convex narrow phase (cylinder-x, capsule-box, ellipsoid-x)
  for (int i = 0; i < m->npair; i++) {
    const int t1 = m->geom_type[m->pair_geom1[i]], t2 = m->geom_type[m->pair_geom2[i]];
    if (t2 == MJH_GEOM_MESH) M.has_convex = 1;
    if (t1 != MJH_GEOM_PLANE && (t1 == MJH_GEOM_ELLIPSOID || t2 == MJH_GEOM_ELLIPSOID || t1 == MJH_GEOM_CYLINDER || t2 == MJH_GEOM_CYLINDER ||
                                 t2 == MJH_GEOM_MESH || (t1 == MJH_GEOM_CAPSULE && t2 == MJH_GEOM_BOX))) M.has_convex = 1;
  }
  {
    // a limited joint can have both sides active only if its margins overlap (range narrower than 2 margins)
    int nlim = 0; for (int j = 0; j < nj; j++) if (m->jnt_limited[j]) nlim += (m->jnt_range[2*j+1] - m->jnt_range[2*j] <= 2 * m->jnt_margin[j]) ? 2 : 1;
    const int nfix = neqrow + (int)fl_dof.size() + nlim;
    M.maxblk = nfix + M.maxcon; M.maxbrow = nfix + 4 * M.maxcon;
  }
  M.iterations = m->opt.iterations; M.disableflags = m->opt.disableflags;
  M.timestep = (float)m->opt.timestep; M.timestep_d = m->opt.timestep; for (int k = 0; k < 3; k++) M.gravity[k] = (float)m->opt.gravity[k];
  M.tolerance = (float)m->opt.tolerance; M.impratio = (float)m->opt.impratio; M.meaninertia = (float)m->meaninertia;
  M.noslip_iterations = m->opt.noslip_iterations; M.noslip_tolerance = (float)m->opt.noslip_tolerance;
  // ---- LDS layout (float offsets, 16-byte aligned)
  {
    Lay& L = hp.L; int off = 0; long long goff = 0;
    auto put = [&](int n) { int o = off; off += ((std::max(n, 1) + 3) / 4) * 4; return o; };
    // many-body models (nv > 64, NROW = 8 kernels): the contact / block / Jacobian pools go to a per-env slice of global
    // memory (negative offsets), only the per-body / per-dof arrays stay in LDS
    const bool big = force_big || nv > 64;
    M.big = big;
    // models with sensors read the position-stage arrays, the contact records and the velocity-stage vectors again AFTER the
    // solver (mj_sensorAcc): nothing of those is aliased then
    const bool keep = m->nsensor > 0;
    auto gput = [&](long long n) { long long o = goff; goff += ((std::max<long long>(n, 1) + 3) / 4) * 4; return (int)(-1 - o); };
    const int nblkcap = std::max(M.maxblk, 1);   // exact: the block builder never creates more than maxblk blocks
    // J / B pools: one row per non-contact block (equality, friction loss, limits), four interleaved rows per contact
    long long jsz = std::max<long long>((long long)(M.maxblk - M.maxcon) * rowW + (long long)M.maxcon * rowW * 4, 4);
    const long long need = (long long)nstage * RAW_STRIDE;            // raw-contact staging aliases J (+B)
    if ((diagM ? 1 : 2) * jsz < need) jsz = diagM ? need : (need + 1) / 2;
    // contact-patch sweep (patch_pgs.h): free bodies only, no noslip pass, pools in LDS, at most 64 contacts (the patch builder
    // keeps one contact per lane).  Same rule as the oracle's patch order (oracle/mjh_oracle.c: m_patch_order).
    // Gauss-Seidel order (mjh_set_pgs_row_order).  1 (default): mj_solPGS's own constraint-row order, blocks without a common kinematic
    // tree side by side under a precedence-preserving list schedule (bit-identical to the sequential sweep: 2).  0: the legacy orders
    // that reorder conflicting blocks (patch / pair / group first fit) — except for the models whose sweeps are sequential in either
    // layout anyway (more than 32 dofs and a kinematic tree of more than 16: articulated robots), which always run row order;
    // same rules as the oracle's (oracle/mjh_oracle.c: m_row_order)
    M.pgs_row_order = g_pgs_row_order;
    if (!M.pgs_row_order && nv > 32) for (int t = 0; t < m->ntree; t++) if (m->tree_dofnum[t] > 16) M.pgs_row_order = 1;
    const bool small_free = allow_patch && diagM && !big && nv <= 32 && M.noslip_iterations == 0 && !keep;
    const bool patch = small_free && M.maxcon <= 64;
    M.patch = patch ? 1 : 0;
    // window sweep (window_pgs.h): mjh_step of a small free-body model in row order = assemble launch + mjh_window_kernel (MJH_WINDOW=0 /
    // mjh_set_window_solver(0): the fused kernel's sweep instead — same order, same iterates up to fp32 rounding).  The window chain
    // has no tie to one contact per lane: models of this class with a contact capacity of 65 .. MJH_WINDOW_MAXCON keep the LDS-resident
    // layout WITHOUT the patch sweep (the fused kernel's dual-block sweep, list-scheduled in the same row order, serves the launches
    // that are not whole steps) and step through the window chain (mj_collision itself has no cap: mj_main.cpp:83)
    M.window = (small_free && M.maxcon <= MJH_WINDOW_MAXCON && M.pgs_row_order != 0 && g_window_solver && nv <= 32 && m->njnt <= 16 && m->nq <= 40) ? 1 : 0;
    M.win_nvt = nv <= 24 ? 24 : 32;
    M.win_maxw = std::max(1, std::min(WN_MAXW, (M.maxefc + 15) / 16));
    // (tests: MJH_WINDOW_MAXW lowers a model's window capacity below its row capacity, so that the hand-over's own capacity rule — whole blocks
    //  dropped behind the last one that fits, window_emit — can be met with a few dozen rows instead of more than 16 WN_MAXW = 384)
    if (const char* v = getenv("MJH_WINDOW_MAXW")) M.win_maxw = std::max(1, std::min(M.win_maxw, atoi(v)));
    L.qpos = put(m->nq);
    L.qvel = put(nv); L.qvref = put(nv); L.ws = put(nv); L.qacc = put(nv); L.smooth = put(nv); L.asmooth = put(nv); L.passive = put(nv);
    L.bias = put(nv); L.applied = put(nv); L.tmpv = put(nv); L.tmpv2 = put(nv);
    if (patch) {   // what the sweep still needs sits in front of the span the patch pool takes over
      L.qM = put(m->nM); L.qLD = L.qM; L.qLDinv = put(nv); L.dofpar = 0; L.dofMadr = 0; L.anc = 0;
      L.p_gsize = put(3*ng); L.p_rbound = put(ng); L.p_mass = put(nb); L.p_inertia = put(3*nb);
      L.zero = put(PP_ZERO);     // (a lane outside a patch reads a whole row record of zeros)
    }
    // K1: frames, composite inertias, joint anchors/axes and geom poses are dead once the position stage, CRBA and the
    // collision stage are done; the solver's per-base scratch vectors (bv, phi: first used by the velocity stage) reuse them
    const int k1 = off;
    // free-body models (diagM) form neither composite / spatial inertias nor motion axes nor the spatial velocity-stage vectors
    // (closed forms in step_kernel.h): those arrays all alias one unused slot
    const int unused = diagM ? put(4) : 0;
    L.xpos = put(3*nb); L.xquat = put(4*nb); L.xmat = put(9*nb); L.ximat = put(9*nb); L.crb = diagM ? unused : put(10*nb);
    L.xanchor = put(3*nj); L.xaxis = put(3*nj); L.gpos = put(3*ng); L.gmat = put(9*ng);
    const int k1_size = off - k1;
    M.k1_floats = k1_size;
    M.scratch_off = (big && keep) ? put(k1_size) : k1;
    L.xipos = put(3*nb); L.com = put(3*nb); L.cinert = diagM ? unused : put(10*nb); L.cdof = diagM ? unused : put(6*nv);
    const int k2_size = off - k1;   // ... and all of these are dead when the solver sweeps run: the X extension of condim-4 models
    if (!patch) {
      // diagonal M: no factor storage; many-body layout: the factor is built in M's place (M itself stays in the env's scratch slice)
      L.qM = put(m->nM); L.qLD = (diagM || big) ? L.qM : put(m->nM); L.qLDinv = put(nv);
      if (diagM) { L.dofpar = 0; L.dofMadr = 0; L.anc = 0; }   // free-body models read the shared chain-walk tables (step_kernel.h)
      else { L.dofpar = put(nv); L.dofMadr = put(nv); L.anc = put(m->nM); }
      // (models without a collision pair never read the per-env geom sizes / bounding radii: their copies land in the geom poses' span instead of
      //  taking their own — C3, four arms per wavefront: 20816 -> 20336 B, 17 -> 16 LDS granules, 7 -> 8 workgroups per CU = all 2048 resident)
      const bool nopairs = m->npair == 0;
      L.p_gsize = nopairs ? L.gpos : put(3*ng); L.p_rbound = nopairs ? L.gmat : put(ng); L.p_mass = put(nb); L.p_inertia = put(3*nb);
    }
    {  // the contact records die once the blocks are built; the velocity-stage spatial vectors reuse their space
      const int a4 = [](int n) { return ((std::max(n, 1) + 3) / 4) * 4; }(6*nb);
      const int velsz = diagM ? 4 : 4 * a4 + ((6*nv + 3) / 4) * 4;
      int vel;
      // (many-body layout: they reuse the position-stage arrays instead, which are dead when the velocity stage starts)
      if (big) { L.con = gput((long long)M.maxcon * CON_STRIDE); L.blkq = gput((long long)nblkcap * BLKQ_STRIDE); vel = (!keep && !diagM && velsz <= k1_size) ? k1 : put(velsz); }
      else if (keep) { L.con = put(M.maxcon * CON_STRIDE); L.blkq = put(nblkcap * BLKQ_STRIDE); vel = put(velsz); }
      else {
        // (and, once those are dead too, the per-block solver matrices A_c / Q, built when the solver starts)
        L.con = put(std::max(std::max(M.maxcon * CON_STRIDE, velsz), nblkcap * BLKQ_STRIDE));
        L.blkq = L.con; vel = L.con;
      }
      if (diagM) { L.cvel = unused; L.cacc = unused; L.cfrc = unused; L.cfrcsub = unused; L.cdofdot = unused; }
      else { L.cvel = vel; L.cacc = vel + a4; L.cfrc = vel + 2*a4; L.cfrcsub = vel + 3*a4; L.cdofdot = vel + 4*a4; }
    }
    const int extsz = (M.has_dim4 && !patch) ? nblkcap * SOLX_N : 0;   // (the patch sweep does not use the condim-4 extension)
    if (big) {
      L.blki = gput((long long)nblkcap * BLKI_STRIDE); L.blkf = gput((long long)nblkcap * BLKF_STRIDE);
      L.bv = gput((long long)nblkcap * 4); L.phi = gput((long long)nblkcap * 4); L.sched = gput((long long)nblkcap * 2); L.order = gput(nblkcap);
      L.ext = gput(extsz); L.J = gput(jsz); L.B = diagM ? L.J : gput(jsz);
    } else {
      L.blki = put(nblkcap * BLKI_STRIDE); L.blkf = put(nblkcap * BLKF_STRIDE);
      // window-only models (65 .. 128 contacts): their assemble-only launch (WPRE instance 2) builds no block schedule and no condim-4
      // extension — both belong to the fused kernel's sweep — and keeps the per-base scratch vectors bv / phi (first written by the
      // velocity stage) in the contact records, which are dead by then (step_kernel.h; the fused instance cannot: its A_c / Q matrices
      // take that span while phi is live).  Its LDS ends in front of all of them: S24D at capacity 96 25.3 -> 16.9 KB, 6 -> 9
      // environments per CU (5.30 -> 5.6 M env-steps/s, bitwise).  MJH_WPRE_SLIM2=0: the former extent
      static const bool slim2 = !(getenv("MJH_WPRE_SLIM2") && atoi(getenv("MJH_WPRE_SLIM2")) == 0);
      const bool wonly = M.window && !patch && slim2 && 8 * nblkcap <= M.maxcon * CON_STRIDE;
      if (wonly) hp.lds_bytes_pre = off * (int)sizeof(float);
      if (2 * nblkcap * 4 <= k1_size && !keep) { L.bv = k1; L.phi = k1 + nblkcap * 4; }
      else { L.bv = put(nblkcap * 4); L.phi = put(nblkcap * 4); }
      L.sched = put(nblkcap * 2);
      // the dual-block sweep (nv <= 32) reads the pair schedule only; `order` is then just scratch of the schedule builder
      L.order = (nv <= 32 && nblkcap <= k1_size && !keep) ? L.bv : put(nblkcap);
      L.ext = (extsz <= k2_size && !keep) ? k1 : put(extsz);
      // (window models beyond 64 contacts: the base-row pool comes last, their assemble-only launch keeps it in global memory and allocates the LDS in front of it)
      if (!wonly) hp.lds_bytes_pre = off * (int)sizeof(float);
      L.J = put((int)jsz); L.B = diagM ? L.J : put((int)jsz);
    }
    M.win_jsz = (int)jsz;
    if (!(M.window && !big && !patch)) hp.lds_bytes_pre = off * (int)sizeof(float);    // everything but the patch pool's own tail: what an assemble-only launch (window chain) touches
    if (patch) {
      // the pool: per patch of nr4 rows (a multiple of 4, at most 16) a record per row (20 floats between two bodies, 12 on one
      // body) + 16 floats per 4x4 tile of the lower triangle of AR: 16 .. 30 floats per row.  It takes the span of everything that
      // is dead by then and no more (S24: 3948 floats; a 20 000-step soak of 4096 envs never fills it); patches beyond it are
      // dropped with the capacity flag, like contacts beyond maxcon
      // The two small tables the sweep reads next to the pool (one descriptor per patch, one per schedule slot) sit in front
      // of it: that space (position-stage arrays, contact records) is dead when they are written, unlike the span's tail.
      M.pdesc = k1; M.pslot = k1 + ((M.maxcon + 3) / 4) * 4; M.pool = M.pslot + 4 * M.maxcon;
      M.pool_floats = std::max(off - M.pool, (33 * M.maxefc) / 2);   // (at least what the span was before free-body models dropped their spatial arrays: a 20 000-step soak never filled that)
      if (const char* cap = getenv("MJH_PATCH_POOL_FLOATS")) M.pool_floats = std::max(512, std::min(M.pool_floats, atoi(cap)));   // (tests: the drop rule)
      off = std::max(off, M.pool + M.pool_floats);
      if (!getenv("MJH_PATCH_POOL_FLOATS")) {
        // LDS is handed out in 1280-byte granules and a CU holds floor(128 / granules) workgroups: the pool takes what is left of
        // the last granule that costs no workgroup (S24: 15 -> 16 granules at 8 per CU, +12 % pool)
        const int gran = (off * 4 + 1279) / 1280, wg = 128 / std::max(gran, 1);
        if (wg >= 1) { const int room = std::min((128 / wg) * 320, 16384); if (room > off) { M.pool_floats += room - off; off = room; } }
      }
    } else L.zero = put(4);
    L.site = m->nsite > 0 ? put(12 * m->nsite) : 0;        // world frame of every site: pos(3) + rotation(9)
    L.fext = m->nsensor > 0 ? put(6 * nb) : 0;             // external spatial force per body (mj_rnePostConstraint)
    // (the site frames / external forces sit behind the patch pool: an assemble-only launch of a model that has them gets the whole layout)
    if (m->nsite > 0 || m->nsensor > 0) hp.lds_bytes_pre = off * (int)sizeof(float);
    if (big) {   // hand-over vectors of the three-launch step (non-negative offsets into the scratch slice)
      auto graw = [&](int n) { long long o = goff; goff += ((std::max(n, 1) + 3) / 4) * 4; return (int)o; };
      L.g_a0 = graw(nv); L.g_minv = graw(nv); L.g_qvel = graw(nv); L.g_smooth = graw(nv); L.g_qacc = graw(nv); L.g_meta = graw(8); L.g_qM = graw(m->nM);
      // dense row-space solver (dense_pgs.h): articulated models (M not diagonal) without noslip sweeps, at most 128 dofs; an
      // env takes it in the steps in which it has at most dense_cap rows.  MJH_DENSE=0 keeps the block solver everywhere.
      const bool dense_on = !(getenv("MJH_DENSE") && atoi(getenv("MJH_DENSE")) == 0);
      M.dense = (dense_on && !diagM && M.noslip_iterations == 0 && nv <= 128 && nv >= 1) ? 1 : 0;
      M.dense_cap = std::min(256, ((std::max(M.maxefc, 1) + 63) / 64) * 64); M.dense_nvs = ((nv + 15) / 16) * 16;
      if (const char* dc = getenv("MJH_DENSE_CAP")) M.dense_cap = std::max(64, std::min(M.dense_cap, (atoi(dc) / 64) * 64));   // (tests: envs beyond the capacity keep the block solver)
      M.dense_min_iter = getenv("MJH_DENSE_MIN_ITER") ? std::max(0, atoi(getenv("MJH_DENSE_MIN_ITER"))) : 32;
      L.g_dense = 0; L.g_qLD = 0; L.g_anc = 0;
      if (M.dense) { L.g_qLD = graw(m->nM); L.g_anc = graw(m->nM); }
      if (M.dense) { long long o = goff; goff += (long long)M.dense_cap * M.dense_cap + (long long)M.dense_cap * M.dense_nvs + 6LL * M.dense_cap; L.g_dense = (int)o; }
    }
    hp.gstride = goff;
    if (const char* pad = getenv("MJH_LDS_PAD")) off += std::max(0, atoi(pad));   // (occupancy experiments: floats of unused LDS per env)
    L.total = off;
    hp.lds_bytes = off * (int)sizeof(float);
  }
}

// Layout choice.  The LDS-resident layout wins for small free-body scenes (S24: 4.7 M env-steps/s against 1.6 M), the
// many-body layout (pools in global memory, three-launch step) wins as soon as the LDS-resident working set leaves only a
// few environments per CU (pr2 / tiago / hsrb4s / ridgeback_panda: 1.6x - 5x); a working set beyond one CU's LDS has no choice.
//   policy 0 (default): many-body layout above MJH_LDS_RESIDENT_MAX bytes;  1: LDS-resident whenever it fits;  2: many-body whenever possible
#define MJH_LDS_RESIDENT_MAX (24 * 1024)
// steps between renewals of a cohort's launch order.  The sort is one workgroup on the cohort's stream in front of a step: 6 us on an idle
// chip, 74 us (S24) to 200 us (C2) beside the other cohorts' resident waves; an env's cost drifts slowly.  8 / 16 / 32 / 64 steps: S24
// 11.29 / 11.46 / 11.52 / 11.40 M, C4 1.41 / 1.43 / 1.45 / 1.39 M, C2 0.541 / 0.545 / 0.546 / 0.544 M env-steps/s (tools/r04_order_every.sh)
#define MJH_ORDER_EVERY 32
// ... 16 for the window models whose envs reach the 64-row form's row counts (S24D: the slowest wavefront is a function of which envs share it,
// and the piles' row counts drift faster than the sweep counts of S24): 8 / 16 / 32 / 64 steps S24D 5.92 / 5.91 / 5.84 / 5.73 M, S24 13.60 /
// 13.75 / 13.82 / 13.75 M (round 6, tools/r06_knobs.sh).  Results do not depend on the launch order (bitwise: tests).
static int order_every_of(const mjh_engine* e);
static int g_layout_policy = 0;
extern "C" void mjh_set_layout_policy(int policy) { g_layout_policy = policy < 0 || policy > 2 ? 0 : policy; }
static int order_every_of(const mjh_engine* e) {
  static const int env = getenv("MJH_ORDER_EVERY") ? std::max(1, atoi(getenv("MJH_ORDER_EVERY"))) : 0;
  return env ? env : ((e->M.window && e->M.win_maxw > 16) ? 16 : MJH_ORDER_EVERY);
}
static void derive_fitting(const mjh_model* m, HostPack& hp) {
  derive_device_model(m, hp);
  int policy = g_layout_policy;
  if (const char* fb = getenv("MJH_FORCE_BIG")) policy = atoi(fb) ? 2 : 1;
  const int limit = policy == 1 ? 160 * 1024 : (policy == 2 ? 0 : MJH_LDS_RESIDENT_MAX);
  // (models on the contact-patch sweep stay LDS-resident under the default policy: their visiting order is the patch order,
  //  which the many-body layout does not have; mjh_solver_order() tells which one an engine runs)
  if ((hp.M.patch || hp.M.window) && policy == 0 && hp.lds_bytes <= 64 * 1024) return;
  // The patch sweep packs two LDS byte addresses into one 32-bit word (patch_pgs.h: laddr(x) | laddr(y) << 16), which holds
  // only while the workgroup's LDS stays within 64 KiB.  A patch model that would stay LDS-resident beyond that (policy 1 /
  // MJH_FORCE_BIG=0 with a large user-set maxefc) is re-derived WITHOUT the patch sweep (pair order, mjh_solver_order() = 0).
  if (hp.M.patch && hp.lds_bytes > 64 * 1024 && hp.lds_bytes <= limit) { HostPack np; derive_device_model(m, np, false, false); hp = np; }
  if ((hp.lds_bytes > 160 * 1024 || hp.lds_bytes > limit) && !hp.M.big && hp.M.rowW <= 64) { HostPack big; derive_device_model(m, big, true); hp = big; }
}

extern "C" int mjh_query_lds_bytes(const mjh_model* m) {
  if (!m) return MJH_ERR_ARG;
  HostPack hp; derive_fitting(m, hp);
  return hp.lds_bytes;
}
// the LDS layout of a model as text: "name offset" per array in float units (negative: the env's global slice), then the totals —
// capacity planning (which array costs a granule: tools/lds_layout.py).  Returns the number of bytes written (without the terminator)
extern "C" int mjh_debug_lds_layout(const mjh_model* m, char* out, int cap) {
  if (!m || !out || cap <= 0) return MJH_ERR_ARG;
  HostPack hp; derive_fitting(m, hp);
  std::string t;
#define X(n) t += std::string(#n) + " " + std::to_string(hp.L.n) + "\n";
  MJH_LDS_ARRAYS(X)
#undef X
  t += "total " + std::to_string(hp.L.total) + "\nlds_bytes " + std::to_string(hp.lds_bytes) + "\nlds_bytes_pre " + std::to_string(hp.lds_bytes_pre) +
       "\nk1_floats " + std::to_string(hp.M.k1_floats) + "\nmaxcon " + std::to_string(hp.M.maxcon) + "\nmaxblk " + std::to_string(hp.M.maxblk) +
       "\nrowW " + std::to_string(hp.M.rowW) + "\nnstage " + std::to_string(hp.M.nstage) + "\nbig " + std::to_string((int)hp.M.big) + "\n";
  const int n = std::min((int)t.size(), cap - 1);
  std::memcpy(out, t.data(), (size_t)n); out[n] = 0;
  return n;
}
extern "C" int mjh_query_lds_bytes_assemble(const mjh_model* m) {   // ... of the assemble-only instance of the window chain (0: the model does not take it)
  if (!m) return MJH_ERR_ARG;
  HostPack hp; derive_fitting(m, hp);
  return hp.M.window ? hp.lds_bytes_pre : 0;
}

extern "C" int mjh_create(const mjh_model* m, int nenv, int device, void* stream, mjh_engine** out) {
  if (!m || nenv <= 0 || !out) { mjh_set_error("mjh_create: bad argument"); return MJH_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    mjh_set_error("mjh_create: no HIP device visible (this engine has no CPU fallback)");
    return MJH_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) { mjh_set_error("mjh_create: bad device index"); return MJH_ERR_ARG; }

  if (m->ngeom > 4095) { mjh_set_error("mjh_create: more than 4095 geoms (contact records pack the geom ids in 12 bits)"); return MJH_ERR_CAPACITY; }
  for (int g = 0; g < m->ngeom; g++) if (m->geom_condim[g] != 1 && m->geom_condim[g] != 3 && m->geom_condim[g] != 4) {
    mjh_set_error("mjh_create: condim must be 1, 3 or 4 (rolling friction, condim 6, is not implemented)"); return MJH_ERR_UNSUPPORTED; }
  HIPCHK(hipSetDevice(device));
  mjh_engine* e = new mjh_engine();
  // from here on every failure path releases the engine and whatever it has allocated so far
#undef HIPCHK
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { mjh_set_error(std::string(#call) + ": " + hipGetErrorString(e_)); mjh_destroy(e); return MJH_ERR_NO_DEVICE; } } while (0)
  if (const char* v = getenv("MJH_LPT")) e->lpt = atoi(v) != 0;
  if (const char* v = getenv("MJH_SPLIT3")) e->split3 = atoi(v) != 0;
  e->model = m; e->nenv = nenv; e->device = device; e->stream = (hipStream_t)stream;

  HostPack hp; derive_fitting(m, hp);
  e->M = hp.M; e->L = hp.L; e->lds_bytes = hp.lds_bytes; e->lds_bytes_pre = hp.lds_bytes_pre; e->o_controlled = hp.o_controlled; e->o_odom = hp.o_odom;
  DModel& M = e->M; std::vector<int>& I = hp.I; std::vector<float>& F = hp.F;
  e->hI = I;
  if (dev_alloc(e, &e->dI, I.size(), false) || dev_alloc(e, &e->dF, F.size(), false)) { mjh_destroy(e); return MJH_ERR_NO_DEVICE; }
  HIPCHK(hipMemcpyAsync(e->dI, I.data(), I.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->dF, F.data(), F.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  M.I = e->dI; M.F = e->dF;

  {
    DConst hc; hc.M = e->M; hc.L = e->L;
    if (dev_alloc(e, &e->dC, 1, false)) { mjh_destroy(e); return MJH_ERR_NO_DEVICE; }
    HIPCHK(hipMemcpyAsync(e->dC, &hc, sizeof hc, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  if ((long long)(e->M.maxblk - e->M.maxcon) * e->M.rowW + (long long)e->M.maxcon * e->M.rowW * 4 >= 4LL * 65536) {   // hd.x >> 16 holds the offset in float4 units
    mjh_set_error("mjh_create: Jacobian pool too large for the 16-bit block offsets (lower the contact capacity)");
    mjh_destroy(e); return MJH_ERR_CAPACITY;
  }
  if (e->M.big && e->M.rowW > 64) {
    mjh_set_error("mjh_create: nv > 64 needs every constraint to touch at most 64 dofs (this model: " + std::to_string(e->M.rowW) + "); the solver maps one dof of a block per lane");
    mjh_destroy(e); return MJH_ERR_CAPACITY;
  }
  if (e->lds_bytes > 160 * 1024) {
    mjh_set_error("mjh_create: per-env working set exceeds the 160 KiB LDS of one CU (" + std::to_string(e->lds_bytes) + " B)");
    mjh_destroy(e); return MJH_ERR_CAPACITY;
  }
#define MJH_ATTR(NR, DG) do { HIPCHK(hipFuncSetAttribute((const void*)mjh_step_kernel<NR, DG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes)); \
                              HIPCHK(hipFuncSetAttribute((const void*)mjh_step_kernel<NR, DG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes)); } while (0)
  if (e->M.dense) {
    // LDS of mjh_dense_build_kernel: 1 / D [nvs] | 1 / AR_qq [cap] | row table [cap] int4 | row starts [maxblk + 1]
    e->dense_lds = ((size_t)e->M.dense_nvs + 5 * (size_t)e->M.dense_cap + (size_t)std::max(e->M.maxblk, 1) + 8) * sizeof(float);
    // ... and of mjh_dense_solve_kernel: start values [cap] | 128 | the factor twice [2 nM] (a 128-dof chain: 67.6 KB, beyond the 64 KB a
    // launch gets without the attribute)
    e->dense_solve_lds = (DN_CAP_MAX + 128 + 2 * (size_t)e->M.nM) * sizeof(float);
    if (e->dense_lds > 160 * 1024 || e->dense_solve_lds > 160 * 1024) { e->M.dense = 0; }
    else {
      HIPCHK(mjh_dense_attributes(e->dense_lds, e->dense_solve_lds));
    }
  }
  MJH_ATTR(1, true); MJH_ATTR(2, true); MJH_ATTR(4, true); MJH_ATTR(8, true); MJH_ATTR(1, false); MJH_ATTR(2, false); MJH_ATTR(4, false); MJH_ATTR(8, false);
#undef MJH_ATTR
  if (e->M.window) {
#define MJH_ATTRW(NR, CX, GJ) HIPCHK(hipFuncSetAttribute((const void*)mjh_step_kernel<NR, true, CX, GJ>, hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes))
    MJH_ATTRW(1, false, 1); MJH_ATTRW(1, true, 1); MJH_ATTRW(2, false, 1); MJH_ATTRW(2, true, 1);
    MJH_ATTRW(1, false, 2); MJH_ATTRW(1, true, 2); MJH_ATTRW(2, false, 2); MJH_ATTRW(2, true, 2);
#undef MJH_ATTRW
  }

  // ---- per-env state
  DState& S = e->S;
  const size_t nq_all = (size_t)nenv * M.nqp, nv_all = (size_t)nenv * M.nvp;
  int rc = 0;
  rc |= dev_alloc(e, &S.qpos, nq_all); rc |= dev_alloc(e, &S.initial_qpos, nq_all);
  {   // columns of the per-env record (derive_device_model): qpos | qvel | qacc_warmstart | ddq | dq
    const int nvc = std::max(M.nv, 1);
    S.qvel = S.qpos + M.nq; S.qacc_ws = S.qvel + nvc; S.ddq = S.qacc_ws + nvc; S.dq = S.ddq + nvc;
  }
  S.qacc = S.qacc_ws;   // one array: after every solve qacc_warmstart = qacc (step_kernel.h, store), so the second row would only double the traffic
  rc |= dev_alloc(e, &S.qvel_ref, nv_all); rc |= dev_alloc(e, &S.qfrc_applied, nv_all);
  rc |= dev_alloc(e, &S.qfrc_inverse, nv_all);
  S.gscratch = nullptr; S.gstride = hp.gstride;
  if (M.big) rc |= dev_alloc(e, &S.gscratch, (size_t)nenv * (size_t)hp.gstride, false);   // many-body models: contact / block / Jacobian pools
  S.wbuf = nullptr; S.wstride = 0;
  // (0: off; experiments: another row threshold.  Models whose rows can exceed 256 — S24D — keep every env below the 64-row threshold in the 16-row form:
  //  four envs per wavefront; the two-env wavefronts of the 32-row form cost the SIMD time the one-env wavefronts of the slowest envs need — round 6)
  S.win32 = getenv("MJH_WINDOW32") ? std::max(0, atoi(getenv("MJH_WINDOW32"))) : (M.win_maxw > 16 ? 0 : WN32_MIN_ROWS);
  // 64-row form above ... rows: models whose envs stay within 256 rows (S24: 9 % of the envs beyond 96 rows) give it every env beyond the 16-row
  // form's register-resident windows — 2 x 64 rows with the chains' wait states filled beat 4 x 32 (S24 12.3 -> 13.3 M); models with more rows
  // (S24D: 30 % of the envs between 97 and 128 rows, 60 % beyond) only the envs a cohort's step waits for (176 / 192 / 208 rows: 4.65 / 4.97 / 5.26 M)
  S.win64 = getenv("MJH_WINDOW64") ? std::max(0, atoi(getenv("MJH_WINDOW64"))) : (M.win_maxw > 16 ? WN64_MIN_ROWS : WN32_MIN_ROWS);   // (0: off)
  if (S.win64 > 0 && S.win64 < S.win32) S.win64 = S.win32;
  if (getenv("MJH_WN_NL") && atoi(getenv("MJH_WN_NL")) < 3) S.win64 = 0;      // (experiments that take the LDS tier away: the 64-row section keeps two 16 KB tiles there)
  if (M.window) {   // window sweep: header + vectors + win_maxw windows of rows + tiles of the streamed windows, per env
    S.wj_off = ((WN_ROWS + M.win_maxw * WN_NK * 16 + M.win_maxw * WN_XREC(M.win_nvt) * 16 + (M.win_maxw / 2 + 1) * WN_XPAIR * 16 + 63) / 64) * 64;
    S.wstride = S.wj_off + ((M.win_jsz + 63) / 64) * 64;
    if (((S.wstride / 64) & 1) == 0) S.wstride += 64;      // an odd number of 256-byte lines per env: the same offset of consecutive envs' slices does not fall on the same memory channel
    rc |= dev_alloc(e, &S.wbuf, (size_t)nenv * (size_t)S.wstride, true);
  }
  if (M.window && e->lpt && nenv >= 1024) {
    HIPCHK(hipHostMalloc((void**)&e->h_wn, MJH_MAX_COHORTS * sizeof(int), hipHostMallocMapped));
    for (int g = 0; g < MJH_MAX_COHORTS; g++) e->h_wn[g] = 1 << 20;        // (nothing seen yet: keep the LDS tier)
    HIPCHK(hipHostGetDevicePointer((void**)&e->d_wn, e->h_wn, 0));
  }
  if (M.big && M.dense && e->lpt && nenv >= 1024) {
    // one word per cohort in host-mapped memory: "a env of the cohort swept long when its launch order was last rebuilt" (mjh_order_kernel)
    HIPCHK(hipHostMalloc((void**)&e->h_dense, MJH_MAX_COHORTS * 4 * sizeof(int), hipHostMallocMapped));
    for (int g = 0; g < MJH_MAX_COHORTS * 4; g++) e->h_dense[g] = 1;
    for (int g = 0; g < MJH_MAX_COHORTS; g++) e->dense_now[g] = true;
    HIPCHK(hipHostGetDevicePointer((void**)&e->d_dense, e->h_dense, 0));
    for (int g = 0; g < MJH_MAX_COHORTS; g++) for (int k = 0; k < 4; k++) HIPCHK(hipEventCreateWithFlags(&e->ev_dense[g][k], hipEventDisableTiming));
  }
  rc |= dev_alloc(e, &S.time, (size_t)nenv); rc |= dev_alloc(e, &S.odom_vel, (size_t)nenv * 6); rc |= dev_alloc(e, &S.stats, (size_t)nenv * 4);
  rc |= dev_alloc(e, &S.x_bias, nv_all); rc |= dev_alloc(e, &S.x_passive, nv_all); rc |= dev_alloc(e, &S.x_smooth, nv_all);
  rc |= dev_alloc(e, &S.x_constraint, nv_all); rc |= dev_alloc(e, &S.x_energy, (size_t)nenv * 2);
  if (M.nsensor > 0) { rc |= dev_alloc(e, &S.sensordata, (size_t)nenv * 3 * M.nsensor); e->split3 = false; }   // (sensors read the position stage after the solve: one fused launch)
  if (M.nmocap > 0) rc |= dev_alloc(e, &S.mocap, (size_t)nenv * 7 * M.nmocap, false);
  if (rc) { mjh_destroy(e); return MJH_ERR_NO_DEVICE; }
  if (M.nmocap > 0) {   // mocap poses start at the bodies' model poses (d->mocap_pos / mocap_quat after mj_makeData)
    std::vector<float> mp((size_t)nenv * 7 * M.nmocap);
    for (int b = 0; b < m->nbody; b++) if (m->body_mocapid && m->body_mocapid[b] >= 0)
      for (int en = 0; en < nenv; en++) {
        float* p = &mp[((size_t)en * M.nmocap + m->body_mocapid[b]) * 7];
        for (int k = 0; k < 3; k++) p[k] = (float)m->body_pos[3*b+k];
        for (int k = 0; k < 4; k++) p[3+k] = (float)m->body_quat[4*b+k];
      }
    HIPCHK(hipMemcpyAsync(S.mocap, mp.data(), mp.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  // initial state = qpos0 for every env
  {
    std::vector<float> q0(nq_all, 0.0f);
    for (int en = 0; en < nenv; en++) for (int i = 0; i < m->nq; i++) q0[(size_t)en * M.nqp + i] = (float)m->qpos0[i];
    HIPCHK(hipMemcpyAsync(S.qpos, q0.data(), nq_all * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(S.initial_qpos, q0.data(), nq_all * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  // Cohorts: 2 recover ~95% of the slot-limited throughput of the fused step, a third one adds 0 - 9 % there (S24 +1 %, C3 +9 %:
  // not the default, a launch then covers a third of the envs for the same duration);
  // the many-body layout's step is three launches (assemble -> solve -> integrate) and three cohorts keep all three busy (C2 +4 %,
  // C4 +11 %).  Four or more lose (C3 -30 %): streams start to share hardware queues.  Three cohort streams + the caller's stream +
  // the export stream need more than the runtime's default of 4 hardware queues: see mjh_library_init below.
  { int nc = nenv >= 1024 ? (e->M.big && e->split3 && nenv >= 1536 ? 3 : 2) : 1;   // (fused step: two by default, `mjh_set_cohorts(e, 3)` / MJH_COHORTS=3 for the last per cent)
    if (const char* v = getenv("MJH_COHORTS")) nc = atoi(v); if (set_cohorts(e, nc)) { mjh_destroy(e); return MJH_ERR_NO_DEVICE; } }
  *out = e;
  return MJH_OK;
#undef HIPCHK
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { mjh_set_error(std::string(#call) + ": " + hipGetErrorString(e_)); return MJH_ERR_NO_DEVICE; } } while (0)
}

extern "C" void mjh_destroy(mjh_engine* e) {
  if (!e) return;
  (void)join_cohorts(e);
  (void)hipStreamSynchronize(e->stream);
  for (int g = 0; g < MJH_MAX_COHORTS; g++) { if (e->cstream[g]) (void)hipStreamDestroy(e->cstream[g]); if (e->ev_join[g]) (void)hipEventDestroy(e->ev_join[g]); }
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  for (int g = 0; g < MJH_MAX_COHORTS; g++) for (int k = 0; k < 8; k++) if (e->cgraph[g][k].exec) (void)hipGraphExecDestroy(e->cgraph[g][k].exec);
  for (auto& p : e->tev) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->scratch) (void)hipFree(e->scratch);
  if (e->h_dense) (void)hipHostFree(e->h_dense);
  if (e->h_wn) (void)hipHostFree(e->h_wn);
  if (e->h_io) (void)hipHostFree(e->h_io);
  for (int k = 0; k < MJH_IO_SLOTS; k++) if (e->io_ev[k]) (void)hipEventDestroy(e->io_ev[k]);
  for (int g = 0; g < MJH_MAX_COHORTS; g++) for (int k = 0; k < 4; k++) if (e->ev_dense[g][k]) (void)hipEventDestroy(e->ev_dense[g][k]);
  delete e;
}

// (every entry point re-selects the engine's device: one process may hold engines on several GPUs, and the caller's
// current device is whatever its framework left it at)
static int flush_step1(mjh_engine* e);
#define ENG_NOJOIN(e) if (!(e)) { mjh_set_error("null engine"); return MJH_ERR_ARG; } \
                      if (hipSetDevice((e)->device) != hipSuccess) { mjh_set_error("hipSetDevice failed"); return MJH_ERR_NO_DEVICE; } \
                      if ((e)->step1_pending) { int rcf_ = flush_step1(e); if (rcf_) return rcf_; } \
                      const bool mjh_ho_ = (e)->handover; (e)->handover = false; (void)mjh_ho_;
// (entry points that cannot change what a pending window hand-over was built from — getters, mjh_set_cmd — put it back)
#define KEEP_HANDOVER(e) (e)->handover = mjh_ho_;
// every entry point except mjh_step first joins the cohort streams back into the caller's stream
#define ENG(e) ENG_NOJOIN(e) { int rcj_ = join_cohorts(e); if (rcj_) return rcj_; }
#define RANGE(e, env0, n) if ((env0) < 0 || (n) < 0 || (env0) + (n) > (e)->nenv) { mjh_set_error("env range out of bounds"); return MJH_ERR_ARG; }

// full-range launch on the caller's stream in longest-job-first order (split API: step1 | inverse | step2 | forward);
// `resort` rebuilds the order from the previous step's solver statistics, otherwise the last order is reused
// With cohorts (mjh_set_cohorts > 1) the range is issued cohort by cohort on the cohort streams, like mjh_step: the reference's loop
// reads and commands ONE environment between the calls (MjHWInterface::read / write, mj_main.cpp:86-106), and only that environment's
// cohort has to wait for the host (mjh_get_joint_state / mjh_set_cmd on a range inside one cohort touch that cohort's stream only).
static int launch_lpt(mjh_engine* e, int ph, int xflags, bool resort, int wmode = 0) {
  const int G = e->ncohort > 1 && e->nenv >= 64 * e->ncohort ? e->ncohort : 1;
  static const bool split_cohorts = !(getenv("MJH_SPLIT_COHORTS") && atoi(getenv("MJH_SPLIT_COHORTS")) == 0);
  if (G > 1 && split_cohorts) {
    if (e->lpt && e->nenv >= 1024 && !e->d_order) { int rc = join_cohorts(e); if (!rc) rc = dev_alloc(e, &e->d_order, (size_t)e->nenv); if (rc) return rc; e->order_valid = false; e->order_G = -1; }
    int rc = fork_cohorts(e);
    if (rc) return rc;
    StateGuard guard(&e->S);
    const bool lpt = e->lpt && e->d_order && e->nenv >= 1024;
    if (lpt) e->S.env_order = e->d_order;
    const int order_every = order_every_of(e);
    const bool sort = lpt && (e->order_G != G || !e->order_valid || (resort && e->order_age % order_every == 0));
    // (the XF_FORCE exports are engine-sized arrays indexed by the row inside the launch's range: shifted to the cohort's first row)
    float* const xb = e->S.x_bias; float* const xp = e->S.x_passive; float* const xs = e->S.x_smooth; float* const xc = e->S.x_constraint; float* const xe = e->S.x_energy;
    for (int g = 0; g < G && !rc; g++) {
      const int g0 = (int)((long long)e->nenv * g / G), g1 = (int)((long long)e->nenv * (g + 1) / G);
      // (window models: the sort also leaves the cohort's largest row count for the window kernel's LDS-tier choice, as in mjh_step —
      //  a host that only ever uses the split API would keep the tier of an engine that has seen nothing yet)
      if (sort) hipLaunchKernelGGL(mjh_order_kernel, dim3(1), dim3(1024), 0, e->cstream[g], (const int*)e->S.stats, e->d_order, g0, g1 - g0, e->d_wn ? e->d_wn + g : (int*)nullptr, e->d_wn ? -1 : 0);
      const size_t o = (size_t)g0 * e->M.nvp;
      e->S.x_bias = xb ? xb + o : nullptr; e->S.x_passive = xp ? xp + o : nullptr; e->S.x_smooth = xs ? xs + o : nullptr;
      e->S.x_constraint = xc ? xc + o : nullptr; e->S.x_energy = xe ? xe + 2 * (size_t)g0 : nullptr;
      e->cur_cohort = g; rc = launch_on(e, e->cstream[g], g0, g1 - g0, 1, ph, xflags, wmode); e->cur_cohort = -1;
    }
    if (sort) { e->order_G = G; e->order_valid = true; }
    return rc;
  }
  { int rc = join_cohorts(e); if (rc) return rc; }
  if (e->lpt && e->nenv >= 1024) {
    if (!e->d_order) { int rc = dev_alloc(e, &e->d_order, (size_t)e->nenv); if (rc) return rc; e->order_valid = false; }
    if (resort || !e->order_valid) {
      hipLaunchKernelGGL(mjh_order_kernel, dim3(1), dim3(1024), 0, e->stream, (const int*)e->S.stats, e->d_order, 0, e->nenv, (int*)nullptr, 0);
      e->order_valid = true; e->order_G = -1;   // (a full-range sort: mjh_step's cohorts must sort their own ranges again)
    }
  }
  StateGuard guard(&e->S);
  if (e->lpt && e->d_order && e->order_valid) e->S.env_order = e->d_order;
  return launch(e, 0, e->nenv, 1, ph, xflags, wmode);
}
static int launch_pd(mjh_engine* e, hipStream_t st, int env0, int n);
// mjh_step1 defers its launch to the next entry point: the reference's loop calls MjHWInterface::read() = mj_inverse right behind
// mj_step1 (mj_main.cpp:83-94), and step1 + inverse as ONE launch share the position and velocity stages (the literal loop's three
// launches per step become two).  Any other entry point first issues the plain step1 launch (ENG / ENG_NOJOIN), so the order of
// effects on the stream is the order of the calls.
// Window chain (small free-body models): the split API's step1 [+ inverse] launch is the chain's ASSEMBLE launch — position and velocity
// stages, controller, [mj_inverse,] and mj_step2's acceleration + constraint rows, handed over in the env's slice; qpos (normalised) and
// qvel (after the controller's velocity override) are stored as mj_step1 leaves them, nothing is integrated — and mjh_step2 is the
// window kernel alone, instead of a second pass over everything the rows need (they only ever exist in LDS).  What the reference's
// loop does between the two (MjHWInterface::read: getters; write: mjh_set_cmd, consumed by the NEXT mj_step1) keeps the hand-over; any
// other call that touches the engine in between drops it, and mjh_step2 runs the whole chain from the state it finds.
static bool split_handover(const mjh_engine* e) {
  static const bool on = !(getenv("MJH_SPLIT_HANDOVER") && atoi(getenv("MJH_SPLIT_HANDOVER")) == 0);
  return on && e->M.window && e->S.wbuf != nullptr;
}
static int flush_step1(mjh_engine* e) {
  e->step1_pending = false;
  if (split_handover(e)) { const int rc = launch_lpt(e, PH_STEP1 | PH_STEP2, XF_FORCE, true, 1); e->handover = rc == MJH_OK; return rc; }
  return launch_lpt(e, PH_STEP1, XF_FORCE, true);
}
extern "C" int mjh_step1(mjh_engine* e) {
  ENG_NOJOIN(e); e->step1_done = true;
  if (e->pd_on) { int rc = join_cohorts(e); if (!rc) rc = launch_pd(e, e->stream, 0, e->nenv); if (rc) return rc; }
  static const bool lazy = !(getenv("MJH_LAZY_STEP1") && atoi(getenv("MJH_LAZY_STEP1")) == 0);
  if (!lazy) return launch_lpt(e, PH_STEP1, XF_FORCE, true);
  e->step1_pending = true;
  return MJH_OK;
}
extern "C" int mjh_inverse(mjh_engine* e) {
  if (e && e->step1_pending) {      // step1 + inverse in one launch
    if (hipSetDevice(e->device) != hipSuccess) { mjh_set_error("hipSetDevice failed"); return MJH_ERR_NO_DEVICE; }
    e->step1_pending = false;
    if (split_handover(e)) { const int rc = launch_lpt(e, PH_STEP1 | PH_INV | PH_STEP2, XF_FORCE, true, 1); e->handover = rc == MJH_OK; return rc; }
    return launch_lpt(e, PH_STEP1 | PH_INV, XF_FORCE, true);
  }
  ENG_NOJOIN(e); return launch_lpt(e, PH_INV, XF_FORCE, false);
}
extern "C" int mjh_step2(mjh_engine* e) {
  ENG_NOJOIN(e);
  if (!e->step1_done) { mjh_set_error("mjh_step2 called before mjh_step1"); return MJH_ERR_STATE; }
  e->step1_done = false;
  e->order_age++;
  if (mjh_ho_) return launch_lpt(e, PH_STEP2, XF_FORCE, false, 2);
  return launch_lpt(e, PH_STEP2, XF_FORCE, false);
}
extern "C" int mjh_forward(mjh_engine* e) {
  ENG(e);
  // mj_forward runs mjcb_control like mj_step1 does, so the in-engine PD law is evaluated in front of it as well
  if (e->pd_on) { int rc = launch_pd(e, e->stream, 0, e->nenv); if (rc) return rc; }
  return launch_lpt(e, PH_STEP1 | PH_NOINT, XF_FORCE, true);
}
extern "C" int mjh_step(mjh_engine* e, int nsteps, int with_inverse) {
  ENG_NOJOIN(e);
  if (e->lpt && !e->d_order && e->nenv >= 1024) {
    int rc = join_cohorts(e);
    if (!rc) rc = dev_alloc(e, &e->d_order, (size_t)e->nenv);
    if (rc) return rc;
  }
  const int G = e->ncohort > 1 && e->nenv >= 64 * e->ncohort ? e->ncohort : 1;
  int rc = G > 1 ? fork_cohorts(e) : join_cohorts(e);
  if (rc) return rc;
  const int ph = PH_STEP1 | PH_STEP2 | (with_inverse ? PH_INV : 0);
  StateGuard guard(&e->S);
  if (e->lpt && e->d_order) e->S.env_order = e->d_order;
  if (e->pd_on && e->pd_target) { e->S.pd_target = e->pd_target; e->S.pd_kp = e->pd_kp; e->S.pd_kd = e->pd_kd; }   // the fused step evaluates the PD law itself
  // In-kernel step loop (step_kernel.h): articulated models in the LDS-resident layout run up to steps_per_launch steps per launch and
  // cohort, the env's state staying in LDS in between (commands are consumed by the first step of the call, the in-engine PD law runs
  // every step); free-body models (contact-patch / window chain) and the many-body chain hand over between launches: one step each
  const int chunk_max = (!e->M.big && !e->M.diagM) ? std::max(1, e->steps_per_launch) : 1;
  for (int s = 0, k = 1; s < nsteps && !rc; s += k, e->order_age += k, e->order_G = G) {
    k = std::min(nsteps - s, chunk_max);
    for (int g = 0; g < G && !rc; g++) {
      const int g0 = (int)((long long)e->nenv * g / G), g1 = (int)((long long)e->nenv * (g + 1) / G);
      hipStream_t st = G > 1 ? e->cstream[g] : e->stream;
      // dispatch the envs with the most solver work first (shorter kernel tail).  An env's cost drifts slowly, so the sort is
      // renewed every MJH_ORDER_EVERY-th step only: its 10 us sit in front of every step launch of the cohort's stream, which
      // is 9 % of a step of the small configs (C3, C5: 0.10 ms kernels)
      const int order_every = order_every_of(e);
      if (e->S.env_order && (e->order_G != G || e->order_age / order_every != (e->order_age - e->last_chunk) / order_every || e->order_age == 0)) {
        const bool dsel = e->M.big && e->split3 && e->M.dense && e->d_dense;
        const unsigned ep = e->dense_epoch[g];
        hipLaunchKernelGGL(mjh_order_kernel, dim3(1), dim3(1024), 0, st, (const int*)e->S.stats, e->d_order, g0, g1 - g0, dsel ? e->d_dense + 4 * g + (ep & 3) : (e->d_wn ? e->d_wn + g : (int*)nullptr), e->d_wn && !dsel ? -1 : e->M.dense_min_iter);
        if (dsel) {
          HIPCHK(hipEventRecord(e->ev_dense[g][ep & 3], st));
          // (this wait bounds how far the host runs ahead of the device for such models: at most two rebuilds of the launch order,
          //  2 x MJH_ORDER_EVERY steps per cohort — INTEGRATION.md §7a)
          if (ep >= 2) { HIPCHK(hipEventSynchronize(e->ev_dense[g][(ep - 2) & 3])); e->dense_now[g] = *(volatile int*)(e->h_dense + 4 * g + ((ep - 2) & 3)) != 0; }
          e->dense_epoch[g] = ep + 1;
        }
      }
      hipEvent_t ta = nullptr, tb = nullptr;
      if (e->timing && (e->timing_count++ % e->timing_stride) == 0) {
        if (e->tev_used == e->tev.size()) { hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b)); e->tev.push_back({a, b}); }
        ta = e->tev[e->tev_used].first; tb = e->tev[e->tev_used].second; e->tev_used++;
        HIPCHK(hipEventRecord(ta, st));
      }
      // the chain of launches of this cohort-step.  dn: dense row-space solver for this cohort's step (the word mjh_order_kernel left two
      // rebuilds of the launch order ago; the assemble launch is TOLD the decision, nothing on the device reads the word); nl: windows of the
      // window kernel's LDS tier (window models: window_tier — an unsynchronised hint read HERE, once per cohort-step)
      const bool chain_big = e->M.big && e->split3;
      const bool chain_win = !chain_big && e->M.window && e->S.wbuf;
      const bool dn = chain_big && e->M.dense && (!e->h_dense || e->dense_now[g]);
      e->cur_cohort = g; const int nl = chain_win ? window_tier(e) : -1; e->cur_cohort = -1;
      auto issue_chain = [&]() -> int {
        int rc = MJH_OK;
        if (e->M.big && e->split3) {
          // many-body layout: assemble -> solve (3 KB of LDS per env instead of ~70 KB: many more resident envs during the
          // sweeps, which are > 90 % of such a step) -> integrate
          // dense row-space solver for this cohort's step?  (the word mjh_order_kernel left two rebuilds of the launch order ago; the
          // assemble launch is TOLD the decision, nothing on the device reads the word)
          rc = launch_on(e, st, g0, g1 - g0, 1, ph | PH_PRE, dn ? XF_DENSE : 0);
          if (!rc && dn) {
            // dense row-space solver (dense_pgs.h): AR = J M^-1 J^T on the matrix cores, then column sweeps, for every env of the
            // launch whose row count fits; the block solver below skips those envs (meta[7])
            HIPCHK(mjh_launch_dense(st, g1 - g0, e->dense_lds, e->dense_solve_lds, e->dC, e->S, g0));
          }
          if (!rc) {
            const size_t lds = (2 * (size_t)(((e->M.nv + 3) / 4) * 4) + 2 * (size_t)std::max(e->M.maxblk, 1) + 4 + 8 + 8 + 8 * (size_t)((e->M.nv + 2) / 3)) * sizeof(float);   // ... + the quad sweep's padded copies (four floats per 3 dofs, twice)   // 2 dof vectors + visiting order + group starts + per-wave partial sums
            const bool xs = e->M.noslip_iterations > 0;      // (the convex narrow phase is not part of the solve launch)
            // wide groups (up to 16 independent blocks) can be shared by several waves per environment (MJH_SOLVE_WAVES = 2, 4: chunk
            // c4 of a group goes to wave c4 % n, a workgroup barrier ends the group).  Measured on C2 (4096 envs, 215 contacts per
            // env): 288 k / 282 k / 259 k env-steps/s with 1 / 2 / 4 waves — the sweep streams 0.4 MB of block operands per env and
            // sweep from L2 / HBM and is bound by that stream, not by the length of the dependent chain; so the default stays 1
            static const int nwave_env = getenv("MJH_SOLVE_WAVES") ? std::max(1, std::min(4, atoi(getenv("MJH_SOLVE_WAVES")))) : 1;
            const dim3 thr(e->M.group_max == 16 && e->M.rowW <= 16 ? 64 * nwave_env : 64);
            if (e->M.diagM) { if (xs) hipLaunchKernelGGL((mjh_solve_kernel<true, true>), dim3(g1 - g0), thr, lds, st, e->dC, e->S, g0);
                              else hipLaunchKernelGGL((mjh_solve_kernel<true, false>), dim3(g1 - g0), thr, lds, st, e->dC, e->S, g0); }
            else { if (xs) hipLaunchKernelGGL((mjh_solve_kernel<false, true>), dim3(g1 - g0), thr, lds, st, e->dC, e->S, g0);
                   else hipLaunchKernelGGL((mjh_solve_kernel<false, false>), dim3(g1 - g0), thr, lds, st, e->dC, e->S, g0); }
            HIPCHK(hipGetLastError());
            rc = launch_on(e, st, g0, g1 - g0, 1, PH_STEP2 | PH_POST, 0);
          }
        } else { e->cur_cohort = g; rc = launch_on(e, st, g0, g1 - g0, k, ph, 0, 0, nl); e->cur_cohort = -1; }
        return rc;
      };
      // One hipGraphLaunch per cohort-step instead of 2 .. 5 kernel launches: the chain is captured once per (cohort, variant) and replayed
      // while nothing its kernels take by value changes (the device-state descriptor, the env range, the phase / tier / dense choice)
      // (not on the NULL stream — it cannot be captured —: small engines without cohort streams on the caller's default stream keep the plain launches)
      if (((chain_big && g_chain_graph >= 1) || (chain_win && g_chain_graph >= 2)) && k == 1 && st != nullptr) {
        const int key = (dn ? 1 : 0) | (with_inverse ? 2 : 0) | ((nl > 0 ? 1 : 0) << 2);
        mjh_engine::ChainGraph& cg = e->cgraph[g][key & 7];
        if (!cg.nocap && (!cg.exec || cg.g0 != g0 || cg.n != g1 - g0 || cg.key != (key | (nl << 8)) || std::memcmp(cg.S, &e->S, sizeof(DState)) != 0)) {
          if (cg.exec) {
            // the retired exec owns the kernel arguments of its launches and mjh_step is asynchronous: its last hipGraphLaunch may still be
            // running on this stream — wait for it before the exec goes (re-capture is rare: a descriptor / range / variant change)
            HIPCHK(hipStreamSynchronize(st));
            (void)hipGraphExecDestroy(cg.exec); cg.exec = nullptr;
          }
          hipGraph_t gr = nullptr;
          if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            rc = issue_chain();
            const hipError_t ce = hipStreamEndCapture(st, &gr);        // (always ended, also when a launch inside failed)
            if (!rc && ce != hipSuccess) { mjh_set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(ce)); rc = MJH_ERR_NO_DEVICE; }
            if (!rc && hipGraphInstantiate(&cg.exec, gr, nullptr, nullptr, 0) != hipSuccess) { cg.exec = nullptr; cg.nocap = true; (void)hipGetLastError(); }   // (latched: no capture attempt per step from here on)
            if (gr) (void)hipGraphDestroy(gr);
            if (!rc && cg.exec) { std::memcpy(cg.S, &e->S, sizeof(DState)); cg.g0 = g0; cg.n = g1 - g0; cg.key = key | (nl << 8); }
          } else (void)hipGetLastError();                                // (a stream that cannot be captured: plain launches below)
        }
        if (!rc) { if (cg.exec) { HIPCHK(hipGraphLaunch(cg.exec, st)); e->last_launches = 1; } else { rc = issue_chain(); e->last_launches = chain_big ? (dn ? 5 : 3) : (chain_win ? 2 : 1); } }
      } else { rc = issue_chain(); e->last_launches = chain_big ? (dn ? 5 : 3) : (chain_win ? 2 : 1); }
      if (ta && !rc) HIPCHK(hipEventRecord(tb, st));
    }
    e->last_chunk = k;
  }
  if (e->lpt && e->d_order && nsteps > 0 && !rc) e->order_valid = true;   // the per-cohort sorts tile a full permutation
  return rc;
}
// Per-launch timing of the step kernels with HIP events on the streams they are launched on (bench.py's roofline leg).
// on = 1: every step launch; on = N > 1: every N-th one (a sample: the event pairs themselves cost several microseconds of
// stream time per launch, which shows in launch-bound configs)
extern "C" int mjh_set_launch_timing(mjh_engine* e, int on) { ENG(e); e->timing = on != 0; e->timing_stride = on > 1 ? on : 1; e->timing_count = 0; e->tev_used = 0; return MJH_OK; }
extern "C" int mjh_get_launch_timing(mjh_engine* e, double* mean_ms, int* count) {
  ENG(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  double acc = 0;
  for (size_t i = 0; i < e->tev_used; i++) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e->tev[i].first, e->tev[i].second)); acc += ms; }
  if (mean_ms) *mean_ms = e->tev_used ? acc / (double)e->tev_used : 0.0;
  if (count) *count = (int)e->tev_used;
  e->tev_used = 0;
  return MJH_OK;
}
// m->opt.timestep is mutable in the reference: simulate() doubles / halves it when the simulation lags the wall clock
// (mj_main.cpp:150-163).  Takes effect with the next launch.
extern "C" int mjh_set_timestep(mjh_engine* e, double dt) {
  ENG(e);
  if (!(dt > 0)) { mjh_set_error("mjh_set_timestep: dt must be positive"); return MJH_ERR_ARG; }
  e->M.timestep = (float)dt; e->M.timestep_d = dt;
  HIPCHK(hipMemcpyAsync((char*)e->dC + offsetof(DConst, M) + offsetof(DModel, timestep), &e->M.timestep, sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync((char*)e->dC + offsetof(DConst, M) + offsetof(DModel, timestep_d), &e->M.timestep_d, sizeof(double), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return MJH_OK;
}
extern "C" double mjh_get_timestep(const mjh_engine* e) { return e ? e->M.timestep_d : 0.0; }
static int reset_dense_choice(mjh_engine* e);
extern "C" int mjh_set_cohorts(mjh_engine* e, int n) { ENG(e); e->last_launches = 0; int rc = set_cohorts(e, n); return rc ? rc : reset_dense_choice(e); }
extern "C" int mjh_get_cohorts(const mjh_engine* e) { return e ? e->ncohort : 0; }
extern "C" int mjh_launches_per_step(const mjh_engine* e) {
  if (!e) return 0;
  // what the last mjh_step actually queued per cohort-step (1: one captured graph; else the kernel launches of the chain it issued — 3 / 5
  // for the many-body layout's block / dense chain, 2 for the window chain); before the first step: what the next one intends
  if (e->last_launches > 0) return e->last_launches;
  const bool big = e->M.big && e->split3, win = !big && e->M.window && e->S.wbuf;
  const int chain = big ? (e->M.dense ? 5 : 3) : (win ? 2 : 1);
  return ((big && g_chain_graph >= 1) || (win && g_chain_graph >= 2)) ? 1 : chain;
}
extern "C" int mjh_set_steps_per_launch(mjh_engine* e, int n) { ENG(e); e->steps_per_launch = std::max(1, std::min(n, 64)); return MJH_OK; }
extern "C" int mjh_get_steps_per_launch(const mjh_engine* e) { return !e ? 0 : ((!e->M.big && !e->M.diagM) ? e->steps_per_launch : 1); }
extern "C" int mjh_synchronize(mjh_engine* e) { ENG(e); KEEP_HANDOVER(e); HIPCHK(hipStreamSynchronize(e->stream)); return MJH_OK; }

// ---- host <-> device marshalling helpers (double on the host side, padded fp32 rows on the device)
// rows of `width` floats at `stride` floats apart; only the `width` floats are touched (rows may be columns of a wider
// per-env record, see the state layout in mjh_create)
// The stream a transfer of the envs [env0, env0 + n) has to be ordered on: while the cohorts run on their own streams, a range inside
// ONE cohort uses that cohort's stream and leaves the others running; anything else joins them into the caller's stream first.
// rows [env0, env0 + n) of up to three per-env arrays <-> one packed block [n x wa | n x wb | n x wc] in host-mapped memory
__global__ void mjh_gather_rows_kernel(const float* __restrict__ a, int sa, int wa, const float* __restrict__ b, int sb, int wb,
                                       const float* __restrict__ c, int sc, int wc, int env0, int n, float* __restrict__ out) {
  const int na = n * wa, nb = n * wb, nc = n * wc;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < na + nb + nc; i += gridDim.x * blockDim.x) {
    float v;
    if (i < na) v = a[(size_t)(env0 + i / wa) * sa + i % wa];
    else if (i < na + nb) { const int j = i - na; v = b[(size_t)(env0 + j / wb) * sb + j % wb]; }
    else { const int j = i - na - nb; v = c[(size_t)(env0 + j / wc) * sc + j % wc]; }
    out[i] = v;
  }
}
__global__ void mjh_scatter_rows_kernel(const float* __restrict__ in, float* __restrict__ a, int sa, int wa, float* __restrict__ b, int sb, int wb, int env0, int n) {
  const int na = n * wa, nb = n * wb;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += gridDim.x * blockDim.x) {
    if (i < na) a[(size_t)(env0 + i / wa) * sa + i % wa] = in[i];
    else { const int j = i - na; b[(size_t)(env0 + j / wb) * sb + j % wb] = in[i]; }
  }
}
static int ensure_io(mjh_engine* e) {
  if (e->h_io) return MJH_OK;
  HIPCHK(hipHostMalloc((void**)&e->h_io, (size_t)(MJH_IO_GET_FLOATS + MJH_IO_SLOTS * MJH_IO_PUT_FLOATS) * sizeof(float), hipHostMallocMapped));
  HIPCHK(hipHostGetDevicePointer((void**)&e->d_io, e->h_io, 0));
  for (int k = 0; k < MJH_IO_SLOTS; k++) HIPCHK(hipEventCreateWithFlags(&e->io_ev[k], hipEventDisableTiming));
  return MJH_OK;
}
static int range_stream(mjh_engine* e, int env0, int n, hipStream_t* st) {
  *st = e->stream;
  if (!e->forked) return MJH_OK;
  const int G = e->ncohort;
  if (n > 0 && G > 1 && e->nenv >= 64 * G)
    for (int g = 0; g < G; g++) {
      const int g0 = (int)((long long)e->nenv * g / G), g1 = (int)((long long)e->nenv * (g + 1) / G);
      if (env0 >= g0 && env0 + n <= g1) { *st = e->cstream[g]; return MJH_OK; }
    }
  return join_cohorts(e);
}
static int put_rows(mjh_engine* e, float* dst, int stride, int width, int env0, int n, const double* src, hipStream_t st = nullptr) {
  if (!src || n == 0) return MJH_OK;
  if (!st) st = e->stream;
  std::vector<float> tmp((size_t)n * width);
  for (size_t i = 0; i < tmp.size(); i++) tmp[i] = (float)src[i];
  HIPCHK(hipMemcpy2DAsync(dst + (size_t)env0 * stride, (size_t)stride * sizeof(float), tmp.data(), (size_t)width * sizeof(float),
                          (size_t)width * sizeof(float), (size_t)n, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  return MJH_OK;
}
// a column block of wider rows: only `width` floats of every row are written (the rest belongs to other tables)
static int put_cols(mjh_engine* e, float* dst, int stride, int width, int env0, int n, const double* src) { return put_rows(e, dst, stride, width, env0, n, src); }
static int get_rows(mjh_engine* e, const float* src, int stride, int width, int env0, int n, double* dst) {
  if (!dst || n == 0) return MJH_OK;
  std::vector<float> tmp((size_t)n * width);
  HIPCHK(hipMemcpy2DAsync(tmp.data(), (size_t)width * sizeof(float), src + (size_t)env0 * stride, (size_t)stride * sizeof(float),
                          (size_t)width * sizeof(float), (size_t)n, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (size_t i = 0; i < tmp.size(); i++) dst[i] = (double)tmp[i];
  return MJH_OK;
}

extern "C" int mjh_set_cmd(mjh_engine* e, int env0, int n, const double* ddq, const double* dq) {
  ENG_NOJOIN(e); KEEP_HANDOVER(e); RANGE(e, env0, n);
  hipStream_t st;
  int rc = range_stream(e, env0, n, &st);
  if (rc) return rc;
  static const bool io_fast = !(getenv("MJH_IO_STAGING") && atoi(getenv("MJH_IO_STAGING")) == 0);
  const int wa = ddq ? e->M.nv : 0, wb = dq ? e->M.nv : 0;
  if (io_fast && n > 0 && wa + wb > 0 && (size_t)n * (wa + wb) <= (size_t)MJH_IO_PUT_FLOATS) {
    // MjHWInterface::write of a few envs: the command goes into the next slot of the host-mapped ring and one small kernel on the
    // range's stream moves it into the per-env records; the call returns without waiting for the device (the slot's event fences
    // its reuse, MJH_IO_SLOTS calls later)
    rc = ensure_io(e);
    if (rc) return rc;
    const unsigned slot = e->io_next++ % MJH_IO_SLOTS;
    if (e->io_used[slot]) HIPCHK(hipEventSynchronize(e->io_ev[slot]));
    float* h = e->h_io + MJH_IO_GET_FLOATS + (size_t)slot * MJH_IO_PUT_FLOATS;
    const size_t na = (size_t)n * wa, nb = (size_t)n * wb;
    for (size_t i = 0; i < na; i++) h[i] = (float)ddq[i];
    for (size_t i = 0; i < nb; i++) h[na + i] = (float)dq[i];
    const int total = (int)(na + nb);
    hipLaunchKernelGGL(mjh_scatter_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)(e->d_io + (h - e->h_io)),
                       wa ? e->S.ddq : e->S.dq, e->M.nvp, wa ? wa : wb, wa ? e->S.dq : nullptr, e->M.nvp, wa ? wb : 0, env0, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->io_ev[slot], st)); e->io_used[slot] = true;
    return MJH_OK;
  }
  rc = put_rows(e, e->S.ddq, e->M.nvp, e->M.nv, env0, n, ddq, st);
  if (rc) return rc;
  return put_rows(e, e->S.dq, e->M.nvp, e->M.nv, env0, n, dq, st);
}

// ---- in-engine PD effort controller.  The reference runs ros_control effort controllers on the host between read() and
// write() (mj_main.cpp:86-106; gains e.g. PID p 200 d 50, model/ontology/box/box.yaml:5-13) for its ONE environment; with
// thousands of environments that hand-off is a PCIe round trip per step, so the same law can run on the device: in front of
// every step, ddq[d] = kp (target[d] - q[d]) - kd qvel[d] on every hinge / slide dof, consumed by the controller stage of
// step1 exactly like a command written through mjh_set_cmd.
__global__ void mjh_pd_kernel(const DConst* __restrict__ C, const DState S, const float* __restrict__ target, float kp, float kd, int env0, int n) {
  const DModel& M = C->M;
  const int nv = M.nv;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)n * nv; i += (size_t)gridDim.x * blockDim.x) {
    const int env = env0 + (int)(i / nv), d = (int)(i % nv);
    const int j = M.I[M.o_dof_jntid + d], jt = M.I[M.o_jnt_type + j];
    if (jt != MJH_JNT_HINGE && jt != MJH_JNT_SLIDE) continue;
    const float q = S.qpos[(size_t)env * M.nqp + M.I[M.o_jnt_qposadr + j]], v = S.qvel[(size_t)env * M.nvp + d];
    S.ddq[(size_t)env * M.nvp + d] = kp * (target[(size_t)env * nv + d] - q) - kd * v;
  }
}
static int launch_pd(mjh_engine* e, hipStream_t st, int env0, int n) {
  if (!e->pd_on || !e->pd_target || n <= 0) return MJH_OK;
  const size_t total = (size_t)n * e->M.nv;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(mjh_pd_kernel, dim3(blocks), dim3(256), 0, st, e->dC, e->S, (const float*)e->pd_target, e->pd_kp, e->pd_kd, env0, n);
  HIPCHK(hipGetLastError());
  return MJH_OK;
}
extern "C" int mjh_set_pd_controller(mjh_engine* e, double kp, double kd) {
  ENG(e);
  if (kp < 0 || kd < 0) { mjh_set_error("mjh_set_pd_controller: gains must be non-negative"); return MJH_ERR_ARG; }
  if ((kp > 0 || kd > 0) && !e->pd_target) {    // targets start at qpos0 of every dof's joint
    int rc = dev_alloc(e, &e->pd_target, (size_t)e->nenv * std::max(e->M.nv, 1));
    if (rc) return rc;
    const mjh_model* m = e->model;
    std::vector<float> t((size_t)e->nenv * std::max(m->nv, 1), 0.0f);
    for (int en = 0; en < e->nenv; en++) for (int d = 0; d < m->nv; d++) {
      const int j = m->dof_jntid[d];
      if (m->jnt_type[j] == MJH_JNT_HINGE || m->jnt_type[j] == MJH_JNT_SLIDE) t[(size_t)en * m->nv + d] = (float)m->qpos0[m->jnt_qposadr[j]];
    }
    HIPCHK(hipMemcpyAsync(e->pd_target, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  e->pd_kp = (float)kp; e->pd_kd = (float)kd; e->pd_on = kp > 0 || kd > 0;
  return MJH_OK;
}
extern "C" int mjh_set_pd_target(mjh_engine* e, int env0, int n, const double* target) {
  ENG(e); RANGE(e, env0, n);
  if (!target) return MJH_ERR_ARG;
  if (!e->pd_target) { mjh_set_error("mjh_set_pd_target: enable the controller first (mjh_set_pd_controller)"); return MJH_ERR_STATE; }
  return put_rows(e, e->pd_target, e->M.nv, e->M.nv, env0, n, target);
}

static int patch_int_table(mjh_engine* e, int off, const int* vals, int n) {
  for (int i = 0; i < n; i++) e->hI[off + i] = vals[i];
  HIPCHK(hipMemcpyAsync(e->dI + off, e->hI.data() + off, n * sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return MJH_OK;
}
extern "C" int mjh_set_controlled_dofs(mjh_engine* e, const int* mask) {
  ENG(e); if (!mask) return MJH_ERR_ARG;
  return patch_int_table(e, e->o_controlled, mask, e->M.nv);
}
extern "C" int mjh_set_odom_dofs(mjh_engine* e, const int lin[3], const int ang[3], const int angq[3]) {
  ENG(e);
  int t[10];
  for (int k = 0; k < 3; k++) { t[k] = lin ? lin[k] : -1; t[3+k] = ang ? ang[k] : -1; t[6+k] = angq ? angq[k] : -1; }
  t[9] = 0; for (int k = 0; k < 6; k++) if (t[k] >= 0) t[9] = 1;
  return patch_int_table(e, e->o_odom, t, 10);
}
extern "C" int mjh_set_odom_vel(mjh_engine* e, int env0, int n, const double* twist) {
  ENG(e); RANGE(e, env0, n);
  return put_rows(e, e->S.odom_vel, 6, 6, env0, n, twist);
}

extern "C" int mjh_get_joint_state(mjh_engine* e, int env0, int n, double* qpos, double* qvel, double* qfrc_inverse) {
  ENG_NOJOIN(e); KEEP_HANDOVER(e); RANGE(e, env0, n);
  hipStream_t st;
  int rc = range_stream(e, env0, n, &st);
  if (rc || n == 0) return rc;
  static const bool io_fast = !(getenv("MJH_IO_STAGING") && atoi(getenv("MJH_IO_STAGING")) == 0);
  {
    const int wa = qpos ? e->M.nq : 0, wb = qvel ? e->M.nv : 0, wc = qfrc_inverse ? e->M.nv : 0;
    if (io_fast && wa + wb + wc > 0 && (size_t)n * (wa + wb + wc) <= (size_t)MJH_IO_GET_FLOATS) {
      // MjHWInterface::read of a few envs (once per step of the reference's loop): one small kernel packs the rows into host-mapped
      // memory, ONE synchronisation of the range's stream (33 -> 12 us on an idle engine against three pageable 2D copies)
      rc = ensure_io(e);
      if (rc) return rc;
      const int total = n * (wa + wb + wc);
      hipLaunchKernelGGL(mjh_gather_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)e->S.qpos, e->M.nqp, wa,
                         (const float*)e->S.qvel, e->M.nvp, wb, (const float*)e->S.qfrc_inverse, e->M.nvp, wc, env0, n, e->d_io);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(st));
      const volatile float* h = e->h_io;
      const size_t na = (size_t)n * wa, nb = (size_t)n * wb, nc = (size_t)n * wc;
      for (size_t i = 0; i < na; i++) qpos[i] = (double)h[i];
      for (size_t i = 0; i < nb; i++) qvel[i] = (double)h[na + i];
      for (size_t i = 0; i < nc; i++) qfrc_inverse[i] = (double)h[na + nb + i];
      return MJH_OK;
    }
  }
  // three strided copies, ONE synchronisation
  const float* src[3] = {e->S.qpos, e->S.qvel, e->S.qfrc_inverse};
  const int stride[3] = {e->M.nqp, e->M.nvp, e->M.nvp}, width[3] = {e->M.nq, e->M.nv, e->M.nv};
  double* dst[3] = {qpos, qvel, qfrc_inverse};
  std::vector<float> tmp[3];
  for (int k = 0; k < 3; k++) {
    if (!dst[k]) continue;
    tmp[k].resize((size_t)n * width[k]);
    HIPCHK(hipMemcpy2DAsync(tmp[k].data(), (size_t)width[k] * sizeof(float), src[k] + (size_t)env0 * stride[k], (size_t)stride[k] * sizeof(float),
                            (size_t)width[k] * sizeof(float), (size_t)n, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  for (int k = 0; k < 3; k++) if (dst[k]) for (size_t i = 0; i < tmp[k].size(); i++) dst[k][i] = (double)tmp[k][i];
  return MJH_OK;
}

// position-stage recompute for a range of envs with export pointers set
static int fk_export(mjh_engine* e, int env0, int n, double* a, int wa, double* b, int wb, bool geoms) {
  const size_t fa = (size_t)n * wa, fb = (size_t)n * wb;
  int rc = ensure_scratch(e, fa + fb);
  if (rc) return rc;
  {
    StateGuard guard(&e->S);
    if (geoms) { e->S.x_gpos = e->scratch; e->S.x_gmat = e->scratch + fa; } else { e->S.x_xpos = e->scratch; e->S.x_xquat = e->scratch + fa; }
    rc = launch(e, env0, n, 1, PH_FKONLY, geoms ? XF_GEOM : XF_BODY);
  }
  if (rc) return rc;
  if (!rc) rc = get_rows(e, e->scratch, wa, wa, 0, n, a);
  if (!rc) rc = get_rows(e, e->scratch + fa, wb, wb, 0, n, b);
  return rc;
}
extern "C" int mjh_get_body_state(mjh_engine* e, int env0, int n, double* xpos, double* xquat) {
  ENG(e); RANGE(e, env0, n);
  return fk_export(e, env0, n, xpos, 3 * e->M.nbody, xquat, 4 * e->M.nbody, false);
}
extern "C" int mjh_get_geom_state(mjh_engine* e, int env0, int n, double* gpos, double* gmat) {
  ENG(e); RANGE(e, env0, n);
  return fk_export(e, env0, n, gpos, 3 * e->M.ngeom, gmat, 9 * e->M.ngeom, true);
}

extern "C" int mjh_get_state(mjh_engine* e, int env0, int n, double* time, double* qpos, double* qvel, double* ws) {
  ENG(e); KEEP_HANDOVER(e); RANGE(e, env0, n);
  int rc = MJH_OK;
  if (time && n) { HIPCHK(hipMemcpyAsync(time, e->S.time + env0, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipStreamSynchronize(e->stream)); }
  if (!rc) rc = get_rows(e, e->S.qpos, e->M.nqp, e->M.nq, env0, n, qpos);
  if (!rc) rc = get_rows(e, e->S.qvel, e->M.nvp, e->M.nv, env0, n, qvel);
  if (!rc) rc = get_rows(e, e->S.qacc_ws, e->M.nvp, e->M.nv, env0, n, ws);
  return rc;
}
extern "C" int mjh_set_state(mjh_engine* e, int env0, int n, const double* time, const double* qpos, const double* qvel, const double* ws) {
  ENG(e); RANGE(e, env0, n);
  int rc = MJH_OK;
  if (time && n) { HIPCHK(hipMemcpyAsync(e->S.time + env0, time, (size_t)n * sizeof(double), hipMemcpyHostToDevice, e->stream)); HIPCHK(hipStreamSynchronize(e->stream)); }
  if (!rc) rc = put_rows(e, e->S.qpos, e->M.nqp, e->M.nq, env0, n, qpos);
  if (!rc) rc = put_rows(e, e->S.qvel, e->M.nvp, e->M.nv, env0, n, qvel);
  if (!rc) rc = put_rows(e, e->S.qacc_ws, e->M.nvp, e->M.nv, env0, n, ws);
  return rc;
}

// add_old_state() (mj_sim.cpp:465-558) for all environments at once: host-side gather / scatter between two engines
extern "C" int mjh_transplant_state(mjh_engine* from, mjh_engine* to, int qpos_mode) {
  if (!from || !to || from == to) { mjh_set_error("mjh_transplant_state: bad engines"); return MJH_ERR_ARG; }
  const mjh_model *ma = from->model, *mb = to->model;
  const int n = std::min(from->nenv, to->nenv);
  static const int QN[4] = {7, 4, 1, 1};
  const size_t nqa = ma->nq, nva = ma->nv, nqb = mb->nq, nvb = mb->nv;
  std::vector<double> ta(n), qa(n * nqa), va(n * nva), wa(n * nva), fa(n * nva), aa(n * nva);
  std::vector<double> tb(n), qb(n * nqb), vb(n * nvb), wb(n * nvb), fb(n * nvb), ab(n * nvb);
  int rc = mjh_get_state(from, 0, n, ta.data(), qa.data(), va.data(), wa.data());
  if (!rc) rc = mjh_get_field(from, "qfrc_applied", 0, n, fa.data());
  if (!rc) rc = mjh_get_field(from, "qacc", 0, n, aa.data());
  if (!rc) rc = mjh_get_state(to, 0, n, tb.data(), qb.data(), vb.data(), wb.data());
  if (!rc) rc = mjh_get_field(to, "qfrc_applied", 0, n, fb.data());
  if (!rc) rc = mjh_get_field(to, "qacc", 0, n, ab.data());
  if (rc) return rc;
  // d_new->xfrc_applied[body] = d->xfrc_applied[body] (mj_sim.cpp:496-500), if the old engine carries any
  std::vector<double> xa, xb;
  const bool has_x = from->S.xfrc_applied != nullptr;
  if (has_x) {
    xa.resize((size_t)n * 6 * ma->nbody); xb.assign((size_t)n * 6 * mb->nbody, 0.0);
    rc = mjh_get_xfrc_applied(from, 0, n, xa.data());
    if (!rc) rc = mjh_get_xfrc_applied(to, 0, n, xb.data());
    if (rc) return rc;
  }
  int matched = 0;
  for (int ba = 1; ba < ma->nbody; ba++) {
    const char* name = ma->body_names ? ma->body_names[ba] : nullptr;
    if (!name || !*name) continue;
    const int bb = mjh_name2id(mb, 0, name);
    if (bb <= 0) continue;
    matched++;
    if (has_x) for (int i = 0; i < n; i++) for (int k = 0; k < 6; k++) xb[((size_t)i * mb->nbody + bb) * 6 + k] = xa[((size_t)i * ma->nbody + ba) * 6 + k];
    const int ja = ma->body_jntnum[ba], jb = mb->body_jntnum[bb];
    if (ja == jb && ja > 0) {
      const int pa = ma->jnt_qposadr[ma->body_jntadr[ba]], pb = mb->jnt_qposadr[mb->body_jntadr[bb]];
      int cnt = ja;                                   // literal: one scalar per joint (mj_sim.cpp:510-513)
      if (qpos_mode) { cnt = 0; for (int j = 0; j < ja; j++) cnt += QN[ma->jnt_type[ma->body_jntadr[ba] + j]]; }
      bool same = true;
      for (int j = 0; j < ja; j++) same &= ma->jnt_type[ma->body_jntadr[ba] + j] == mb->jnt_type[mb->body_jntadr[bb] + j];
      if (same || !qpos_mode)
        for (int i = 0; i < n; i++) for (int k = 0; k < cnt; k++) qb[i * nqb + pb + k] = qa[i * nqa + pa + k];
    }
    const int da = ma->body_dofnum[ba], db = mb->body_dofnum[bb];
    if (da == db && da > 0) {
      const int oa = ma->body_dofadr[ba], ob = mb->body_dofadr[bb];
      for (int i = 0; i < n; i++) for (int k = 0; k < da; k++) {
        vb[i * nvb + ob + k] = va[i * nva + oa + k]; wb[i * nvb + ob + k] = wa[i * nva + oa + k];
        fb[i * nvb + ob + k] = fa[i * nva + oa + k]; ab[i * nvb + ob + k] = aa[i * nva + oa + k];
      }
    }
  }
  for (int i = 0; i < n; i++) tb[i] = ta[i];          // d_new->time = d->time
  rc = put_rows(to, to->S.qacc, to->M.nvp, to->M.nv, 0, n, ab.data());      // (qacc shares its array with qacc_warmstart, which is written next)
  if (!rc) rc = mjh_set_state(to, 0, n, tb.data(), qb.data(), vb.data(), wb.data());
  if (!rc) rc = put_rows(to, to->S.qfrc_applied, to->M.nvp, to->M.nv, 0, n, fb.data());
  if (!rc && has_x) rc = mjh_set_xfrc_applied(to, 0, n, xb.data());
  return rc ? rc : matched;
}

extern "C" int mjh_get_field(mjh_engine* e, const char* name, int env0, int n, double* out) {
  ENG(e); KEEP_HANDOVER(e); RANGE(e, env0, n);
  if (!name) return MJH_ERR_ARG;
  const std::string s(name);
  const int nvp = e->M.nvp, nv = e->M.nv;
  if (s == "qacc") return get_rows(e, e->S.qacc, nvp, nv, env0, n, out);
  if (s == "qfrc_bias") return get_rows(e, e->S.x_bias, nvp, nv, env0, n, out);
  if (s == "qfrc_passive") return get_rows(e, e->S.x_passive, nvp, nv, env0, n, out);
  if (s == "qacc_smooth") return get_rows(e, e->S.x_smooth, nvp, nv, env0, n, out);
  if (s == "qfrc_constraint") return get_rows(e, e->S.x_constraint, nvp, nv, env0, n, out);
  if (s == "qfrc_applied") return get_rows(e, e->S.qfrc_applied, nvp, nv, env0, n, out);
  if (s == "qfrc_inverse") return get_rows(e, e->S.qfrc_inverse, nvp, nv, env0, n, out);
  if (s == "qacc_warmstart") return get_rows(e, e->S.qacc_ws, nvp, nv, env0, n, out);
  if (s == "energy") return get_rows(e, e->S.x_energy, 2, 2, env0, n, out);
  mjh_set_error("mjh_get_field: unknown field " + s);
  return MJH_ERR_ARG;
}

extern "C" int mjh_get_stats(mjh_engine* e, int env0, int n, int* out) {
  ENG(e); KEEP_HANDOVER(e); RANGE(e, env0, n);
  HIPCHK(hipMemcpyAsync(out, e->S.stats + (size_t)env0 * 4, (size_t)n * 4 * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int i = 0; i < n; i++) out[4*i + 3] &= 0xff;   // (bits 8.. : the kernel's cost hint for the launch order, not a flag)
  return MJH_OK;
}

extern "C" int mjh_get_contacts(mjh_engine* e, int env, double* dist, double* pos, double* frame, int* geom) {
  ENG(e); RANGE(e, env, 1);
  const int mc = e->M.maxcon;
  int rc = ensure_scratch(e, (size_t)mc * CON_STRIDE + 4);
  if (rc) return rc;
  {
    // read-only snapshot: position stage up to the collision stage, the records and their count go to the scratch buffer
    // and the launch ends there (XF_NOSTORE): state, statistics, time and warm start of the env are not touched
    StateGuard guard(&e->S);
    e->S.x_contacts = e->scratch;
    rc = launch(e, env, 1, 1, 0, XF_CON | XF_NOSTORE);
  }
  if (rc) return rc;
  std::vector<float> tmp((size_t)mc * CON_STRIDE + 4);
  HIPCHK(hipMemcpyAsync(tmp.data(), e->scratch, tmp.size() * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  int ncon; std::memcpy(&ncon, tmp.data() + (size_t)mc * CON_STRIDE, 4);
  for (int c = 0; c < ncon; c++) {
    const float* r = tmp.data() + (size_t)c * CON_STRIDE;
    if (dist) dist[c] = r[0];
    if (pos) for (int k = 0; k < 3; k++) pos[3*c+k] = r[1+k];
    if (frame) for (int k = 0; k < 9; k++) frame[9*c+k] = r[4+k];
    if (geom) { int g; std::memcpy(&g, r + CON_GEOMS, 4); geom[2*c] = g & 0xfff; geom[2*c+1] = (g >> 12) & 0xfff; }
  }
  return ncon;
}

extern "C" int mjh_mulM(mjh_engine* e, int env0, int n, const double* vec, double* res) {
  ENG(e); RANGE(e, env0, n);
  if (!vec || !res) return MJH_ERR_ARG;
  const int nvp = e->M.nvp;
  int rc = ensure_scratch(e, (size_t)2 * n * nvp);
  if (rc) return rc;
  rc = put_rows(e, e->scratch, nvp, e->M.nv, 0, n, vec);
  if (rc) return rc;
  {
    StateGuard guard(&e->S);
    e->S.x_vec = e->scratch; e->S.x_res = e->scratch + (size_t)n * nvp;
    rc = launch(e, env0, n, 1, PH_MULM, 0);
  }
  if (rc) return rc;
  return get_rows(e, e->scratch + (size_t)n * nvp, nvp, e->M.nv, 0, n, res);
}

extern "C" int mjh_set_env_param(mjh_engine* e, int which, int env0, int n, const double* values) {
  ENG(e); RANGE(e, env0, n);
  if (which < 0 || which >= MJH_EP_COUNT || !values) return MJH_ERR_ARG;
  const mjh_model* m = e->model;
  const int widths[MJH_EP_COUNT] = {3 * m->ngeom, m->ngeom, m->nbody, 3 * m->nbody, 2 * m->nbody, m->nv};
  const double* defaults[MJH_EP_COUNT] = {m->geom_size, m->geom_rbound, m->body_mass, m->body_inertia, m->body_invweight0, m->dof_invweight0};
  const int w = widths[which];
  if (!e->p_tables[0]) {
    // first use: ONE array with a row per env holding all six tables back to back (rows padded to 64 B), every table
    // initialised with the shared model's values; the six device pointers address their column of it
    int off[MJH_EP_COUNT], total = 0;
    for (int k = 0; k < MJH_EP_COUNT; k++) { off[k] = total; total += widths[k]; }
    const int P = ((total + 15) / 16) * 16;
    float* p = nullptr;
    int rc = dev_alloc(e, &p, (size_t)e->nenv * P, false);
    if (rc) return rc;
    std::vector<float> init((size_t)e->nenv * P, 0.0f);
    for (int en = 0; en < e->nenv; en++) for (int k = 0; k < MJH_EP_COUNT; k++) for (int i = 0; i < widths[k]; i++) init[(size_t)en * P + off[k] + i] = (float)defaults[k][i];
    HIPCHK(hipMemcpyAsync(p, init.data(), init.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    const float** slots[MJH_EP_COUNT] = {&e->S.p_geom_size, &e->S.p_geom_rbound, &e->S.p_body_mass, &e->S.p_body_inertia, &e->S.p_body_invweight0, &e->S.p_dof_invweight0};
    for (int k = 0; k < MJH_EP_COUNT; k++) { e->p_tables[k] = p + off[k]; *slots[k] = p + off[k]; }
    e->S.p_stride = P;
  }
  return put_cols(e, e->p_tables[which], e->S.p_stride, w, env0, n, values);
}

// d->xfrc_applied (mj_sim.cpp:499): force + torque per body at its centre of mass, world frame
extern "C" int mjh_set_xfrc_applied(mjh_engine* e, int env0, int n, const double* xfrc) {
  ENG(e); RANGE(e, env0, n);
  if (!xfrc) return MJH_ERR_ARG;
  const int w = 6 * e->M.nbody;
  if (!e->S.xfrc_applied) {
    const int stride = ((w + 3) / 4) * 4;
    int rc = dev_alloc(e, &e->S.xfrc_applied, (size_t)e->nenv * stride);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    e->S.xfrc_stride = stride;
  }
  return put_rows(e, e->S.xfrc_applied, e->S.xfrc_stride, w, env0, n, xfrc);
}
extern "C" int mjh_get_xfrc_applied(mjh_engine* e, int env0, int n, double* xfrc) {
  ENG(e); RANGE(e, env0, n);
  if (!xfrc) return MJH_ERR_ARG;
  const int w = 6 * e->M.nbody;
  if (!e->S.xfrc_applied) { std::memset(xfrc, 0, (size_t)n * w * sizeof(double)); return MJH_OK; }
  return get_rows(e, e->S.xfrc_applied, e->S.xfrc_stride, w, env0, n, xfrc);
}
// d->sensordata as mj_step2's mj_sensorAcc left it (the reference's publisher reads it unlocked at its own rate, mj_ros.cpp:1933-1966)
extern "C" int mjh_get_sensordata(mjh_engine* e, int env0, int n, double* out) {
  ENG(e); RANGE(e, env0, n);
  if (!out) return MJH_ERR_ARG;
  if (e->M.nsensor == 0) return MJH_OK;
  return get_rows(e, e->S.sensordata, 3 * e->M.nsensor, 3 * e->M.nsensor, env0, n, out);
}
// d->mocap_pos / d->mocap_quat of one mocap body for a range of environments
extern "C" int mjh_set_mocap_pose(mjh_engine* e, int env0, int n, int mocapid, const double* pos, const double* quat) {
  ENG(e); RANGE(e, env0, n);
  if (mocapid < 0 || mocapid >= e->M.nmocap) { mjh_set_error("mjh_set_mocap_pose: no such mocap body"); return MJH_ERR_ARG; }
  const int stride = 7 * e->M.nmocap;
  if (pos) { int rc = put_rows(e, e->S.mocap + 7 * mocapid, stride, 3, env0, n, pos); if (rc) return rc; }
  if (quat) {
    std::vector<double> q((size_t)n * 4);
    for (int i = 0; i < n; i++) {       // stored normalised (mj_kinematics normalises mocap_quat on use)
      const double* s = quat + 4 * (size_t)i; double nn = std::sqrt(s[0]*s[0] + s[1]*s[1] + s[2]*s[2] + s[3]*s[3]);
      if (!(nn > 1e-12)) { mjh_set_error("mjh_set_mocap_pose: zero quaternion"); return MJH_ERR_ARG; }
      for (int k = 0; k < 4; k++) q[4 * (size_t)i + k] = s[k] / nn;
    }
    return put_rows(e, e->S.mocap + 7 * mocapid + 3, stride, 4, env0, n, q.data());
  }
  return MJH_OK;
}

extern "C" int mjh_set_initial_qpos(mjh_engine* e, int env0, int n, const double* qpos) {
  ENG(e); RANGE(e, env0, n);
  return put_rows(e, e->S.initial_qpos, e->M.nqp, e->M.nq, env0, n, qpos);
}

// The per-cohort dense / block solver choice (the words mjh_order_kernel leaves in host-mapped memory, adopted two sorts later) is part of
// what a run from reset reproduces: a full reset and a new cohort split start it over (dense for everyone, epoch 0, launch order renewed).
// Drains the stream: the order kernels in flight still write the words.
static int reset_dense_choice(mjh_engine* e) {
  if (!e->h_dense) return MJH_OK;
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int g = 0; g < MJH_MAX_COHORTS * 4; g++) e->h_dense[g] = 1;
  for (int g = 0; g < MJH_MAX_COHORTS; g++) { e->dense_now[g] = true; e->dense_epoch[g] = 0; }
  e->order_age = 0; e->order_G = -1;
  return MJH_OK;
}
extern "C" int mjh_reset(mjh_engine* e, const int* env_ids, int n) {
  ENG(e);
  if (!env_ids) {
    if (e->h_wn) for (int g = 0; g < MJH_MAX_COHORTS; g++) e->h_wn[g] = 1 << 20;     // (row counts of the old state: keep the window kernel's LDS tier until the next look)
    int rc = reset_dense_choice(e); return rc ? rc : launch(e, 0, e->nenv, 1, PH_RESET, 0);
  }
  for (int i = 0; i < n; i++) {
    if (env_ids[i] < 0 || env_ids[i] >= e->nenv) { mjh_set_error("mjh_reset: env id out of range"); return MJH_ERR_ARG; }
    int rc = launch(e, env_ids[i], 1, 1, PH_RESET, 0);
    if (rc) return rc;
  }
  return MJH_OK;
}

// spawn/destroy as batched slot re-layout (reference: spawn_objects / destroy_objects,
// mj_ros.cpp:906-1507 -> full XML round trip + recompile): a pre-allocated free body is toggled per env.
extern "C" int mjh_set_slot_active(mjh_engine* e, int env0, int n, int body, int active) {
  ENG(e); RANGE(e, env0, n);
  const int sbase = e->M.nbody > 32 ? e->M.nbody - 32 : 0;     // the last 32 bodies of a big model are the toggleable slots
  if (body <= 0 || body >= e->M.nbody || body < sbase) {
    mjh_set_error("mjh_set_slot_active: body must be one of the last 32 bodies, [" + std::to_string(std::max(1, sbase)) + ", " + std::to_string(e->M.nbody) + ")");
    return MJH_ERR_ARG;
  }
  const int bit = body - sbase;
  if (!e->S.slot_mask) { int rc = dev_alloc(e, &e->S.slot_mask, (size_t)e->nenv); if (rc) return rc; }
  std::vector<unsigned> h(n);
  HIPCHK(hipMemcpyAsync(h.data(), e->S.slot_mask + env0, n * sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int i = 0; i < n; i++) h[i] = active ? (h[i] & ~(1u << bit)) : (h[i] | (1u << bit));
  HIPCHK(hipMemcpyAsync(e->S.slot_mask + env0, h.data(), n * sizeof(unsigned), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return MJH_OK;
}
// Batched spawn / destroy: the reference's services take a LIST of objects per call (spawn_objects / destroy_objects,
// mj_ros.cpp:859-904,1430-1507); with many environments one call carries (env, body) pairs and the whole list goes to the device
// in one upload + one small kernel instead of three copies and syncs per object.
__global__ void mjh_slot_kernel(const DConst* __restrict__ C, const DState S, const int* __restrict__ env, const int* __restrict__ body,
                                const float* __restrict__ pose /* [n][13] pos3 quat4 vel6, or null: destroy */, int n, int sbase,
                                const int* __restrict__ qadr, const int* __restrict__ dadr) {
  const DModel& M = C->M;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int e = env[i], b = body[i];
  const unsigned bit = 1u << (b - sbase);
  if (pose) {
    atomicAnd(&S.slot_mask[e], ~bit);
    const float* p = pose + 13 * (size_t)i;
    float* q = S.qpos + (size_t)e * M.nqp + qadr[i]; float* v = S.qvel + (size_t)e * M.nvp + dadr[i];
    for (int k = 0; k < 7; k++) q[k] = p[k];
    for (int k = 0; k < 6; k++) v[k] = p[7 + k];
  } else atomicOr(&S.slot_mask[e], bit);
}
static int slot_batch(mjh_engine* e, int n, const int* env, const int* body, const double* pos, const double* quat, const double* vel, bool spawn) {
  if (n <= 0) return MJH_OK;
  if (!env || !body || (spawn && !pos)) { mjh_set_error("mjh_spawn/destroy_objects: null argument"); return MJH_ERR_ARG; }
  const mjh_model* m = e->model;
  const int sbase = e->M.nbody > 32 ? e->M.nbody - 32 : 0;
  std::vector<int> ib(4 * (size_t)n);     // env | body | qpos address | dof address
  std::vector<float> fp(spawn ? 13 * (size_t)n : 0);
  for (int i = 0; i < n; i++) {
    const int b = body[i];
    if (env[i] < 0 || env[i] >= e->nenv || b <= 0 || b >= m->nbody || b < sbase) { mjh_set_error("mjh_spawn/destroy_objects: env or body out of range (slots are the last 32 bodies)"); return MJH_ERR_ARG; }
    ib[i] = env[i]; ib[n + i] = b;
    if (spawn) {
      if (m->body_jntnum[b] != 1 || m->jnt_type[m->body_jntadr[b]] != MJH_JNT_FREE) { mjh_set_error("mjh_spawn_objects: not a free body"); return MJH_ERR_ARG; }
      ib[2 * (size_t)n + i] = m->jnt_qposadr[m->body_jntadr[b]]; ib[3 * (size_t)n + i] = m->body_dofadr[b];
      float* p = &fp[13 * (size_t)i];
      for (int k = 0; k < 3; k++) p[k] = (float)pos[3 * (size_t)i + k];
      for (int k = 0; k < 4; k++) p[3 + k] = (float)(quat ? quat[4 * (size_t)i + k] : (k == 0));
      for (int k = 0; k < 6; k++) p[7 + k] = (float)(vel ? vel[6 * (size_t)i + k] : 0.0);
    }
  }
  if (!e->S.slot_mask) { int rc = dev_alloc(e, &e->S.slot_mask, (size_t)e->nenv); if (rc) return rc; }
  DevBuf di, df;
  HIPCHK(hipMalloc(&di.p, ib.size() * sizeof(int)));
  HIPCHK(hipMemcpyAsync(di.p, ib.data(), ib.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
  if (spawn) { HIPCHK(hipMalloc(&df.p, fp.size() * sizeof(float))); HIPCHK(hipMemcpyAsync(df.p, fp.data(), fp.size() * sizeof(float), hipMemcpyHostToDevice, e->stream)); }
  const int* d = (const int*)di.p;
  hipLaunchKernelGGL(mjh_slot_kernel, dim3((n + 127) / 128), dim3(128), 0, e->stream, e->dC, e->S, d, d + n, (const float*)df.p, n, sbase, d + 2 * (size_t)n, d + 3 * (size_t)n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));      // (the temporaries are freed on return)
  return MJH_OK;
}
extern "C" int mjh_spawn_objects(mjh_engine* e, int n, const int* env, const int* body, const double* pos, const double* quat, const double* vel) {
  ENG(e); return slot_batch(e, n, env, body, pos, quat, vel, true);
}
extern "C" int mjh_destroy_objects(mjh_engine* e, int n, const int* env, const int* body) {
  ENG(e); return slot_batch(e, n, env, body, nullptr, nullptr, nullptr, false);
}
// pose + twist of one free body of one env (initial state of a spawned object, mj_ros.cpp:1406-1412)
extern "C" int mjh_set_body_pose(mjh_engine* e, int env, int body, const double pos[3], const double quat[4], const double vel[6]) {
  ENG(e); RANGE(e, env, 1);
  const mjh_model* m = e->model;
  if (body <= 0 || body >= m->nbody || m->body_jntnum[body] != 1 || m->jnt_type[m->body_jntadr[body]] != MJH_JNT_FREE) {
    mjh_set_error("mjh_set_body_pose: not a free body"); return MJH_ERR_ARG; }
  const int qa = m->jnt_qposadr[m->body_jntadr[body]], da = m->body_dofadr[body];
  float q[7], v[6];
  for (int k = 0; k < 3; k++) q[k] = (float)pos[k];
  for (int k = 0; k < 4; k++) q[3+k] = (float)(quat ? quat[k] : (k == 0));
  for (int k = 0; k < 6; k++) v[k] = (float)(vel ? vel[k] : 0.0);
  HIPCHK(hipMemcpyAsync(e->S.qpos + (size_t)env * e->M.nqp + qa, q, sizeof q, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->S.qvel + (size_t)env * e->M.nvp + da, v, sizeof v, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return MJH_OK;
}

// ---- pinned host mirror of a range of environments: what the reference's publisher threads read from
// mjData (mj_ros.cpp:1968-2194: qpos / qvel / qfrc_inverse for joint states, xpos / xquat for tf and object states,
// geom_xpos / geom_xmat for markers), refreshed asynchronously at each topic's rate instead of per step.
struct mjh_mirror {
  mjh_engine* e = nullptr; int env0 = 0, n = 0;
  float* host = nullptr; float* dev = nullptr;          // pinned block / device staging of the FK exports
  size_t off[8] = {0}; int width[8] = {0}; size_t total = 0, fk_floats = 0;
  hipEvent_t ev = nullptr;
};
extern "C" int mjh_mirror_create(mjh_engine* e, int env0, int n, mjh_mirror** out) {
  ENG(e); RANGE(e, env0, n);
  if (!out || n <= 0) { mjh_set_error("mjh_mirror_create: bad argument"); return MJH_ERR_ARG; }
  mjh_mirror* m = new mjh_mirror(); m->e = e; m->env0 = env0; m->n = n;
  // (field 0, time, is one DOUBLE per env = two floats of row width: mjh_mirror_time)
  const int w[8] = {2, e->M.nq, e->M.nv, e->M.nv, 3 * e->M.nbody, 4 * e->M.nbody, 3 * e->M.ngeom, 9 * e->M.ngeom};
  for (int k = 0; k < 8; k++) { m->width[k] = w[k]; m->off[k] = m->total; m->total += (size_t)n * w[k]; }
  m->fk_floats = m->total - m->off[4];
  if (hipHostMalloc((void**)&m->host, m->total * sizeof(float), hipHostMallocDefault) != hipSuccess ||
      hipMalloc((void**)&m->dev, std::max<size_t>(m->fk_floats, 1) * sizeof(float)) != hipSuccess ||
      hipEventCreateWithFlags(&m->ev, hipEventDisableTiming) != hipSuccess) {
    mjh_set_error("mjh_mirror_create: allocation failed");
    if (m->host) (void)hipHostFree(m->host); if (m->dev) (void)hipFree(m->dev); delete m; return MJH_ERR_NO_DEVICE;
  }
  std::memset(m->host, 0, m->total * sizeof(float));
  *out = m;
  return MJH_OK;
}
extern "C" void mjh_mirror_destroy(mjh_mirror* m) {
  if (!m) return;
  (void)hipEventSynchronize(m->ev);
  (void)hipHostFree(m->host); (void)hipFree(m->dev); (void)hipEventDestroy(m->ev);
  delete m;
}
// enqueue a refresh of the selected parts (MJH_MIRROR_* bits) behind everything queued so far; returns at once
extern "C" int mjh_mirror_update(mjh_mirror* m, int what) {
  if (!m) { mjh_set_error("null mirror"); return MJH_ERR_ARG; }
  mjh_engine* e = m->e;
  ENG(e);
  const int n = m->n, env0 = m->env0;
  auto rows = [&](int k, const float* src, int stride) {
    return hipMemcpy2DAsync(m->host + m->off[k], (size_t)m->width[k] * sizeof(float), src + (size_t)env0 * stride, (size_t)stride * sizeof(float),
                            (size_t)m->width[k] * sizeof(float), (size_t)n, hipMemcpyDeviceToHost, e->stream);
  };
  if (what & MJH_MIRROR_JOINTS) {
    HIPCHK(hipMemcpyAsync(m->host + m->off[0], e->S.time + env0, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->stream));   // time: fp64
    HIPCHK(rows(1, e->S.qpos, e->M.nqp)); HIPCHK(rows(2, e->S.qvel, e->M.nvp)); HIPCHK(rows(3, e->S.qfrc_inverse, e->M.nvp));
  }
  if (what & (MJH_MIRROR_BODIES | MJH_MIRROR_GEOMS)) {
    float* d = m->dev;
    int rc;
    {
      StateGuard guard(&e->S);
      e->S.x_xpos = d; e->S.x_xquat = d + (m->off[5] - m->off[4]); e->S.x_gpos = d + (m->off[6] - m->off[4]); e->S.x_gmat = d + (m->off[7] - m->off[4]);
      const int xf = ((what & MJH_MIRROR_BODIES) ? XF_BODY : 0) | ((what & MJH_MIRROR_GEOMS) ? XF_GEOM : 0);
      rc = launch(e, env0, n, 1, PH_FKONLY, xf);
    }
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(m->host + m->off[4], d, m->fk_floats * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  }
  HIPCHK(hipEventRecord(m->ev, e->stream));
  return MJH_OK;
}
extern "C" int mjh_mirror_wait(mjh_mirror* m) {
  if (!m) { mjh_set_error("null mirror"); return MJH_ERR_ARG; }
  HIPCHK(hipEventSynchronize(m->ev));
  return MJH_OK;
}
extern "C" const double* mjh_mirror_time(const mjh_mirror* m) { return m ? (const double*)(m->host + m->off[0]) : nullptr; }
extern "C" const float* mjh_mirror_field(const mjh_mirror* m, int which, int* row_width) {
  if (!m || which < 0 || which > 7) return nullptr;
  if (row_width) *row_width = m->width[which];
  return m->host + m->off[which];
}

extern "C" int mjh_state_stride(const mjh_engine* e) { return e ? 1 + e->M.nq + e->M.nv : 0; }
extern "C" int mjh_export_state_device(mjh_engine* e, void* d_out) {
  ENG_NOJOIN(e); if (!d_out) return MJH_ERR_ARG;
  const int stride = 1 + e->M.nq + e->M.nv;
  auto run = [&](hipStream_t st, int g0, int n) {
    const size_t total = (size_t)n * stride;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(mjh_export_kernel, dim3(blocks), dim3(256), 0, st, e->S, (float*)d_out, g0, n, e->M.nq, e->M.nv, e->M.nqp, e->M.nvp);
  };
  if (!e->forked) { run(e->stream, 0, e->nenv); HIPCHK(hipGetLastError()); return MJH_OK; }
  // Read-only and forked: every cohort exports its own range on its own stream, right behind its last step, and the
  // caller's stream waits for those exports only.  The cohorts stay forked, so publishing does not drain the step
  // pipeline.  (The cohort streams first wait for the caller's earlier work: it may still be reading d_out.)
  HIPCHK(hipEventRecord(e->ev_fork, e->stream));
  for (int g = 0; g < e->ncohort; g++) {
    const int g0 = (int)((long long)e->nenv * g / e->ncohort), g1 = (int)((long long)e->nenv * (g + 1) / e->ncohort);
    HIPCHK(hipStreamWaitEvent(e->cstream[g], e->ev_fork, 0));
    run(e->cstream[g], g0, g1 - g0);
    HIPCHK(hipEventRecord(e->ev_join[g], e->cstream[g]));
    HIPCHK(hipStreamWaitEvent(e->stream, e->ev_join[g], 0));
  }
  HIPCHK(hipGetLastError());
  return MJH_OK;
}

// debug: one fused step over all envs with s_memtime stamps at the 16 stage boundaries of the kernel;
// out[16] = mean over envs of (stamp[k] - stamp[0]) in shader-clock ticks
extern "C" int mjh_debug_stage_cycles(mjh_engine* e, int with_inverse, double* out) {
  ENG(e);
  DevBuf db;
  HIPCHK(hipMalloc(&db.p, (size_t)e->nenv * PROF_STRIDE * sizeof(long long)));
  long long* buf = (long long*)db.p;
  HIPCHK(hipMemsetAsync(buf, 0, (size_t)e->nenv * PROF_STRIDE * sizeof(long long), e->stream));
  int rc;
  {
    StateGuard guard(&e->S);
    e->S.x_prof = buf;
    const int ph = PH_STEP1 | PH_STEP2 | ((with_inverse & 1) ? PH_INV : 0);
    if ((with_inverse & 6) && e->M.big && e->split3) {
      // the launch chain of the many-body layout on all envs, stamps from its assemble (2) or integrate (4) launch
      const bool dn = e->M.dense != 0;
      rc = launch(e, 0, e->nenv, 1, ph | PH_PRE, (dn ? XF_DENSE : 0) | ((with_inverse & 2) ? XF_PROF : 0));
      if (!rc && dn) {
        HIPCHK(mjh_launch_dense(e->stream, e->nenv, e->dense_lds, e->dense_solve_lds, e->dC, e->S, 0));
      }
      if (!rc) {
        const size_t lds = (2 * (size_t)(((e->M.nv + 3) / 4) * 4) + 2 * (size_t)std::max(e->M.maxblk, 1) + 4 + 8 + 8 + 8 * (size_t)((e->M.nv + 2) / 3)) * sizeof(float);
        if (e->M.diagM) hipLaunchKernelGGL((mjh_solve_kernel<true, false>), dim3(e->nenv), dim3(64), lds, e->stream, e->dC, e->S, 0);
        else hipLaunchKernelGGL((mjh_solve_kernel<false, false>), dim3(e->nenv), dim3(64), lds, e->stream, e->dC, e->S, 0);
        HIPCHK(hipGetLastError());
        rc = launch(e, 0, e->nenv, 1, PH_STEP2 | PH_POST, (with_inverse & 4) ? XF_PROF : 0);
      }
    } else
    rc = launch(e, 0, e->nenv, 1, ph, XF_PROF);
  }
  std::vector<long long> h((size_t)e->nenv * PROF_STRIDE);
  if (!rc) { HIPCHK(hipMemcpyAsync(h.data(), buf, h.size() * sizeof(long long), hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipStreamSynchronize(e->stream)); }
  if (rc) return rc;
  for (int k = 0; k < 16; k++) out[k] = 0;
  for (int en = 0; en < e->nenv; en++) for (int k = 0; k < 16; k++) { long long v = h[(size_t)en*PROF_STRIDE+k]; out[k] += v ? (double)(v - h[(size_t)en*PROF_STRIDE]) : 0.0; }
  for (int k = 0; k < 16; k++) out[k] /= e->nenv;
  return MJH_OK;
}

// Debug: the solve launch of the many-body chain (free-body models: C2) on the pools the last step left, timed with HIP events —
// once as it is, once with every wave reading the block operands of one of `slices` environments (a working set that stays in L2):
// the difference is what the operand stream from MALL / HBM costs.  out_ms[2], out_iter[2] (mean sweeps).  The probe's results are
// garbage: the next step rebuilds the pools; state, clock and warm start are not touched (the launches write qacc only into the
// chain's hand-over slice).
extern "C" int mjh_debug_solve_probe(mjh_engine* e, int slices, int reps, double* out_ms, double* out_iter) {
  ENG(e);
  if (!(e->M.big && e->split3 && e->M.diagM)) { mjh_set_error("mjh_debug_solve_probe: free-body model in the many-body layout only"); return MJH_ERR_ARG; }
  hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
  const size_t lds = (2 * (size_t)(((e->M.nv + 3) / 4) * 4) + 2 * (size_t)std::max(e->M.maxblk, 1) + 4 + 8 + 8 + 8 * (size_t)((e->M.nv + 2) / 3)) * sizeof(float);
  std::vector<float> meta((size_t)e->nenv);
  for (int mode = 0; mode < 2; mode++) {
    StateGuard guard(&e->S);
    e->S.probe_slices = mode ? std::max(1, slices) : 0; e->S.env_order = nullptr;
    HIPCHK(hipEventRecord(a, e->stream));
    for (int r = 0; r < std::max(1, reps); r++) hipLaunchKernelGGL((mjh_solve_kernel<true, false>), dim3(e->nenv), dim3(64), lds, e->stream, e->dC, e->S, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(b, e->stream)); HIPCHK(hipEventSynchronize(b));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, a, b)); out_ms[mode] = ms / std::max(1, reps);
    double it = 0;
    HIPCHK(hipMemcpy2D(meta.data(), sizeof(float), e->S.gscratch + e->L.g_meta + 5, (size_t)e->S.gstride * sizeof(float), sizeof(float), (size_t)e->nenv, hipMemcpyDeviceToHost));
    for (int i = 0; i < e->nenv; i++) { int v; std::memcpy(&v, &meta[i], 4); it += v; }
    out_iter[mode] = it / e->nenv;
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  return MJH_OK;
}

// raw per-env stamps of one LPT-ordered step launch (debug timeline tool): out[nenv*PROF_STRIDE]
extern "C" int mjh_debug_stage_raw(mjh_engine* e, int with_inverse, long long* out) {
  ENG(e);
  DevBuf db;
  HIPCHK(hipMalloc(&db.p, (size_t)e->nenv * PROF_STRIDE * sizeof(long long)));
  long long* buf = (long long*)db.p;
  HIPCHK(hipMemsetAsync(buf, 0, (size_t)e->nenv * PROF_STRIDE * sizeof(long long), e->stream));
  int rc;
  {
    StateGuard guard(&e->S);
    e->S.x_prof = buf;
    if (e->lpt && e->d_order) {
      hipLaunchKernelGGL(mjh_order_kernel, dim3(1), dim3(1024), 0, e->stream, (const int*)e->S.stats, e->d_order, 0, e->nenv, (int*)nullptr, 0);
      e->S.env_order = e->d_order; e->order_valid = true; e->order_G = -1;
    }
    rc = launch(e, 0, e->nenv, 1, PH_STEP1 | PH_STEP2 | (with_inverse ? PH_INV : 0), XF_PROF);
  }
  if (!rc) { HIPCHK(hipMemcpyAsync(out, buf, (size_t)e->nenv * PROF_STRIDE * sizeof(long long), hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipStreamSynchronize(e->stream)); }
  return rc;
}

// debug: one full-range step launch that returns at stage boundary `stage` (1..14) without storing anything
extern "C" int mjh_debug_stop_at(mjh_engine* e, int stage, int with_inverse) {
  ENG(e);
  if (stage < 1 || stage > 14) { mjh_set_error("mjh_debug_stop_at: stage must be 1..14"); return MJH_ERR_ARG; }
  return launch(e, 0, e->nenv, 1, PH_STEP1 | PH_STEP2 | (with_inverse ? PH_INV : 0), stage << 8);
}

extern "C" int mjh_nenv(const mjh_engine* e) { return e ? e->nenv : 0; }
extern "C" const mjh_model* mjh_engine_model(const mjh_engine* e) { return e ? e->model : nullptr; }
extern "C" int mjh_lds_bytes(const mjh_engine* e) { return e ? e->lds_bytes : 0; }
extern "C" int mjh_solver_order(const mjh_engine* e) { return !e ? 0 : e->M.pgs_row_order ? 2 : e->M.patch ? 1 : 0; }
extern "C" int mjh_pgs_schedule(const mjh_engine* e) { return !e ? 0 : e->M.pgs_row_order; }
extern "C" int mjh_patch_sweep(const mjh_engine* e) { return !e ? 0 : e->M.patch; }
extern "C" int mjh_window_solver(const mjh_engine* e) { return !e ? 0 : (e->M.window && e->S.wbuf ? 1 : 0); }
extern "C" int mjh_dense_solver(const mjh_engine* e) { return e && e->M.big && e->split3 && e->M.dense ? 1 : 0; }
extern "C" const char* mjh_version(void) { return "mjhip 0.1 (gfx950)"; }
