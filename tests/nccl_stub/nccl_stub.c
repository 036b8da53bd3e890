/* nccl_stub.c — TEST INFRASTRUCTURE, not product: a recording stand-in for the handful of RCCL entry points csrc/group.hip resolves with
 * dlopen (ncclCommInitAll / ncclCommDestroy / ncclCommAbort / ncclAllGather / ncclGroupStart / ncclGroupEnd / ncclGetErrorString), named
 * by MJH_RCCL_LIB.  It lets the N-rank exchange of mjh_group_publish (one RCCL all-gather of the published slice per rank
 * and publish) run where no 8-GPU node exists:
 *   - on the CPU box it records every call (order, rank, count, dtype, buffers, stream, calling thread, group nesting), so the tests assert
 *     the grouped sequence GroupStart, 8 x AllGather, GroupEnd and the per-thread sequence for N = 8 (tests/test_rccl_sequence.py);
 *   - on the GPU box (NCCL_STUB_COPY=1) it also PERFORMS the all-gather between N ranks that live on one device, with the stream semantics
 *     of the real call (each rank's result is ordered on that rank's stream behind every rank's send buffer), so that the gathered state
 *     of eight shards through this path can be compared bitwise with the peer-copy transport (tests/test_gpu_round6.py).
 * Nothing of RCCL's implementation is restated here: an all-gather is N x N device copies. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MAXR 64
#define MAXLOG 8192

typedef struct stub_comm { int rank, nranks, dev, alive, aborted; } stub_comm;
typedef struct stub_rec {
  int seq, rank, nranks, dtype, in_group, group_id, epoch;
  unsigned long count;
  unsigned long long send, recv, stream, comm, tid;
} stub_rec;

static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
static stub_rec logv[MAXLOG];
static int nlog, n_init, n_destroy, n_abort, n_gstart, n_gend, init_ndev, fail_rank = -1, group_ids;
static int init_devs[MAXR];
static __thread int depth;            /* ncclGroupStart nesting of the calling thread */
static __thread int tl_group_id;
static __thread int tl_ncall;
static __thread struct { stub_comm* c; const void* send; void* recv; size_t count; void* stream; } tl_call[MAXR];

/* copy mode: HIP resolved from the process (libmjhip.so has loaded libamdhip64) */
static int copy_mode = -1;
static int (*p_hipEventCreateWithFlags)(void**, unsigned);
static int (*p_hipEventRecord)(void*, void*);
static int (*p_hipStreamWaitEvent)(void*, void*, unsigned);
static int (*p_hipMemcpyAsync)(void*, const void*, size_t, int, void*);
static void* ev[MAXR];
static int hip_err, hip_where;     /* first failing HIP call of the copy mode: its code and which one (1 event create, 2 record, 3 wait, 4 copy, 5 barrier timeout) */
#define HIPTRY(w, call) do { int e_ = (call); if (e_) { if (!hip_err) { hip_err = e_; hip_where = (w); } return 1; } } while (0)
static struct { const void* send; size_t count; } pend[MAXR];
static int arrived, epoch;

static pthread_once_t copy_once = PTHREAD_ONCE_INIT;
static void copy_init(void) {
  const char* e = getenv("NCCL_STUB_COPY");
  int mode = 0;
  if (e && atoi(e)) {
    /* the HIP runtime the process already holds (libmjhip.so's): never a second copy */
    void* h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("libamdhip64.so.6", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = RTLD_DEFAULT;
    p_hipEventCreateWithFlags = (int (*)(void**, unsigned))dlsym(h, "hipEventCreateWithFlags");
    p_hipEventRecord = (int (*)(void*, void*))dlsym(h, "hipEventRecord");
    p_hipStreamWaitEvent = (int (*)(void*, void*, unsigned))dlsym(h, "hipStreamWaitEvent");
    p_hipMemcpyAsync = (int (*)(void*, const void*, size_t, int, void*))dlsym(h, "hipMemcpyAsync");
    if (p_hipEventCreateWithFlags && p_hipEventRecord && p_hipStreamWaitEvent && p_hipMemcpyAsync) mode = 1;
  }
  copy_mode = mode;
}
static int copying(void) { pthread_once(&copy_once, copy_init); return copy_mode; }     /* (eight threads ask at once) */
static int ready_event(int rank, void* stream) {
  if (!ev[rank]) HIPTRY(1, p_hipEventCreateWithFlags(&ev[rank], 2u /* hipEventDisableTiming */));
  HIPTRY(2, p_hipEventRecord(ev[rank], stream));
  return 0;
}
static int gather_into(int nranks, void* recv, size_t count, void* stream, const void* const* sends) {
  for (int r = 0; r < nranks; r++) {
    HIPTRY(3, p_hipStreamWaitEvent(stream, ev[r], 0));
    HIPTRY(4, p_hipMemcpyAsync((char*)recv + (size_t)r * count * 4, sends[r], count * 4, 3 /* hipMemcpyDeviceToDevice */, stream));
  }
  return 0;
}

int ncclCommInitAll(stub_comm** comms, int ndev, const int* devs) {
  pthread_mutex_lock(&mu);
  n_init++; init_ndev = ndev;
  for (int k = 0; k < ndev && k < MAXR; k++) init_devs[k] = devs ? devs[k] : k;
  pthread_mutex_unlock(&mu);
  if (ndev <= 0 || ndev > MAXR) return 4; /* ncclInvalidArgument */
  for (int k = 0; k < ndev; k++) {
    comms[k] = (stub_comm*)calloc(1, sizeof(stub_comm));
    comms[k]->rank = k; comms[k]->nranks = ndev; comms[k]->dev = devs ? devs[k] : k; comms[k]->alive = 1;
  }
  return 0;
}
int ncclCommDestroy(stub_comm* c) { pthread_mutex_lock(&mu); n_destroy++; pthread_mutex_unlock(&mu); if (c) { c->alive = 0; free(c); } return 0; }
int ncclCommAbort(stub_comm* c) { pthread_mutex_lock(&mu); n_abort++; pthread_mutex_unlock(&mu); if (c) { c->aborted = 1; free(c); } return 0; }
const char* ncclGetErrorString(int r) { return r == 0 ? "no error" : (r == 1 ? "unhandled device error (injected by the stub)" : "stub error"); }

int ncclGroupStart(void) {
  pthread_mutex_lock(&mu);
  n_gstart++;
  if (depth == 0) { tl_group_id = ++group_ids; tl_ncall = 0; }
  pthread_mutex_unlock(&mu);
  depth++;
  return 0;
}
int ncclGroupEnd(void) {
  pthread_mutex_lock(&mu); n_gend++; pthread_mutex_unlock(&mu);
  if (depth <= 0) return 5;
  if (--depth > 0) return 0;
  int rc = 0;
  if (copying() && tl_ncall > 0) {
    /* the deferred calls of the group: every rank's send buffer is ready where its stream stands now */
    const void* sends[MAXR] = {0};
    const int nr = tl_call[0].c->nranks;
    if (tl_ncall != nr) return 5;        /* a grouped all-gather must name every rank */
    for (int i = 0; i < tl_ncall && !rc; i++) { sends[tl_call[i].c->rank] = tl_call[i].send; rc = ready_event(tl_call[i].c->rank, tl_call[i].stream); }
    for (int i = 0; i < tl_ncall && !rc; i++) rc = gather_into(nr, tl_call[i].recv, tl_call[i].count, tl_call[i].stream, sends);
  }
  tl_ncall = 0;
  return rc ? 1 : 0;
}
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, stub_comm* c, void* stream) {
  if (!c) return 4;
  pthread_mutex_lock(&mu);
  int my_epoch = epoch;
  if (nlog < MAXLOG) {
    stub_rec* r = &logv[nlog];
    r->seq = nlog; r->rank = c->rank; r->nranks = c->nranks; r->dtype = dtype; r->in_group = depth > 0; r->group_id = depth > 0 ? tl_group_id : 0; r->epoch = my_epoch;
    r->count = (unsigned long)count; r->send = (unsigned long long)(uintptr_t)send; r->recv = (unsigned long long)(uintptr_t)recv;
    r->stream = (unsigned long long)(uintptr_t)stream; r->comm = (unsigned long long)(uintptr_t)c; r->tid = (unsigned long long)(uintptr_t)pthread_self();
    nlog++;
  }
  const int fail = c->rank == fail_rank;
  pthread_mutex_unlock(&mu);
  if (fail) return 1;
  if (dtype != 7) return 4;
  if (depth > 0) {
    if (tl_ncall < MAXR) { tl_call[tl_ncall].c = c; tl_call[tl_ncall].send = send; tl_call[tl_ncall].recv = recv; tl_call[tl_ncall].count = count; tl_call[tl_ncall].stream = stream; tl_ncall++; }
    return 0;
  }
  if (!copying()) return 0;
  /* one thread per rank: like the real call, nothing is enqueued before every rank has joined (10 s: a missing partner fails the test
   * instead of hanging it) */
  int rc = 0;
  const void* sends[MAXR];
  pthread_mutex_lock(&mu);
  rc = ready_event(c->rank, stream);
  pend[c->rank].send = send; pend[c->rank].count = count;
  my_epoch = epoch;
  if (++arrived == c->nranks) { arrived = 0; epoch++; pthread_cond_broadcast(&cv); }
  else {
    struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); ts.tv_sec += 10;
    while (epoch == my_epoch && !rc) if (pthread_cond_timedwait(&cv, &mu, &ts)) { rc = 1; if (!hip_err) { hip_err = -1; hip_where = 5; } }
  }
  for (int r = 0; r < c->nranks; r++) sends[r] = pend[r].send;
  pthread_mutex_unlock(&mu);
  if (!rc) rc = gather_into(c->nranks, recv, count, stream, sends);
  return rc ? 1 : 0;
}

/* ---- what the tests read */
void stub_reset(void) { pthread_mutex_lock(&mu); nlog = n_init = n_destroy = n_abort = n_gstart = n_gend = init_ndev = 0; fail_rank = -1; pthread_mutex_unlock(&mu); }
void stub_fail_rank(int rank) { pthread_mutex_lock(&mu); fail_rank = rank; pthread_mutex_unlock(&mu); }
int stub_nlog(void) { return nlog; }
int stub_get(int i, stub_rec* out) { if (i < 0 || i >= nlog) return 1; *out = logv[i]; return 0; }
void stub_counters(int* out) { out[0] = n_init; out[1] = n_destroy; out[2] = n_abort; out[3] = n_gstart; out[4] = n_gend; out[5] = init_ndev; out[6] = copying(); out[7] = epoch; }
void stub_hip_error(int* out) { out[0] = hip_err; out[1] = hip_where; }
int stub_init_dev(int k) { return k >= 0 && k < MAXR ? init_devs[k] : -1; }
