// comm_api.hip -- the multi-GPU exchange of the front-end behind the C ABI (include/urf.h, "exchange").
// The reference is single-GPU; its caller Tracking::ExtractFeatureAndMatch (src/tracking.cc:338-377) is what
// shards here: one rank (process, or device of one process) per GPU, frames of a batch block-distributed,
// SuperPoint locally, ONE all-gather of fixed-size feature slots (RCCL over xGMI), every rank matches the
// pairs whose second frame it owns, match lists gathered to rank 0 where the serial tracker lives.
//
// RCCL is loaded with dlopen at the first communicator: liburf_front.so itself links only the HIP runtime, a
// single-GPU user never maps librccl.
#include "../../include/urf.h"
#include "urf_common.h"

#include <dlfcn.h>
#include <string.h>

#include <deque>
#include <memory>
#include <mutex>
#include <vector>

// The few RCCL types this file needs, declared here (values as in <rccl/rccl.h> of ROCm 7.2: ncclSuccess = 0,
// ncclChar = 0, a 128-byte unique id): a single-GPU build of the library does not need the RCCL headers.
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[URF_COMM_ID_BYTES]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
static const ncclResult_t ncclSuccess = 0;
static const ncclDataType_t ncclChar = 0;

namespace urf {
namespace {

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (g_rccl.lib) return 0;
  void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  URF_CHECK(lib, "cannot load librccl.so: %s", dlerror());
  bool ok = true;
  auto sym = [&](const char *name) { void *p = dlsym(lib, name); ok = ok && p; return p; };
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
  g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
  g_rccl.Send = (decltype(g_rccl.Send))sym("ncclSend");
  g_rccl.Recv = (decltype(g_rccl.Recv))sym("ncclRecv");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
  URF_CHECK(ok, "librccl.so lacks an expected nccl* symbol");
  g_rccl.lib = lib;
  return 0;
}

#define URF_NCCL(call)                                                                              \
  do {                                                                                              \
    ncclResult_t r_ = (call);                                                                       \
    if (r_ != ncclSuccess) {                                                                        \
      urf::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r_));       \
      return -1;                                                                                    \
    }                                                                                               \
  } while (0)

}  // namespace
}  // namespace urf
using namespace urf;

// Loopback world: one process acts as `world` logical ranks on ONE device (urf_comm_init_loopback).  A collective is a
// rendezvous: every rank's call queues its arguments and records an event on its stream; the call that completes the head
// of every rank's queue issues the device copies -- on each receiver's stream, behind every sender's event -- and then makes
// every rank's stream wait for all copies (a rank may reuse its send buffer after the collective, as with RCCL).  Host calls
// may come in any rank order from one thread; every rank makes the same sequence of collectives (also as with RCCL).
struct LoopHub {
  int world = 0, device = 0;
  std::mutex mu;
  // a queued call carries its OWN "my send buffer is ready" event, recorded on the stream of that call: a rank may issue two
  // collectives on different streams before its peers arrive, and each stays ordered behind its own producer stream
  struct Arg { int kind; const void *src; void *dst; size_t bytes; int root; hipStream_t st; hipEvent_t ready; };
  std::vector<std::deque<Arg>> q;          // pending calls per rank, in call order
  std::vector<hipEvent_t> ev_out;          // per rank: "my copies are done" (recorded and waited for inside one rendezvous)
  std::vector<hipEvent_t> pool;            // free "ready" events
  long collectives = 0;
  int take(hipEvent_t *e) {
    if (!pool.empty()) { *e = pool.back(); pool.pop_back(); return 0; }
    URF_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return 0;
  }
  ~LoopHub() {
    for (auto &d : q) for (Arg &a : d) (void)hipEventDestroy(a.ready);
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_out) (void)hipEventDestroy(e);
  }
};

struct urf_comm {
  int world = 1, rank = 0, device = 0;
  ncclComm_t nccl = nullptr;     // null: a world of one without RCCL, or a loopback rank
  std::shared_ptr<LoopHub> hub;  // loopback world
};

extern "C" int urf_comm_unique_id(void *id) {
  URF_CHECK(id, "urf_comm_unique_id: null");
  if (rccl_load()) return -1;
  ncclUniqueId u;
  URF_NCCL(g_rccl.GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return 0;
}

extern "C" int urf_comm_init(int world, int rank, int device, const void *id, urf_comm **out) {
  URF_CHECK(out && world >= 1 && rank >= 0 && rank < world, "urf_comm_init: rank %d of %d", rank, world);
  URF_CHECK(world == 1 || id, "urf_comm_init: a world of %d ranks needs the unique id of rank 0 (urf_comm_unique_id)", world);
  int ndev = 0;
  URF_HIP(hipGetDeviceCount(&ndev));
  URF_CHECK(device >= 0 && device < ndev, "device %d out of range (%d devices)", device, ndev);
  urf_comm *c = new urf_comm();
  c->world = world; c->rank = rank; c->device = device;
  if (id) {   // also for a world of one: the same RCCL code path, on one rank
    if (rccl_load()) { delete c; return -1; }
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    if (hipSetDevice(device) != hipSuccess || g_rccl.CommInitRank(&c->nccl, world, u, rank) != ncclSuccess) {
      urf::set_error("ncclCommInitRank(world %d, rank %d, device %d) failed", world, rank, device);
      delete c;
      return -1;
    }
  }
  *out = c;
  return 0;
}

extern "C" int urf_comm_init_all(int ndev, const int *devices, urf_comm **out) {
  URF_CHECK(out && devices && ndev >= 1 && ndev <= 64, "urf_comm_init_all: bad argument");
  if (rccl_load()) return -1;
  std::vector<ncclComm_t> comms((size_t)ndev);
  URF_NCCL(g_rccl.CommInitAll(comms.data(), ndev, devices));
  for (int i = 0; i < ndev; ++i) {
    out[i] = new urf_comm();
    out[i]->world = ndev; out[i]->rank = i; out[i]->device = devices[i]; out[i]->nccl = comms[i];
  }
  return 0;
}

// One host thread that drives several devices' communicators (urf_comm_init_all) brackets the collective calls of a step with
// these: RCCL then launches them together (ncclGroupStart / ncclGroupEnd) -- issued one by one, the first rank's call would wait
// for peers the thread has not called yet.  Not needed with one rank per process, nor in a loopback world (no-ops without RCCL).
extern "C" int urf_comm_group_start(void) {
  if (!g_rccl.lib) return 0;
  URF_NCCL(g_rccl.GroupStart());
  return 0;
}
extern "C" int urf_comm_group_end(void) {
  if (!g_rccl.lib) return 0;
  URF_NCCL(g_rccl.GroupEnd());
  return 0;
}

extern "C" int urf_comm_init_loopback(int world, int device, urf_comm **out) {
  URF_CHECK(out && world >= 1 && world <= 64, "urf_comm_init_loopback: world %d outside [1, 64]", world);
  int ndev = 0;
  URF_HIP(hipGetDeviceCount(&ndev));
  URF_CHECK(device >= 0 && device < ndev, "device %d out of range (%d devices)", device, ndev);
  URF_HIP(hipSetDevice(device));
  auto hub = std::make_shared<LoopHub>();
  hub->world = world; hub->device = device;
  hub->q.resize(world);
  hub->ev_out.resize(world);
  for (int r = 0; r < world; ++r) URF_HIP(hipEventCreateWithFlags(&hub->ev_out[r], hipEventDisableTiming));
  for (int r = 0; r < world; ++r) {
    out[r] = new urf_comm();
    out[r]->world = world; out[r]->rank = r; out[r]->device = device; out[r]->hub = hub;
  }
  return 0;
}

// a rank's call of a loopback collective (kind 0 = all-gather, 1 = gather).  Returns 0, or < 0 with the error text set.
static int loop_collective(urf_comm *c, int kind, const void *src, void *dst, size_t bytes, int root, hipStream_t st) {
  LoopHub &h = *c->hub;
  std::lock_guard<std::mutex> lock(h.mu);
  hipEvent_t ready = nullptr;
  if (h.take(&ready)) return -1;
  URF_HIP(hipEventRecord(ready, st));
  h.q[c->rank].push_back({kind, src, dst, bytes, root, st, ready});
  for (;;) {
    for (int r = 0; r < h.world; ++r)
      if (h.q[r].empty()) return 0;
    std::vector<LoopHub::Arg> set((size_t)h.world);
    for (int r = 0; r < h.world; ++r) { set[r] = h.q[r].front(); h.q[r].pop_front(); }
    // (the events go back to the pool whatever happens below: a wait already enqueued keeps the record it was given)
    struct Back { LoopHub &h; std::vector<LoopHub::Arg> &set; ~Back() { for (auto &a : set) h.pool.push_back(a.ready); } } back{h, set};
    for (int r = 1; r < h.world; ++r)
      URF_CHECK(set[r].kind == set[0].kind && set[r].bytes == set[0].bytes && set[r].root == set[0].root,
                "loopback ranks 0 and %d disagree on a collective (kind %d/%d, %zu/%zu bytes, root %d/%d)", r, set[0].kind,
                set[r].kind, set[0].bytes, set[r].bytes, set[0].root, set[r].root);
    const bool gather = set[0].kind == 1;
    const size_t nb = set[0].bytes;
    const int rt = set[0].root;
    // copies on the receivers' streams, behind every sender's event
    for (int q = 0; q < h.world; ++q) {
      if (gather && q != rt) continue;
      URF_CHECK(set[q].dst, "loopback collective: rank %d has no receive buffer", q);
      for (int r = 0; r < h.world; ++r) URF_HIP(hipStreamWaitEvent(set[q].st, set[r].ready, 0));
      for (int r = 0; r < h.world; ++r) {
        char *to = (char *)set[q].dst + (size_t)r * nb;
        if ((const void *)to != set[r].src) URF_HIP(hipMemcpyAsync(to, set[r].src, nb, hipMemcpyDeviceToDevice, set[q].st));
      }
      URF_HIP(hipEventRecord(h.ev_out[q], set[q].st));
    }
    for (int r = 0; r < h.world; ++r)
      for (int q = 0; q < h.world; ++q)
        if ((!gather || q == rt) && q != r) URF_HIP(hipStreamWaitEvent(set[r].st, h.ev_out[q], 0));
    h.collectives += 1;
  }
}

extern "C" void urf_comm_destroy(urf_comm *c) {
  if (!c) return;
  if (c->nccl) { (void)hipSetDevice(c->device); (void)g_rccl.CommDestroy(c->nccl); }
  delete c;
}

extern "C" int urf_comm_world(const urf_comm *c) { return c ? c->world : 0; }
extern "C" int urf_comm_rank(const urf_comm *c) { return c ? c->rank : -1; }

// d_all[world][nslots] <- every rank's d_local[nslots]; asynchronous on `stream`
extern "C" int urf_comm_allgather_slots(urf_comm *c, const void *d_local, int nslots, void *d_all, void *stream) {
  URF_CHECK(c && d_local && d_all && nslots >= 1, "urf_comm_allgather_slots: bad argument");
  URF_HIP(hipSetDevice(c->device));
  const size_t bytes = (size_t)nslots * urf_slot_bytes();
  if (c->hub) return loop_collective(c, 0, d_local, d_all, bytes, -1, (hipStream_t)stream);
  if (!c->nccl) {
    if (d_all != d_local) URF_HIP(hipMemcpyAsync(d_all, d_local, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
  }
  URF_NCCL(g_rccl.AllGather(d_local, d_all, bytes, ncclChar, c->nccl, (hipStream_t)stream));
  return 0;
}

// rank `root` receives every rank's `bytes` at d_recv + r * bytes (the others may pass d_recv = NULL)
extern "C" int urf_comm_gather(urf_comm *c, const void *d_send, size_t bytes, void *d_recv, int root, void *stream) {
  URF_CHECK(c && d_send && bytes > 0 && root >= 0 && root < c->world, "urf_comm_gather: bad argument");
  URF_CHECK(c->rank != root || d_recv, "urf_comm_gather: the root needs a receive buffer");
  URF_HIP(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  if (c->hub) return loop_collective(c, 1, d_send, d_recv, bytes, root, st);
  if (!c->nccl) {
    if (d_recv != d_send) URF_HIP(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, st));
    return 0;
  }
  URF_NCCL(g_rccl.GroupStart());
  if (c->rank == root)
    for (int r = 0; r < c->world; ++r) URF_NCCL(g_rccl.Recv((char *)d_recv + (size_t)r * bytes, bytes, ncclChar, r, c->nccl, st));
  URF_NCCL(g_rccl.Send(d_send, bytes, ncclChar, root, c->nccl, st));
  URF_NCCL(g_rccl.GroupEnd());
  return 0;
}

// The pairs rank `rank` matches in one step of `per_rank` frames per rank: global frame g = rank * per_rank + j is
// matched against its predecessor g - 1 (both as indices into the gathered slots); -1 = the last frame of the
// previous step, which the caller carries over (urf.h).
extern "C" int urf_comm_plan_pairs(int world, int rank, int per_rank, int *first, int *second) {
  URF_CHECK(first && second && world >= 1 && rank >= 0 && rank < world && per_rank >= 1, "urf_comm_plan_pairs: bad argument");
  for (int j = 0; j < per_rank; ++j) {
    const int g = rank * per_rank + j;
    first[j] = g - 1;
    second[j] = g;
  }
  return 0;
}
