// comm.cpp — the path's one exchange step over RCCL (include/jn_stereo.h, "cross-rig merge").  Product code.
//
// Per-rig obstacle scans -> robot-level scan: element-wise MIN over the bins (point_cloud.cpp:264-266 applied
// across rigs) and min / max / min / max of the LaserScan extrema (:255-260).  One process per GPU; the merge of a
// whole batch is ONE ncclAllReduce(ncclMin, ncclDouble) on a packed buffer (maxima negated for the trip) — a few tens
// of KB, latency-bound over xGMI, so the number of collectives is what matters, not the link rate.
//
// RCCL is bound at run time: dlopen by soname returns the copy a host process already mapped (PyTorch ships its own
// librccl.so.1), so a Python/torch host and this library talk to one RCCL; a plain C/C++ host gets /opt/rocm/lib's.
#include "../../include/jn_stereo.h"
#include "kernels.h"

#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <mutex>
#include <vector>

#include <rccl/rccl.h>

using namespace jnav;

struct jn_comm;
namespace jnav {
jn_status comm_merge_async(jn_comm* c, int n, int bins, double* dBins, double* dMeta, hipEvent_t ready, hipEvent_t done, double* packed);
jn_status comm_merge_identity(jn_comm* c, int n, int bins);
void comm_abort(jn_comm* c);
bool comm_dead(const jn_comm* c);
int comm_device(const jn_comm* c);
}

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;            // optional
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {getenv("JN_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) { fprintf(stderr, "libjn_stereo: cannot load librccl.so.1 (%s)\n", dlerror()); return; }
#define JN_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name))
    JN_SYM(GetUniqueId, "ncclGetUniqueId"); JN_SYM(CommInitRank, "ncclCommInitRank"); JN_SYM(CommDestroy, "ncclCommDestroy");
    JN_SYM(CommCount, "ncclCommCount"); JN_SYM(CommCuDevice, "ncclCommCuDevice"); JN_SYM(CommUserRank, "ncclCommUserRank");
    JN_SYM(AllReduce, "ncclAllReduce"); JN_SYM(GetErrorString, "ncclGetErrorString"); JN_SYM(CommAbort, "ncclCommAbort");
#undef JN_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.CommCuDevice && r.CommUserRank && r.AllReduce &&
           r.GetErrorString;
    if (!r.ok) fprintf(stderr, "libjn_stereo: librccl lacks an expected symbol\n");
  });
  return r.ok ? &r : nullptr;
}

#define RCCL_TRY(R, expr)                                                                              \
  do {                                                                                                 \
    ncclResult_t e__ = (expr);                                                                         \
    if (e__ != ncclSuccess) {                                                                          \
      fprintf(stderr, "libjn_stereo: %s failed: %s (%s:%d)\n", #expr, (R)->GetErrorString(e__), __FILE__, __LINE__); \
      return JN_ERR_COMM;                                                                              \
    }                                                                                                  \
  } while (0)
#define HIP_TRY_C(expr)                                                                                \
  do {                                                                                                 \
    hipError_t e__ = (expr);                                                                           \
    if (e__ != hipSuccess) {                                                                           \
      fprintf(stderr, "libjn_stereo: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return JN_ERR_NO_DEVICE;                                                                         \
    }                                                                                                  \
  } while (0)

}  // namespace

struct jn_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  double* flat = nullptr;          // packed [n*bins | n*4] doubles, grow-only
  size_t cap = 0;
  double* h_inf = nullptr;         // pinned, every element +inf: what a failed batch feeds the reduction (comm_merge_identity), grow-only
  size_t inf_cap = 0;
  // Buffers a growth replaced.  They are NOT freed while the communicator lives: hipFree waits for the device, and the stream may be
  // inside a collective whose peer is gone — the one wait that must never happen under `m` (comm_abort needs it).  jn_comm_destroy frees them.
  std::vector<void*> retired_dev, retired_host;
  std::mutex m;
  std::atomic<bool> dead{false};   // aborted after a merge that did not complete (comm_abort): every later collective returns JN_ERR_COMM
};

extern "C" {

jn_status jn_comm_unique_id(uint8_t id[JN_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == JN_COMM_ID_BYTES, "ncclUniqueId size");
  if (!id) return JN_ERR_INVALID;
  Rccl* R = rccl();
  if (!R) return JN_ERR_COMM;
  ncclUniqueId u;
  RCCL_TRY(R, R->GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return JN_OK;
}

jn_status jn_comm_create(const uint8_t id[JN_COMM_ID_BYTES], int32_t rank, int32_t world, int32_t device, jn_comm** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return JN_ERR_INVALID;
  *out = nullptr;
  Rccl* R = rccl();
  if (!R) return JN_ERR_COMM;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return JN_ERR_NO_DEVICE;
  HIP_TRY_C(hipSetDevice(device));
  jn_comm* c = new jn_comm();
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclResult_t e = R->CommInitRank(&c->comm, world, u, rank);
  if (e != ncclSuccess) {
    fprintf(stderr, "libjn_stereo: ncclCommInitRank(rank %d of %d, device %d) failed: %s\n", rank, world, device, R->GetErrorString(e));
    delete c;
    return JN_ERR_COMM;
  }
  // JN_COMM_PRIORITY=1: the communicator's stream at the highest priority.  Measured on one MI355X (bench.py --force-merge,
  // profiles/r03_merge_in_worker.txt): it does NOT help — scan -> merged bins 0.94 ms against 0.66 ms on an ordinary stream.
  int least = 0, greatest = 0;
  hipError_t se = hipErrorUnknown;
  if (getenv("JN_COMM_PRIORITY") && atoi(getenv("JN_COMM_PRIORITY")) != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
    se = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest);
  if (se != hipSuccess) se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (se != hipSuccess) { R->CommDestroy(c->comm); delete c; return JN_ERR_NO_DEVICE; }
  *out = c;
  return JN_OK;
}

jn_status jn_comm_info(jn_comm* c, int32_t* rank, int32_t* world, int32_t* device) {
  if (!c) return JN_ERR_INVALID;
  Rccl* R = rccl();
  if (!R) return JN_ERR_COMM;
  int r = 0, w = 0, d = 0;
  RCCL_TRY(R, R->CommUserRank(c->comm, &r));
  RCCL_TRY(R, R->CommCount(c->comm, &w));
  RCCL_TRY(R, R->CommCuDevice(c->comm, &d));
  if (rank) *rank = r;
  if (world) *world = w;
  if (device) *device = d;
  return JN_OK;
}

// Room for `count` doubles in c->flat.  Called under c->m; never waits: the old buffer may still be read by a merge queued earlier (it is
// retired, not freed), and nothing here synchronises with a stream that may be stuck in a collective.
static jn_status grow_flat(jn_comm* c, size_t count) {
  if (count <= c->cap) return JN_OK;
  double* fresh = nullptr;
  HIP_TRY_C(hipMalloc(reinterpret_cast<void**>(&fresh), count * sizeof(double)));
  if (c->flat) c->retired_dev.push_back(c->flat);
  c->flat = fresh; c->cap = count;
  return JN_OK;
}

// pack -> all-reduce -> unpack queued on the communicator's stream.  `ready` (may be null) is an event the inputs are
// complete behind; `done` (may be null) is recorded behind the unpack.  Every collective of a communicator goes through
// here under its mutex and onto its ONE stream, so all ranks execute them in the order they were queued.
static jn_status queue_merge(jn_comm* c, Rccl* R, int n, int bins, double* dBins, double* dMeta, hipEvent_t ready, hipEvent_t done, double* packed = nullptr) {
  std::lock_guard<std::mutex> guard(c->m);
  if (c->dead.load()) return JN_ERR_COMM;
  HIP_TRY_C(hipSetDevice(c->device));
  const size_t count = (size_t)n * (bins + 4);
  if (packed) {                                              // the caller's own packed buffer (k_scan_finish wrote it): reduce it in place, unpack
    if (ready) HIP_TRY_C(hipStreamWaitEvent(c->stream, ready, 0));
    RCCL_TRY(R, R->AllReduce(packed, packed, count, ncclDouble, ncclMin, c->comm, c->stream));
    launch_scan_pack(c->stream, n, bins, dBins, dMeta, packed, false);
    if (done) HIP_TRY_C(hipEventRecord(done, c->stream));
    return JN_OK;
  }
  if (const jn_status gs = grow_flat(c, count); gs != JN_OK) return gs;
  if (ready) HIP_TRY_C(hipStreamWaitEvent(c->stream, ready, 0));
  launch_scan_pack(c->stream, n, bins, dBins, dMeta, c->flat, true);
  RCCL_TRY(R, R->AllReduce(c->flat, c->flat, count, ncclDouble, ncclMin, c->comm, c->stream));
  launch_scan_pack(c->stream, n, bins, dBins, dMeta, c->flat, false);
  if (done) HIP_TRY_C(hipEventRecord(done, c->stream));
  return JN_OK;
}

// Wait for everything queued on the communicator's stream, at most JN_COMM_TIMEOUT_MS (default 30 s, 0 = for ever): a peer that never
// joins must not hang this rank — the communicator is aborted instead and the call reports JN_ERR_COMM.  Called WITHOUT c->m.
static jn_status bounded_stream_wait(jn_comm* c) {
  int timeout_ms = 30000;
  if (const char* e = getenv("JN_COMM_TIMEOUT_MS")) timeout_ms = atoi(e);
  if (timeout_ms <= 0) { HIP_TRY_C(hipStreamSynchronize(c->stream)); }
  else {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) { HIP_TRY_C(q); }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) { comm_abort(c); return JN_ERR_COMM; }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  if (c->dead.load()) return JN_ERR_COMM;                    // another thread aborted meanwhile: what completed was the aborted kernels, not a reduction
  return JN_OK;
}

jn_status jn_scan_allreduce(jn_comm* c, int32_t n, int32_t bins, double* dBins, double* dMeta) {
  if (!c || n < 1 || bins < 1 || !dBins || !dMeta) return JN_ERR_INVALID;
  if (c->dead.load()) return JN_ERR_COMM;
  Rccl* R = rccl();
  if (!R) return JN_ERR_COMM;
  const jn_status st = queue_merge(c, R, n, bins, dBins, dMeta, nullptr, nullptr);
  if (st != JN_OK) return st;
  const jn_status ws = bounded_stream_wait(c);
  if (ws != JN_OK) return ws;
  HIP_TRY_C(hipGetLastError());
  return JN_OK;
}

extern "C++" {
namespace jnav {
// The batch pipeline's form (jn_elas_set_comm): nothing waits on the host; the slot's stream continues behind `done`.
// The flat buffer is shared by consecutive merges: they run one after the other on the communicator's stream.
jn_status comm_merge_async(jn_comm* c, int n, int bins, double* dBins, double* dMeta, hipEvent_t ready, hipEvent_t done, double* packed) {
  if (!c || n < 1 || bins < 1 || !dBins || !dMeta) return JN_ERR_INVALID;
  Rccl* R = rccl();
  if (!R) return JN_ERR_COMM;
  return queue_merge(c, R, n, bins, dBins, dMeta, ready, done, packed);
}
// A batch that failed on this rank before its merge: take part in the collective with the identity of MIN (+inf everywhere), so that
// the other ranks, which are already inside or about to enter the same all-reduce, get the remaining rigs' scan instead of
// waiting for ever.  Synchronous (the failure path is not a hot path).
jn_status comm_merge_identity(jn_comm* c, int n, int bins) {
  if (!c || n < 1 || bins < 1) return JN_ERR_INVALID;
  Rccl* R = rccl();
  if (!R) return JN_ERR_COMM;
  const size_t count = (size_t)n * (bins + 4);
  {
    std::lock_guard<std::mutex> guard(c->m);
    if (c->dead.load()) return JN_ERR_COMM;
    HIP_TRY_C(hipSetDevice(c->device));
    if (const jn_status gs = grow_flat(c, count); gs != JN_OK) return gs;
    if (count > c->inf_cap) {                                // the +inf source: pinned and persistent, so the fill below can be asynchronous
      double* fresh = nullptr;
      HIP_TRY_C(hipHostMalloc(reinterpret_cast<void**>(&fresh), count * sizeof(double), hipHostMallocDefault));
      for (size_t i = 0; i < count; i++) fresh[i] = INFINITY;
      if (c->h_inf) c->retired_host.push_back(c->h_inf);
      c->h_inf = fresh; c->inf_cap = count;
    }
    // ON the communicator's stream: ordered behind any in-place all-reduce on c->flat that is still running there (a null-stream copy
    // is not ordered against a non-blocking stream and could overwrite partially reduced chunks)
    HIP_TRY_C(hipMemcpyAsync(c->flat, c->h_inf, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(R, R->AllReduce(c->flat, c->flat, count, ncclDouble, ncclMin, c->comm, c->stream));
  }
  // bounded like every other wait on the communicator (the peer this rank is feeding may itself be gone), and outside c->m so that an
  // abort from another slot's worker is not locked out
  return bounded_stream_wait(c);
}
// A collective that does not complete (a peer died or never joined): ncclCommAbort ends the kernel this rank is stuck in and
// releases the communicator's resources; the handle stays allocated (jn_comm_destroy frees it) but is dead from here on.
void comm_abort(jn_comm* c) {
  if (!c || c->dead.exchange(true)) return;
  Rccl* R = rccl();
  fprintf(stderr, "libjn_stereo: cross-rig merge on rank %d of %d did not complete in time: aborting the communicator\n", c->rank, c->world);
  // under c->m: queue_merge / comm_merge_identity check `dead` and use c->comm under the same mutex, so none of them can be between
  // its check and its ncclAllReduce while the communicator goes away (enqueueing a collective does not block, the lock is short)
  std::lock_guard<std::mutex> guard(c->m);
  if (R && R->CommAbort && c->comm) { R->CommAbort(c->comm); c->comm = nullptr; }
}
bool comm_dead(const jn_comm* c) { return c && c->dead.load(); }
int comm_device(const jn_comm* c) { return c ? c->device : -1; }
}  // namespace jnav
}  // extern "C++"

void jn_comm_destroy(jn_comm* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream && !c->dead.load()) { hipStreamSynchronize(c->stream); }
  if (Rccl* R = rccl()) if (c->comm) R->CommDestroy(c->comm);
  if (c->stream) hipStreamDestroy(c->stream);
  if (c->flat) hipFree(c->flat);
  for (void* q : c->retired_dev) hipFree(q);
  if (c->h_inf) hipHostFree(c->h_inf);
  for (void* q : c->retired_host) hipHostFree(q);
  delete c;
}

}  // extern "C"
