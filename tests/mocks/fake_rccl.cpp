// fake_rccl.cpp — TEST INFRASTRUCTURE, not product code.  A stand-in for librccl.so that lets TWO rank processes share ONE GPU.
//
// RCCL refuses two ranks of a communicator on one device, and the build machines have one GPU, so the product's cross-rig merge
// (csrc/comm.cpp: jn_comm_create, jn_scan_allreduce, jn_elas_set_comm — submission-order turnstile, identity on failure, abort on
// time-out) would meet its first two-rank run on the driver's multi-GPU node.  comm.cpp binds librccl at run time and honours
// JN_RCCL_LIB; the tests point it HERE.  What this library keeps of the real thing is exactly what that logic depends on:
//   * ncclAllReduce is stream-ordered and asynchronous, every rank must issue a communicator's collectives in the same order, and a
//     rank whose peer never issues the matching collective WAITS inside it, for ever;
//   * ncclCommAbort ends such a wait on the calling rank only.
// The reduction itself goes through POSIX shared memory: device -> pinned host copy, a host function on the stream that exchanges the
// contributions with the peers and takes the element-wise minimum (ncclMin over ncclDouble is all the product uses), copy back.
// Built by tests/test_gpu_comm_two_ranks.py with hipcc; needs no RCCL at run time (only <rccl/rccl.h>'s declarations).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>

namespace {
constexpr int kMaxWorld = 8, kSlots = 8, kMaxCount = 8192, kStaging = 32;

struct Slot {
  std::atomic<unsigned long long> gen;   // how many times this slot has been used up (use k of the slot waits for gen == k)
  std::atomic<int> arrived, done;
  double data[kMaxWorld][kMaxCount];
};
struct Shared {
  std::atomic<int> joined;
  Slot slots[kSlots];
};
}  // namespace

struct ncclComm {
  Shared* sh = nullptr;
  int rank = 0, world = 1, device = 0;
  unsigned long long next = 0;           // collectives issued so far (host side, in issue order)
  std::atomic<bool> abort{false};
  double* staging[kStaging] = {};        // pinned, one per collective in flight (ring)
  char name[64] = {};
};

namespace {
struct Op { ncclComm* c; unsigned long long k; size_t count; double* host; };

bool wait_until(ncclComm* c, const std::function<bool()>& ready) {
  while (!ready()) {
    if (c->abort.load()) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
  return true;
}

void reduce_cb(void* p) {
  Op* op = static_cast<Op*>(p);
  ncclComm* c = op->c;
  Slot& s = c->sh->slots[op->k % kSlots];
  const unsigned long long use = op->k / kSlots;
  bool ok = wait_until(c, [&] { return s.gen.load() == use; });                 // the slot's previous use is finished on every rank
  if (ok) {
    memcpy(s.data[c->rank], op->host, op->count * sizeof(double));
    s.arrived.fetch_add(1);
    ok = wait_until(c, [&] { return s.arrived.load() >= c->world; });            // a peer that never issues this collective: wait here for ever (or until ncclCommAbort)
  }
  if (ok) {
    for (size_t i = 0; i < op->count; i++) {
      double m = s.data[0][i];
      for (int r = 1; r < c->world; r++) m = s.data[r][i] < m ? s.data[r][i] : m;
      op->host[i] = m;
    }
    if (s.done.fetch_add(1) + 1 == c->world) { s.arrived.store(0); s.done.store(0); s.gen.store(use + 1); }
  }
  delete op;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/jnfake_%d_%llx", (int)getpid(),
           (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
  const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); return ncclSystemError; }
  close(fd);                                                                   // zero-filled: every atomic starts at 0
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
  if (!out || world < 1 || world > kMaxWorld || rank < 0 || rank >= world) return ncclInvalidArgument;
  int fd = -1;
  for (int tries = 0; tries < 3000 && fd < 0; tries++) {                        // rank 0 may not have created it yet
    fd = shm_open(id.internal, O_RDWR, 0600);
    if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
  if (fd < 0) return ncclSystemError;
  void* m = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return ncclSystemError;
  ncclComm* c = new ncclComm();
  c->sh = static_cast<Shared*>(m); c->rank = rank; c->world = world;
  strncpy(c->name, id.internal, sizeof(c->name) - 1);
  if (hipGetDevice(&c->device) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
  for (auto& b : c->staging)
    if (hipHostMalloc(reinterpret_cast<void**>(&b), kMaxCount * sizeof(double), hipHostMallocDefault) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
  c->sh->joined.fetch_add(1);
  const auto t0 = std::chrono::steady_clock::now();
  while (c->sh->joined.load() < world) {                                        // a collective, like the real one
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { delete c; return ncclSystemError; }
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuf, void* recvbuf, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t stream) {
  if (!c || dt != ncclDouble || op != ncclMin || count > (size_t)kMaxCount) return ncclInvalidArgument;
  Op* o = new Op{c, c->next, count, c->staging[c->next % kStaging]};
  c->next++;
  if (hipMemcpyAsync(o->host, sendbuf, count * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipLaunchHostFunc(stream, reduce_cb, o) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpyAsync(recvbuf, o->host, count * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t c) {                                      // ends this rank's waits; the handle stays allocated (as far as the product goes: it drops it)
  if (!c) return ncclInvalidArgument;
  c->abort.store(true);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclInvalidArgument;
  c->abort.store(true);
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->sh, sizeof(Shared));
  // (staging buffers are left to the process' end: a host function of an aborted collective may still hold one)
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->world; return ncclSuccess; }
ncclResult_t ncclCommCuDevice(const ncclComm_t c, int* d) { *d = c->device; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl error"; }

}  // extern "C"
