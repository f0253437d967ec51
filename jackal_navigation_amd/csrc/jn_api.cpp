// jn_api.cpp — C-ABI of libjn_stereo.so (include/jn_stereo.h).  Product code.
//
// Pipeline per batch (one "slot" = one HIP stream + its buffers + one worker thread):
//   GPU stage A : Sobel planes -> support matching -> support filters -> support list -> alternating-cut arrangement   (kernels.hip)
//                 the list (uc, vc, d) is written by the GPU straight into pinned host memory
//   host stage  : Delaunay's hull recursion x2 per frame                       (delaunay.cpp, thread pool)
//                 (+ the support filters when no kernel takes the lattice or JN_HOST_FILTERS=1: host_stage.cpp)
//   H2D         : one copy per batch: support points + triangle corner indices
//   GPU stage B : grid prior, plane fits, raster bins, ownership -> dense L/R -> L/R check -> speckle -> gaps -> adaptive mean
//                 [-> u8 map + obstacle scan when submitted through jn_elas_submit_scan]
// Several slots in flight overlap one batch's host stage with another batch's GPU stages.
// Batch handles of processes with few cores of their own have NO host stage: the hull recursion runs on the GPU too (delaunay_gpu.hip),
// FrameInfo and the payload are written on the device and stage B is queued right behind it (run_batch_route).
#include "../../include/jn_stereo.h"
#include "hooks.h"
#include "kernels.h"
#include "host_stage.h"
#include "pool.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>
#include <sys/prctl.h>

using namespace jnav;

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess) {                                                                \
      fprintf(stderr, "libjn_stereo: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return JN_ERR_NO_DEVICE;                                                              \
    }                                                                                       \
  } while (0)

namespace {

struct Job {
  int n = 0; const uint8_t* dI1 = nullptr; const uint8_t* dI2 = nullptr; int pitch = 0; int64_t stride = 0;
  float* dD1 = nullptr; float* dD2 = nullptr; int32_t* status = nullptr;
  // host-pointer form (jn_elas_submit_host): the worker stages the images in and the maps out around the batch
  uint64_t seq = 0; bool merge = false;                       // scan batch whose bins are MIN-reduced across ranks before it completes
  bool staged = false;                                        // the images were written on the slot's ordinary stream (run_batch_host): stage A stays there
  bool host = false; const uint8_t* hI1 = nullptr; const uint8_t* hI2 = nullptr; float* hD1 = nullptr; float* hD2 = nullptr;
  // optional tail of the node on the same stream (jn_elas_submit_scan): u8 map + LUT scan of D1
  bool scan = false; jn_scan_params sp = {}; const uint8_t* dLut = nullptr; uint8_t* dDispU8 = nullptr; double* dBins = nullptr; double* dMeta = nullptr;
};

enum { EV_BEGIN, EV_DESC, EV_SUPPORT, EV_D2H, EV_H2D0, EV_H2D, EV_RASTER, EV_DENSE, EV_LR, EV_SPECKLE, EV_GAP, EV_AM, EV_END, EV_COUNT };

struct Slot {
  hipStream_t stream = nullptr;
  hipEvent_t ev_scan = nullptr, ev_merged = nullptr;          // around the cross-rig merge (created with the slot)
  hipEvent_t ev_head = nullptr;                                // behind the heavy head of stage A (descriptors + support matches): start-up pacing
  float merge_ms = 0.f;
  double* d_flat = nullptr;                                   // the merge's packed buffer of this slot [max_batch][1024 + 4] (written by k_scan_finish)
  hipStream_t stream_a = nullptr;                             // highest-priority stream for stage A (see run_batch); only with JN_STAGE_A_PRIORITY=1
  uint32_t* gate = nullptr; uint32_t gate_seq = 0;            // latency mode: the word stage B's queued launches wait on (hipMallocSignalMemory), see run_batch
  hipEvent_t ev[EV_COUNT] = {};
  // device
  uint4* desc = nullptr; uint8_t* planes = nullptr; int16_t* d_can = nullptr;   // descriptors: materialised (the old flow) OR the two Sobel planes (h->plane_flow)
  FrameInfo* info = nullptr; uint8_t* payload = nullptr; int32_t* bin_count = nullptr; BinEntry* bin_list = nullptr; int16_t* raw = nullptr;
  float* tmp = nullptr; int32_t* label = nullptr; int32_t* size = nullptr;
  uint32_t* mark = nullptr; uint32_t* gridbits = nullptr; TriRec* recs = nullptr;
  unsigned long long* scan_scratch = nullptr;                 // extrema of the scan tail, 4 per frame
  uint8_t* st_img = nullptr; float* st_D = nullptr;           // device staging of jn_elas_submit_host: [2][max_batch] images / maps, allocated on first use
  std::vector<FrameScratch> scratch;
  std::vector<HostWorker::SideState> sides;                  // [2 * max_batch]: per frame side, for the phased (parallel) triangulation
  // pinned host
  int16_t* h_can = nullptr; FrameInfo* h_info = nullptr; uint8_t* h_payload = nullptr;
  int16_t* h_list = nullptr; int32_t* h_cnt = nullptr;       // support lists the GPU writes straight into pinned memory
  uint16_t* h_arr = nullptr; int32_t* h_arr_ok = nullptr;    // alternating-cut arrangements per frame side (k_arrange), same route
  // the same four buffers in DEVICE memory, for handles that triangulate on the GPU: k_arrange and k_delaunay then read the list and the
  // arrangement from HBM instead of pulling ~26 KB per frame side over PCIe at the start of two latency-bound kernels
  int16_t* d_list = nullptr; int32_t* d_cnt = nullptr; uint16_t* d_arr = nullptr; int32_t* d_arr_ok = nullptr;
  uint8_t* dt_scratch = nullptr;                              // frames whose sides exceed one workgroup's LDS (1920x1080): the global structure of k_delaunay_sub / _top
  int arr_hint = 0;                                           // most support points a frame of this slot's last kArrHist batches had
  static constexpr int kArrHist = 4;
  int arr_hist[kArrHist] = {0, 0, 0, 0}; int arr_pos = 0;
  void* arr_scratch = nullptr;                                // device: working arrays of k_arrange for sides beyond its LDS capacity
  // worker
  std::thread th; std::mutex m; std::condition_variable cv;
  bool has_job = false, busy = false, quit = false;
  Job job; jn_status result = JN_OK;
  jn_stage_times times = {};
  float dense_ms = 0, owner_ms = 0; int dense_launches = 0;
  int last_n = 0;                                              // frames of the slot's last batch (jn_elas_bin_stats)
  int32_t* need_host = nullptr; int32_t* h_need = nullptr;     // per frame: sides k_delaunay handed back (device / pinned copy)
  long long gpu_dt_fallbacks = 0;                              // batches that went through the host stage after all
  hipEvent_t ev_owner = nullptr;                               // between k_owner and k_dense_row (plane flow, stage events on)
};

}  // namespace

struct jn_elas {
  jn_elas_params p;
  DevParams dp;
  HostParams hp;
  int W = 0, H = 0, max_batch = 0, device = 0;
  size_t payload_cap = 0;
  int tri_cap = 0;
  // Where the support filters run.  The wavefront kernel is a serial chain of ~6*cw steps on one workgroup per
  // frame, so its duration does not depend on the batch size; the host filters take one pool round per
  // `threads` frames.  The device wins once a batch needs more than one round (and it frees the pool for Delaunay);
  // for a lone pair or a batch the pool swallows at once the host is quicker.  JN_HOST_FILTERS at create time:
  // unset = device when the classify + resolve kernels apply (no serial sweep; lattice and codes fit the LDS) or the
  // batch exceeds the pool size, "1" = always host, "0" = always device.  The host also takes over when no kernel can
  // take the lattice.
  int filter_min_batch = 4;
  int wait_spin_us = 60;            // JN_WAIT_SPIN_US; 1000 for max_batch == 1 (see wait_event)
  bool stage_events = true;         // JN_STAGE_EVENTS: default on, off for max_batch == 1 (see run_batch)
  bool gpu_arrange = true;          // JN_GPU_ARRANGE=0: the host computes the alternating-cut arrangement itself (A/B, tests)
  int arr_cap = 0, arr_stride = 0;  // vertices per frame side k_arrange orders in LDS / at all (more: in global scratch / on the host)
  int dt_gcap = 0;                  // GPU triangulation: vertices per side beyond one workgroup's LDS that the global scratch lets through (0: none)
  bool split_delaunay = true;       // JN_SPLIT_DELAUNAY=0 keeps one task per frame side whatever the pool size (A/B, tests)
  bool filters_fast = false;        // the classify + resolve kernels apply (short, no serial sweep): device route for any batch size
  // cross-rig merge as the tail of a scan batch (jn_elas_set_comm): merges are queued in submission order on every rank
  jn_comm* comm = nullptr;
  std::mutex merge_m; std::condition_variable merge_cv;
  // Start-up pacing (JN_PACE, default on for batch handles).  After a synchronisation several batches are submitted at once and their
  // descriptor / support kernels share the GPU: all of them reach their host stage late, and the GPU then idles while the pool works
  // through four host stages.  A batch's stage A therefore waits (on the device) until the batch submitted before it has finished its two
  // heavy kernels — the phase the pipeline settles into by itself.  In steady state that event is long complete: the wait is a no-op.
  std::mutex pace_m; hipEvent_t pace_prev = nullptr; bool pace = false;
  bool sub = false;                 // param.subsampling: half-size maps (elas.h:82, :160-162); dph = the post-processing's parameters at that size
  DevParams dph = {};
  bool zero_copy_payload = false;   // latency mode: stage B reads the host stage's output in pinned memory instead of a copy of it
  bool arrange_sorts = false;       // hooks build, JN_ARRANGE_SORTS=1: k_arrange's sort forms where its rank form would run (A/B, tests)
  bool gpu_delaunay = false;        // batch handles: the triangulations' hull recursion on the GPU too (delaunay_gpu.hip; JN_GPU_DELAUNAY=0/1), no host stage
  bool plane_flow = true;           // descriptors assembled from the Sobel planes inside the matching kernels (JN_DESC_FLOW=desc: materialised, the old flow)
  std::atomic<bool> gate_stage_b{false};   // latency mode: stage B is queued behind a gate while the GPU runs stage A (JN_GATE_STAGE_B=0/1), see run_batch
  uint64_t submit_seq = 0, merge_seq = 0;                     // next number handed to a scan batch / next batch allowed to queue its merge
  std::vector<uint64_t> merge_log;                            // submission numbers in the order their merges were queued (the last 4096; jn_elas_merge_order)
  int comm_timeout_ms = 30000;                                // JN_COMM_TIMEOUT_MS: a merge not complete by then is aborted (0: wait for ever)
  long long test_fail_seq = -1;                               // JN_TEST_FAIL_SEQ=k: the scan batch with submission number k fails before its kernels (tests: a rank's batch dies, the merge order must survive)
  std::vector<int> test_slot_delay_us;                        // JN_TEST_SLOT_DELAY_US="a,b,c,d": slot i's batches pause that long before their merge turn (tests: host stages of unequal length)
  std::unique_ptr<Pool> pool;
  std::vector<std::unique_ptr<Slot>> slots;
  // staging for the host-pointer drop-in call
  uint8_t* s_img = nullptr; float* s_D = nullptr; int s_pitch = 0;
  std::mutex api_m;
};

namespace jnav {
jn_status comm_merge_async(jn_comm* c, int n, int bins, double* dBins, double* dMeta, hipEvent_t ready, hipEvent_t done, double* packed);
jn_status comm_merge_identity(jn_comm* c, int n, int bins);
void comm_abort(jn_comm* c);
bool comm_dead(const jn_comm* c);
int comm_device(const jn_comm* c);
}

namespace {

// Runs when the library is loaded.  The HIP runtime multiplexes streams onto 4 hardware queues by default; with one
// stream per slot plus the caller's, two slots then share a queue and serialise.  Ask for 16 unless the user chose a
// value (8 until round 4: enough for one four-slot handle, but a process that also runs the SGM or block-matching mode
// with its own slots lost 10 % there, profiles/r04_hw_queues_ab.txt; 16 measured neutral for a single handle); it only
// takes effect if the HIP runtime has not initialised yet (load this library first, or export it).
__attribute__((constructor)) void prefer_one_queue_per_slot() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

// Waiting for the GPU without burning the host's cores.  hipEventSynchronize spins on this runtime even for events created
// with hipEventBlockingSync: the four slot workers then cost 2.5 cores of pure waiting (measured: 4.2 ms of CPU per 32-pair
// batch), and on the GPU boxes the container's CPU quota (16 CPUs) is what the Delaunay pool needs.  So: poll the event —
// tightly for the first 60 us, then between short sleeps (a batch's stage lasts milliseconds; the other slots keep the GPU
// busy meanwhile).  A latency-mode handle (max_batch 1) polls tightly for 1 ms: its stages are short and a sleep's wake-up
// would show in every call.  JN_WAIT_SPIN_US overrides (-1: plain hipEventSynchronize).
hipError_t wait_event(hipEvent_t ev, int spin_us) {
  if (spin_us < 0) return hipEventSynchronize(ev);
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    const auto waited = std::chrono::steady_clock::now() - t0;
    if (waited < std::chrono::microseconds(spin_us)) { __builtin_ia32_pause(); continue; }
    std::this_thread::sleep_for(std::chrono::microseconds(waited < std::chrono::microseconds(500) ? 20 : 50));
  }
}

// The same with a deadline: hipErrorNotReady when `timeout_ms` (> 0) passed without the event completing.
hipError_t wait_event_bounded(hipEvent_t ev, int spin_us, int timeout_ms) {
  if (timeout_ms <= 0) return wait_event(ev, spin_us);
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    const auto waited = std::chrono::steady_clock::now() - t0;
    if (waited > std::chrono::milliseconds(timeout_ms)) return hipErrorNotReady;
    if (waited < std::chrono::microseconds(std::max(spin_us, 0))) { __builtin_ia32_pause(); continue; }
    std::this_thread::sleep_for(std::chrono::microseconds(waited < std::chrono::microseconds(500) ? 20 : 50));
  }
}

// CPUs this process may really use: its affinity mask, cut down to the container's CPU quota (cgroup v2 cpu.max) — a pool
// sized by the machine's core count inside a container with a smaller quota gets the whole container throttled.
int usable_cpus() {
  int n = (int)std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    long long quota = 0, period = 0;
    if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) n = std::min<long long>(n, std::max<long long>(1, quota / period));
    fclose(f);
  }
  return n;
}

// Parts a triangulation is cut into on the host: idle pool threads (a lone pair, a few large frames) are put to work inside it.
int delaunay_parts(const jn_elas* h, int n) {
  const int threads = h->pool->size();
  return h->split_delaunay ? (threads >= 8 * n ? 4 : (threads >= 4 * n ? 2 : 1)) : 1;
}

// A scan batch that carries a merge owns one place in the handle's merge order.  If the batch ends early (a HIP error on the
// way), the place must still be given up, or every later batch of this handle would wait for it for ever — and the OTHER ranks
// of the communicator are inside, or about to enter, the same all-reduce: this rank still takes part in it, contributing the
// identity of MIN (comm_merge_identity), so the peers get the remaining rigs' scan while this rank reports its error.
struct MergeTurn {
  jn_elas* h; uint64_t seq; bool armed; int n, bins;
  MergeTurn(jn_elas* h_, const Job& j) : h(h_), seq(j.seq), armed(j.merge), n(j.n), bins(j.sp.bins) {}
  void done() { armed = false; }
  ~MergeTurn() {
    if (!armed) return;
    {
      std::unique_lock<std::mutex> l(h->merge_m);
      h->merge_cv.wait(l, [&] { return h->merge_seq == seq; });
      if (h->comm) comm_merge_identity(h->comm, n, bins);
      if (h->merge_log.size() >= 4096) h->merge_log.erase(h->merge_log.begin(), h->merge_log.begin() + 2048);
      h->merge_log.push_back(seq);
      h->merge_seq++;
    }
    h->merge_cv.notify_all();
  }
};

jn_status run_batch_route(jn_elas* h, Slot& s, const Job& j, MergeTurn& turn, bool force_host);
jn_status run_batch(jn_elas* h, Slot& s, const Job& j) {
  MergeTurn turn(h, j);
  if (j.merge && h->test_fail_seq >= 0 && (long long)j.seq == h->test_fail_seq) return JN_ERR_INTERNAL;
  return run_batch_route(h, s, j, turn, false);
}
// force_host: the triangulations on the host (the route of latency-mode handles, of parameter sets with corner points, and the second pass
// of a batch whose frames the GPU's triangulation handed back)
jn_status run_batch_route(jn_elas* h, Slot& s, const Job& j, MergeTurn& turn, bool force_host) {
  const DevParams& dp = h->dp;
  const int n = j.n;
  hipStream_t st = s.stream;
  HIP_TRY(hipSetDevice(h->device));
  auto t_begin = std::chrono::steady_clock::now();
  // Stage boundaries for jn_elas_last_times.  A timing event between two kernels costs ~6 us of idle GPU: nothing when
  // other slots fill the gap, 7 % of a lone 640x480 pair — a latency-mode handle (max_batch 1) leaves them out.
  const bool stage_events = h->stage_events;
  auto mark = [&](int e) { return stage_events ? hipEventRecord(s.ev[e], st) : hipSuccess; };
  // Stage A (descriptors -> support matches -> filters -> list -> arrangement) ends in the host stage, which the whole batch
  // waits for; its small kernels (one workgroup per frame or side) would otherwise queue behind the dense kernels of the
  // other slots.  It runs on a stream of the highest priority; stage B stays on the slot's ordinary stream.  The two never
  // overlap within a slot (the worker waits for stage A, and for the batch's end before the next stage A), so no events tie
  // them together.  Host-pointer jobs stage their images on the ordinary stream and keep everything there.
  hipStream_t sa = (s.stream_a && !j.staged) ? s.stream_a : st;
  const DescSrc dsrc = h->plane_flow ? DescSrc{s.planes, plane_pitch(dp.W), true} : DescSrc{s.desc, 0, false};
  auto mark_a = [&](int e) { return stage_events ? hipEventRecord(s.ev[e], sa) : hipSuccess; };
  {
    std::unique_lock<std::mutex> pl(h->pace_m, std::defer_lock);
    if (h->pace) {
      pl.lock();
      if (h->pace_prev && h->pace_prev != s.ev_head) HIP_TRY(hipStreamWaitEvent(sa, h->pace_prev, 0));
    }
    HIP_TRY(mark_a(EV_BEGIN));
    if (h->plane_flow) launch_sobel_planes(sa, dp, j.dI1, j.dI2, j.pitch, j.stride, n, s.planes);
    else launch_descriptor(sa, dp, j.dI1, j.dI2, j.pitch, j.stride, n, s.desc);
    HIP_TRY(mark_a(EV_DESC));
    launch_support(sa, dp, n, dsrc, s.d_can);
    if (h->pace) { HIP_TRY(hipEventRecord(s.ev_head, sa)); h->pace_prev = s.ev_head; }
  }
  const int list_cap = dp.cw * dp.ch;
  bool listed = false;                                   // k_filter_resolve wrote the support list too
  // Where the list and the arrangement live: in device memory when this batch is going to triangulate on the GPU (everything that decides
  // it is known here except whether the filter kernel lists the points itself: if it does not, the list goes to pinned memory and the host
  // route is taken), in pinned host memory for the host stage.
  const bool want_gpu_dt = h->gpu_delaunay && !force_host && sa == st && s.d_list &&
                           s.arr_hint <= (s.dt_scratch ? h->dt_gcap : delaunay_gpu_capacity(152 * 1024)) && h->gpu_arrange && s.arr_hint <= h->arr_stride;
  int16_t* const list_buf = want_gpu_dt ? s.d_list : s.h_list; int32_t* const cnt_buf = want_gpu_dt ? s.d_cnt : s.h_cnt;
  uint16_t* const arr_buf = want_gpu_dt ? s.d_arr : s.h_arr; int32_t* const arr_ok_buf = want_gpu_dt ? s.d_arr_ok : s.h_arr_ok;
  const bool filtered = (n >= h->filter_min_batch || (h->filter_min_batch < (1 << 30) && h->filters_fast)) &&
      launch_support_filters(sa, dp, n, h->p.incon_window_size, h->p.incon_threshold, h->p.incon_min_support, s.d_can, s.tmp, list_buf, cnt_buf, list_cap, &listed);
  HIP_TRY(mark_a(EV_SUPPORT));
  bool arranged = false, gpu_dt = false;
  if (filtered) {                                        // the GPU lists the support points itself (into pinned host memory for the host stage)
    if (!listed) { launch_support_list(sa, dp, n, s.d_can, list_buf, cnt_buf, list_cap); listed = true; }
    // the arrangement the triangulations start from, unless the pool has idle threads and will cut them into parts itself
    // Sized by what this slot's previous batch held (+25 %): a 720p frame has 3.2 k support points and needs 52 KB of LDS, not
    // the 104 KB of the 8192-vertex maximum — a workgroup that asks for less finds room among the other slots' kernels sooner.
    // Frames beyond the maximum (1920x1080: 11 k points) skip the launch: it could only hand every side back.
    // The triangulation itself on the GPU (delaunay_gpu.hip) wherever it applies: batch handles, no corner points, lattices the LDS holds.
    // Then there is NO host stage: k_delaunay writes FrameInfo and the payload on the device, stage B is queued right behind it with
    // capacity-sized launches, and the worker only waits for the batch's end.
    gpu_dt = want_gpu_dt && listed;
    arranged = h->gpu_arrange && (gpu_dt || delaunay_parts(h, n) == 1) && s.arr_hint <= h->arr_stride;
    if (arranged) {
      const int want = s.arr_hint ? s.arr_hint + s.arr_hint / 4 + 64 : h->arr_cap;
      // more points than the LDS can order (1920x1080: 11 k): every side works in its slice of the global scratch, the launch asks for the minimum of LDS
      const int cap = s.arr_hint > h->arr_cap ? 1024 : std::min(h->arr_cap, std::max(1024, (want + 1023) / 1024 * 1024));
      {
        // (the global-scratch form only when the slot's recent batches held a side beyond the LDS form: at 1280x720 it would be an empty launch per batch)
        const bool big = s.arr_scratch && (s.arr_hint == 0 || s.arr_hint > h->arr_cap);   // (0: the slot's first batch — nothing known yet)
        launch_arrange(sa, n, list_buf, cnt_buf, list_cap, dp.step, cap, h->arr_stride, arr_buf, arr_ok_buf, big ? s.arr_scratch : nullptr, big ? h->arr_stride : 0,
                       h->arrange_sorts ? ArrBounds{0, 0, 0, 0} : ArrBounds{dp.ch, dp.cw, -dp.disp_max, (dp.cw - 1) * dp.step + dp.disp_max + 1});
      }
      if (gpu_dt)                                        // LDS for what the slot's last batches held + 6 % (a tight request: 32 bytes a vertex leave a k_dense_row workgroup room on the same CU); a side beyond it goes to the host
        HIP_TRY(launch_delaunay(sa, n, list_buf, cnt_buf, list_cap, dp.step, arr_buf, arr_ok_buf, h->arr_stride, s.arr_hint ? std::max(1024, s.arr_hint + s.arr_hint / 16 + 32) : (1 << 30), s.payload,
                                (long long)h->payload_cap, s.info, s.need_host, nullptr, s.dt_scratch, h->dt_gcap, s.arr_hint, dp.W >= 2048 || dp.H >= 2048));
    }
  } else {
    const size_t can_bytes = (size_t)dp.cw * dp.ch * sizeof(int16_t);
    HIP_TRY(hipMemcpyAsync(s.h_can, s.d_can, can_bytes * n, hipMemcpyDeviceToHost, sa));
  }
  HIP_TRY(hipEventRecord(s.ev[EV_D2H], sa));

  // ---- stage B, as a function of what the host stage yields: the largest support / triangle counts (launch sizes), whether any frame
  // has a triangulation, where the payload is read from, and whether the two clears were queued ahead ----
  const bool fused = gap_mean_fusable(dp, n) && ((dp.W * dp.H) & 3) == 0;
  // The candidate grid (elas.cpp:582-680) needs the support points, not the triangulation: without corner points they are the list the
  // GPU has just written, so the grid is queued HERE, behind stage A, and is built while the host triangulates (JN_GRID_EARLY=0: in stage B).
  static const bool grid_early_env = !(getenv("JN_GRID_EARLY") && atoi(getenv("JN_GRID_EARLY")) == 0);
  const bool grid_early = grid_early_env && filtered && !dp.add_corners && sa == st;
  if (grid_early) launch_grid_from_list(st, dp, n, list_buf, cnt_buf, list_cap, s.mark, s.gridbits);
  auto queue_stage_b = [&](int max_sup, int max_tri, bool any_ok, const uint8_t* payload, size_t payload_bytes, bool cleared, bool device_info = false) -> jn_status {
    HIP_TRY(mark(EV_H2D0));
    if (!device_info) HIP_TRY(hipMemcpyAsync(s.info, s.h_info, sizeof(FrameInfo) * n, hipMemcpyHostToDevice, st));
    if (payload_bytes && payload == s.payload) HIP_TRY(hipMemcpyAsync(s.payload, s.h_payload, payload_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(mark(EV_H2D));
    if (any_ok) {
      if (!grid_early) launch_grid(st, dp, n, s.info, payload, 0, max_sup, s.mark, s.gridbits, !cleared);      // offsets in FrameInfo are batch-absolute
      launch_bin(st, dp, n, s.info, s.recs, h->tri_cap, max_tri, s.bin_count, s.bin_list, !cleared, payload, 0);   // (forms the triangles' records on the way: k_tri_setup's work)
      HIP_TRY(mark(EV_RASTER));
      launch_dense(st, dp, n, s.info, s.recs, h->tri_cap, s.bin_count, s.bin_list, s.gridbits, dsrc, s.raw, false, (stage_events && h->plane_flow) ? s.ev_owner : nullptr);
      HIP_TRY(mark(EV_DENSE));
      // Post-processing.  When gap interpolation and adaptive mean can run as one pass (gap_mean_fusable), the left image
      // travels raw -> tmp (L/R check) -> tmp (speckle, run lists in the still idle output image) -> D1 (fused pass), so that
      // every stage reads and writes the image once; otherwise the stages run in place on D1 with tmp as scratch.
      if (h->sub) {
        // subsampling: the matcher ran on every pixel (findMatch is per pixel, so the reference's half-size map is the full one at even
        // (u, v)); the L/R check picks those out, everything behind it works on (W/2) x (H/2) maps with dph
        const DevParams& dph = h->dph;
        launch_lr_sub(st, dp, n, s.info, s.raw, j.dD1, j.dD2);
        HIP_TRY(mark(EV_LR));
        launch_speckle(st, dph, n, s.info, j.dD1, s.label, s.size, s.tmp);
        if (!h->p.postprocess_only_left) launch_speckle(st, dph, n, s.info, j.dD2, s.label, s.size, s.tmp);
        HIP_TRY(mark(EV_SPECKLE));
        launch_gap(st, dph, n, s.info, j.dD1, s.tmp);
        if (!h->p.postprocess_only_left) launch_gap(st, dph, n, s.info, j.dD2, s.tmp);
        HIP_TRY(mark(EV_GAP));
        if (h->p.filter_adaptive_mean) {
          launch_adaptive_mean_sub(st, dph, n, s.info, j.dD1, s.tmp);
          if (!h->p.postprocess_only_left) launch_adaptive_mean_sub(st, dph, n, s.info, j.dD2, s.tmp);
        }
        if (h->p.filter_median) {
          launch_median(st, dph, n, s.info, j.dD1, s.tmp);
          if (!h->p.postprocess_only_left) launch_median(st, dph, n, s.info, j.dD2, s.tmp);
        }
        HIP_TRY(mark(EV_AM));
      } else {
      if (fused) {
        launch_lr_speckle(st, dp, n, s.info, s.raw, s.tmp, j.dD2, s.label, s.size, j.dD1);   // (the L/R check and the speckle pass' row labelling are one kernel here)
        HIP_TRY(mark(EV_LR));
        HIP_TRY(mark(EV_SPECKLE));
        launch_gap_mean_fused(st, dp, n, s.info, s.tmp, j.dD1, h->p.filter_adaptive_mean != 0);
        if (!h->p.postprocess_only_left) {                     // right image: in place, fused pass into tmp, copied back
          launch_speckle(st, dp, n, s.info, j.dD2, s.label, s.size, s.tmp);
          launch_gap_mean_fused(st, dp, n, s.info, j.dD2, s.tmp, h->p.filter_adaptive_mean != 0);
          launch_copy_ok(st, dp, n, s.info, s.tmp, j.dD2);
        }
        HIP_TRY(mark(EV_GAP));
      } else {
        launch_lr_speckle(st, dp, n, s.info, s.raw, j.dD1, j.dD2, s.label, s.size, s.tmp);   // L/R check of both maps + the left map's speckle pass (its row labelling in the L/R kernel)
        HIP_TRY(mark(EV_LR));
        if (!h->p.postprocess_only_left) launch_speckle(st, dp, n, s.info, j.dD2, s.label, s.size, s.tmp);
        HIP_TRY(mark(EV_SPECKLE));
        launch_gap(st, dp, n, s.info, j.dD1, s.tmp);
        if (!h->p.postprocess_only_left) launch_gap(st, dp, n, s.info, j.dD2, s.tmp);
        HIP_TRY(mark(EV_GAP));
        if (h->p.filter_adaptive_mean) {
          launch_adaptive_mean(st, dp, n, s.info, j.dD1, s.tmp);
          if (!h->p.postprocess_only_left) launch_adaptive_mean(st, dp, n, s.info, j.dD2, s.tmp);
        }
      }
      if (h->p.filter_median) {                                                            // elas.cpp:133-139
        launch_median(st, dp, n, s.info, j.dD1, s.tmp);
        if (!h->p.postprocess_only_left) launch_median(st, dp, n, s.info, j.dD2, s.tmp);
      }
      HIP_TRY(mark(EV_AM));
      }
    } else {
      for (int e = EV_RASTER; e <= EV_AM; e++) HIP_TRY(mark(e));
    }
    if (j.scan)                                            // the node's tail: depth map + obstacle scan of whatever D1 now holds
      launch_scan(st, j.sp, n, j.dD1, j.dDispU8, j.dLut, dp.W, dp.H, j.dBins, j.dMeta, s.scan_scratch, j.merge ? s.d_flat : nullptr);
    HIP_TRY(hipEventRecord(s.ev[EV_END], st));
    return JN_OK;
  };
  auto t_host0 = std::chrono::steady_clock::now(), t_host1 = t_host0;
  int any_ok = 0;
  if (gpu_dt) {
    any_ok = 1;
    const jn_status qs = queue_stage_b(list_cap, h->tri_cap, true, s.payload, 0, false, true);
    if (qs != JN_OK) return qs;
    // what the host needs of the batch: which frames matched out (status), how many support points they held (the next launches' LDS),
    // whether a side was handed back — copied behind everything else, read after the one wait
    HIP_TRY(hipMemcpyAsync(s.h_info, s.info, sizeof(FrameInfo) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(s.h_need, s.need_host, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(s.ev[EV_END], st));
    HIP_TRY(wait_event(s.ev[EV_END], h->wait_spin_us));
    HIP_TRY(hipGetLastError());
    int batch_most = 0, handed_back = 0;
    for (int i = 0; i < n; i++) { batch_most = std::max(batch_most, (int)s.h_info[i].reserved); handed_back |= s.h_need[i]; }   // (k_delaunay leaves the frame's support count, clipped or not, in `reserved`)
    s.arr_hist[s.arr_pos] = batch_most; s.arr_pos = (s.arr_pos + 1) % Slot::kArrHist;
    s.arr_hint = *std::max_element(s.arr_hist, s.arr_hist + Slot::kArrHist);
    if (handed_back) {                                   // coinciding vertices or more of them than the launch's LDS held: the whole batch again, host stage and all
      s.gpu_dt_fallbacks++;
      return run_batch_route(h, s, j, turn, true);
    }
    for (int i = 0; i < n; i++) if (j.status) j.status[i] = s.h_info[i].ok ? JN_OK : JN_ERR_FEW_SUPPORT;
  } else {
  // Latency mode (a handle of max_batch 1): a lone pair's stage B is two dozen launches of a few microseconds each, and queued after the
  // host stage they reach the GPU slower than it finishes them (~30 us of idle gaps at 640x480).  They are queued NOW instead, while the
  // GPU runs stage A, behind a wait on a word of signal memory that the host sets when its stage is done (hipStreamWaitValue32).  What
  // the host stage decides is then not known at launch time: the three launches sized by support / triangle counts take their capacity
  // (the kernels return on indices beyond the frame's counts), the payload and FrameInfo are read where the host will have written them,
  // and the two clears that depend on nothing run ahead of the gate.  Whatever happens afterwards, the gate is opened (GateGuard): a
  // stream left waiting would hang the handle.
  struct GateGuard {
    volatile uint32_t* word = nullptr; uint32_t value = 0;
    FrameInfo* info = nullptr; int n = 0; hipStream_t st = nullptr;
    void open() { if (word) { std::atomic_thread_fence(std::memory_order_seq_cst); *word = value; word = nullptr; } }
    // An early return with the gate still shut: stage B is on the stream and WILL run once the gate opens, on whatever FrameInfo holds —
    // the previous batch's, if the host stage never ran.  Every frame is therefore marked as failed first (the matching and the
    // post-processing return on !ok; the scan tail still scans whatever D1 holds into the caller's buffers), and the stream is drained
    // before the error goes back: the caller may free its buffers as soon as it has it.
    ~GateGuard() {
      if (!word) return;
      for (int i = 0; i < n; i++) info[i].ok = 0;
      open();
      hipStreamSynchronize(st);
    }
  } gate;
  bool gated = h->gate_stage_b && s.gate && filtered && h->zero_copy_payload && sa == st;
  bool cleared = false;                                  // the two clears are on the stream already
  if (gated) {
    if (!grid_early) launch_grid_clear(st, dp, n, s.mark);
    launch_bin_clear(st, dp, n, s.bin_count);
    cleared = true;
    const uint32_t v = ++s.gate_seq;
    if (hipStreamWaitValue32(st, s.gate, v, hipStreamWaitValueEq, 0xFFFFFFFFu) != hipSuccess) {
      (void)hipGetLastError();                           // a runtime that reports the capability but refuses the call: this handle goes on without the gate
      h->gate_stage_b = false; gated = false;
    } else {
      gate.word = s.gate; gate.value = v; gate.info = s.h_info; gate.n = n; gate.st = st;
      const jn_status qs = queue_stage_b(list_cap + HostWorker::kCornerPoints, h->tri_cap, true, s.h_payload, 0, true);
      if (qs != JN_OK) return qs;
    }
  }
  HIP_TRY(wait_event(s.ev[EV_D2H], h->wait_spin_us));

  t_host0 = std::chrono::steady_clock::now();
  size_t payload_bytes = 0;                              // frames packed back to back: one H2D copy per batch
  if (filtered) {
    // the counts are known, so the frames can be placed at once and the batch is one flat set of frame-side tasks
    int batch_most = 0;
    for (int i = 0; i < n; i++) batch_most = std::max(batch_most, (int)s.h_cnt[i]);
    s.arr_hist[s.arr_pos] = batch_most; s.arr_pos = (s.arr_pos + 1) % Slot::kArrHist;   // a lone sparse frame no longer shrinks
    s.arr_hint = *std::max_element(s.arr_hist, s.arr_hist + Slot::kArrHist);            // the next batch's arrangement space
    for (int i = 0; i < n; i++) {
      FrameInfo& fi = s.h_info[i];
      memset(&fi, 0, sizeof(fi));
      fi.nsup = std::min(s.h_cnt[i], list_cap) + (h->hp.add_corners ? HostWorker::kCornerPoints : 0);   // elas.cpp:435
      fi.ok = fi.nsup >= 3;                              // elas.cpp:66-71
      payload_bytes += HostWorker::place(&fi, payload_bytes);
    }
    // Idle pool threads (a lone pair, a few large frames) are put to work inside the triangulations: every frame side
    // is cut into 2 or 4 independent parts (delaunay.h), three short pool rounds instead of one long one.
    const int want_parts = delaunay_parts(h, n);
    if (want_parts == 1) {
      h->pool->run(2 * n, [&](HostWorker& w, int k) {
        const int i = k >> 1;
        const uint16_t* arr = (arranged && s.h_arr_ok[k]) ? s.h_arr + (size_t)k * h->arr_stride : nullptr;
        w.triangulate_side_from_list(k & 1, s.h_list + (size_t)i * list_cap * 3, s.h_payload, &s.h_info[i], arr);
      });
    } else {
      h->pool->run(2 * n, [&](HostWorker& w, int k) {
        const int i = k >> 1;
        w.side_prepare(k & 1, s.h_list + (size_t)i * list_cap * 3, s.h_payload, &s.h_info[i], &s.sides[k], want_parts);
      });
      h->pool->run(2 * n * want_parts, [&](HostWorker&, int k) {
        HostWorker::SideState& st = s.sides[k / want_parts];
        if (k % want_parts < st.parts) st.dt.subtree(k % want_parts);
      });
      h->pool->run(2 * n, [&](HostWorker&, int k) { HostWorker::side_finish(k & 1, s.h_payload, &s.h_info[k >> 1], &s.sides[k]); });
    }
  } else {
    h->pool->run(n, [&](HostWorker& w, int i) {          // phase 1: filters + support list, per frame
      w.filter_and_list(s.h_can + (size_t)i * dp.cw * dp.ch, &s.h_info[i], &s.scratch[i], false);
    });
    for (int i = 0; i < n; i++) payload_bytes += HostWorker::place(&s.h_info[i], payload_bytes);
    h->pool->run(2 * n, [&](HostWorker& w, int k) {      // phase 2: one triangulation per frame and side
      const int i = k >> 1;
      w.triangulate_side(k & 1, s.scratch[i], s.h_payload, &s.h_info[i]);
    });
  }
  t_host1 = std::chrono::steady_clock::now();

  int max_tri = 0, max_sup = 0;
  for (int i = 0; i < n; i++) {
    const FrameInfo& fi = s.h_info[i];
    if (j.status) j.status[i] = fi.ok ? JN_OK : JN_ERR_FEW_SUPPORT;
    if (!fi.ok) continue;
    any_ok = 1;
    max_tri = std::max(max_tri, std::max(fi.ntri[0], fi.ntri[1]));
    max_sup = std::max(max_sup, fi.nsup);
  }
  if (gated) gate.open();
  else {
    // A latency-mode handle lets the two kernels that consume the payload read it where the host wrote it (pinned memory is visible to
    // the device): a lone pair's payload is ~50 KB read once, and the copy plus the pause behind it cost more than that (JN_ZERO_COPY=0/1).
    const jn_status qs = queue_stage_b(max_sup, max_tri, any_ok != 0, h->zero_copy_payload ? s.h_payload : s.payload, payload_bytes, cleared);
    if (qs != JN_OK) return qs;
  }
  HIP_TRY(wait_event(s.ev[EV_END], h->wait_spin_us));
  HIP_TRY(hipGetLastError());
  }   // (host-stage route)
  bool merged = false;
  float merge_host_ms = 0.f;
  if (j.merge) {
    // The path's one exchange step (point_cloud.cpp:264-266 across rigs): the bins of this batch MIN-reduced over the ranks,
    // as the batch's tail, issued by THIS worker (the submitting thread is not involved, the other slots keep the GPU busy).
    // RCCL wants every rank to issue a communicator's collectives in one order: batches take their turn in submission order
    // (every rank submits the same sequence), whatever order their host stages finished in.
    // The scan is complete here (the wait above), so pack -> all-reduce -> unpack need no cross-stream dependency: chaining
    // them to the slot's stream with events cost 0.66 ms per batch on a busy GPU (two queue hand-overs), this costs the
    // kernels themselves plus one host wait (profiles/r03_merge_in_worker.txt).
    const auto t_m0 = std::chrono::steady_clock::now();
    jn_status ms_ = JN_OK;
    if (!h->test_slot_delay_us.empty()) {                  // tests only: this slot's host side takes longer, so batches reach their merge out of submission order
      size_t si = 0;
      while (si < h->slots.size() && h->slots[si].get() != &s) si++;
      const int us = h->test_slot_delay_us[si % h->test_slot_delay_us.size()];
      if (us > 0) std::this_thread::sleep_for(std::chrono::microseconds(us));
    }
    {
      std::unique_lock<std::mutex> l(h->merge_m);
      h->merge_cv.wait(l, [&] { return h->merge_seq == j.seq; });
      ms_ = comm_merge_async(h->comm, n, j.sp.bins, j.dBins, j.dMeta, nullptr, s.ev_merged, s.d_flat);   // packed by k_scan_finish: all-reduce in place + unpack
      if (h->merge_log.size() >= 4096) h->merge_log.erase(h->merge_log.begin(), h->merge_log.begin() + 2048);
      h->merge_log.push_back(j.seq);
      h->merge_seq++;                                      // even on failure: the batches behind must not wait for ever
    }
    turn.done();
    h->merge_cv.notify_all();
    if (ms_ != JN_OK) return ms_;
    {
      // a short wait (two small kernels): poll tightly, a sleep's granularity would show.  Bounded: a peer that died or never issued its
      // collective must not hang this rank — the communicator is aborted and this and all later scan batches return JN_ERR_COMM.
      const hipError_t we = wait_event_bounded(s.ev_merged, std::max(h->wait_spin_us, 400), h->comm_timeout_ms);
      if (we == hipErrorNotReady) { comm_abort(h->comm); return JN_ERR_COMM; }
      HIP_TRY(we);
      // another slot's merge timed out and aborted the communicator meanwhile: this merge's event completed because the aborted kernels
      // exited, its bins were never reduced
      if (comm_dead(h->comm)) return JN_ERR_COMM;
    }
    merge_host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_m0).count();
    merged = true;
  }
  auto t_end = std::chrono::steady_clock::now();

  auto ms = [&](int a, int b) { float v = 0; if (stage_events) hipEventElapsedTime(&v, s.ev[a], s.ev[b]); return v; };
  jn_stage_times& t = s.times;
  t.gpu_descriptor = ms(EV_BEGIN, EV_DESC); t.gpu_support = ms(EV_DESC, EV_SUPPORT); t.d2h = ms(EV_SUPPORT, EV_D2H);
  t.host_stage = std::chrono::duration<float, std::milli>(t_host1 - t_host0).count();
  t.h2d = ms(EV_H2D0, EV_H2D);
  t.gpu_matching = ms(EV_H2D, EV_DENSE); t.gpu_lr = ms(EV_DENSE, EV_LR); t.gpu_speckle = ms(EV_LR, EV_SPECKLE);
  t.gpu_gap = ms(EV_SPECKLE, EV_GAP); t.gpu_adaptive_mean = ms(EV_GAP, EV_AM);
  t.total = std::chrono::duration<float, std::milli>(t_end - t_begin).count();
  s.last_n = n;
  s.dense_launches = any_ok && stage_events ? 1 : 0;
  if (s.dense_launches && h->plane_flow) {                   // k_bin | k_owner | k_dense_row: the matcher proper is timed from behind k_owner
    float a = 0, b = 0;
    hipEventElapsedTime(&a, s.ev[EV_RASTER], s.ev_owner); hipEventElapsedTime(&b, s.ev_owner, s.ev[EV_DENSE]);
    s.owner_ms = a; s.dense_ms = b;
  } else { s.dense_ms = ms(EV_RASTER, EV_DENSE); s.owner_ms = 0; }
  s.merge_ms = merged ? merge_host_ms : 0.f;               // scan complete -> merged bins in place, on the worker's clock (its turn in the order included)
  return JN_OK;
}

// Host pointers: images in (one copy per image when the caller's rows are padded or the images are apart, else one per
// side), the batch, the maps of the pairs that matched out (elas.cpp:66-71: a pair with too few support points leaves the
// caller's D1 / D2 untouched).  Runs on the slot's worker thread, so the copies of one slot overlap the kernels of the others.
jn_status run_batch_host(jn_elas* h, Slot& s, const Job& j) {
  HIP_TRY(hipSetDevice(h->device));
  const size_t px = (size_t)h->W * h->H, B = (size_t)h->max_batch;
  if (!s.st_img || !s.st_D) {                            // both or neither: a failed second allocation must not leave a half-made pair
    if (s.st_img) { hipFree(s.st_img); s.st_img = nullptr; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.st_img), 2 * B * px));
    if (hipMalloc(reinterpret_cast<void**>(&s.st_D), 2 * B * px * sizeof(float)) != hipSuccess) {
      hipFree(s.st_img); s.st_img = nullptr; s.st_D = nullptr;
      return JN_ERR_NO_DEVICE;
    }
  }
  hipStream_t st = s.stream;
  const uint8_t* src[2] = {j.hI1, j.hI2};
  for (int side = 0; side < 2; side++) {
    uint8_t* dst = s.st_img + side * B * px;
    if (j.pitch == h->W && j.stride == (int64_t)px) HIP_TRY(hipMemcpyAsync(dst, src[side], (size_t)j.n * px, hipMemcpyHostToDevice, st));
    else
      for (int b = 0; b < j.n; b++)
        HIP_TRY(hipMemcpy2DAsync(dst + b * px, h->W, src[side] + (size_t)b * j.stride, j.pitch, h->W, h->H, hipMemcpyHostToDevice, st));
  }
  std::vector<int32_t> local(j.n, JN_OK);
  Job d = j;
  d.host = false; d.staged = true; d.dI1 = s.st_img; d.dI2 = s.st_img + B * px; d.pitch = h->W; d.stride = (int64_t)px;
  const size_t opx = h->sub ? (size_t)(h->W / 2) * (h->H / 2) : px;          // pixels of an output map
  d.dD1 = s.st_D; d.dD2 = s.st_D + B * px; d.status = local.data();
  const jn_status r = run_batch(h, s, d);                 // stream-ordered behind the copies; synchronises at its end
  if (r != JN_OK) return r;
  for (int b = 0; b < j.n;) {                             // runs of matched pairs go out together
    if (local[b] != JN_OK) { b++; continue; }
    int e = b;
    while (e < j.n && local[e] == JN_OK) e++;
    HIP_TRY(hipMemcpyAsync(j.hD1 + b * opx, s.st_D + b * opx, (size_t)(e - b) * opx * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(j.hD2 + b * opx, s.st_D + B * px + b * opx, (size_t)(e - b) * opx * sizeof(float), hipMemcpyDeviceToHost, st));
    b = e;
  }
  HIP_TRY(hipEventRecord(s.ev[EV_END], st));
  HIP_TRY(wait_event(s.ev[EV_END], h->wait_spin_us));
  if (j.status) for (int b = 0; b < j.n; b++) j.status[b] = local[b];
  return JN_OK;
}

void slot_loop(jn_elas* h, Slot* s) {
  pthread_setname_np(pthread_self(), "jn-slot");
  prctl(PR_SET_TIMERSLACK, 2000UL, 0, 0, 0);                 // the short sleeps of wait_event mean what they say (default slack: 50 us)
  hipSetDevice(h->device);
  for (;;) {
    Job j;
    {
      std::unique_lock<std::mutex> l(s->m);
      s->cv.wait(l, [s] { return s->quit || s->has_job; });
      if (s->quit) return;
      j = s->job; s->has_job = false;
    }
    const jn_status r = j.host ? run_batch_host(h, *s, j) : run_batch(h, *s, j);
    {
      std::lock_guard<std::mutex> l(s->m);
      s->result = r; s->busy = false;
    }
    s->cv.notify_all();
  }
}

template <typename T>
hipError_t dmalloc(T** p, size_t count) { return hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)); }

}  // namespace

extern "C" void jn_elas_destroy(jn_elas* h);

extern "C" {

const char* jn_version(void) { return "jn_stereo 0.3 (gfx950)"; }

uint64_t jn_fnv1a64_u32(const uint32_t* words, int64_t n) {
  uint64_t h = 1469598103934665603ull;
  for (int64_t i = 0; i < n; i++) h = (h ^ words[i]) * 1099511628211ull;
  return h;
}

void jn_elas_params_default(jn_elas_params* p, int32_t setting) {
  // elas.h:92-145
  p->disp_min = 0; p->disp_max = 255; p->support_texture = 10; p->candidate_stepsize = 5;
  p->incon_window_size = 5; p->incon_threshold = 5; p->incon_min_support = 5; p->grid_size = 20;
  p->beta = 0.02f; p->sigma = 1; p->lr_threshold = 2; p->speckle_sim_threshold = 1; p->speckle_size = 200;
  p->subsampling = 0;
  if (setting == JN_SETTING_ROBOTICS) {
    p->support_threshold = 0.85f; p->add_corners = 0; p->gamma = 3; p->sradius = 2; p->match_texture = 1;
    p->ipol_gap_width = 3; p->filter_median = 0; p->filter_adaptive_mean = 1; p->postprocess_only_left = 1;
  } else {
    p->support_threshold = 0.95f; p->add_corners = 1; p->gamma = 5; p->sradius = 3; p->match_texture = 0;
    p->ipol_gap_width = 5000; p->filter_median = 1; p->filter_adaptive_mean = 0; p->postprocess_only_left = 0;
  }
}

jn_status jn_device_count(int32_t* count) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) { *count = 0; return JN_ERR_NO_DEVICE; }
  *count = c;
  return c > 0 ? JN_OK : JN_ERR_NO_DEVICE;
}

jn_status jn_elas_create(const jn_elas_params* p, int32_t W, int32_t H, int32_t max_batch, int32_t device,
                         int32_t host_threads, int32_t slots, jn_elas** out) {
  if (!p || !out || W < 32 || H < 32 || W > 8192 || H > 8192 || max_batch < 1 || slots < 1) return JN_ERR_INVALID;
  *out = nullptr;
  const int radius = (int)std::max((float)std::ceil(p->sigma * p->sradius), 2.0f);        // elas.cpp:806
  if ((p->subsampling && ((W | H) & 1)) || p->disp_max > 255 || p->disp_max < 10 ||      // odd sizes with subsampling: the reference's half-size addressing runs over its rows
      p->disp_min > p->disp_max || p->ipol_gap_width < 0 || p->candidate_stepsize < 1 ||
      p->grid_size < 1 || radius > 7 || p->incon_window_size < 0)
    return JN_ERR_UNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return JN_ERR_NO_DEVICE;
  HIP_TRY(hipSetDevice(device));

  std::unique_ptr<jn_elas> h(new jn_elas());
  // any failure from here on releases whatever was allocated so far (jn_elas_destroy tolerates null buffers and
  // workers that were never started)
#define CREATE_TRY(expr)                                                                      \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      fprintf(stderr, "libjn_stereo: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      jn_elas_destroy(h.release());                                                           \
      return JN_ERR_NO_DEVICE;                                                                \
    }                                                                                         \
  } while (0)
  CREATE_TRY(configure_device_kernels());
  h->p = *p; h->W = W; h->H = H; h->max_batch = max_batch; h->device = device;
  DevParams& dp = h->dp;
  memset(&dp, 0, sizeof(dp));
  dp.W = W; dp.H = H; dp.pitch = (W + 63) / 64 * 64;
  dp.disp_max = p->disp_max; dp.disp_min = std::max(p->disp_min, 0); dp.support_texture = p->support_texture; dp.step = p->candidate_stepsize;
  h->sub = p->subsampling != 0;
  if (h->sub) dp.step += dp.step % 2;                                                    // elas.cpp:379-381: only even lines hold descriptors at half resolution
  dp.lr_threshold = p->lr_threshold; dp.support_threshold = p->support_threshold;
  dp.cw = (W + dp.step - 1) / dp.step; dp.ch = (H + dp.step - 1) / dp.step;            // elas.cpp:384-387
  dp.grid_size = p->grid_size;
  dp.grid_magic = p->grid_size > 1 ? (uint32_t)((1ull << 32) / (uint64_t)p->grid_size) + 1u : 0u;   // grid_size 1: kernels divide
  dp.gw = (int)std::ceil((float)W / (float)p->grid_size); dp.gh = (int)std::ceil((float)H / (float)p->grid_size);   // elas.cpp:90-91
  dp.match_texture = p->match_texture; dp.radius = radius;
  const float two_sigma_sq = 2 * p->sigma * p->sigma;
  for (int dd = 0; dd <= radius; dd++)                                                    // elas.cpp:802-805 (float math)
    dp.P[dd] = (int32_t)((-std::log(p->gamma + std::exp(-dd * dd / two_sigma_sq)) + std::log(p->gamma)) / p->beta);
  for (int dd = 0; dd <= radius; dd++)            // k_dense packs cost + prior into 24 bits of a key (bias 2^20)
    if (dp.P[dd] <= -(1 << 19) || dp.P[dd] >= (1 << 19)) return JN_ERR_UNSUPPORTED;
  dp.speckle_sim = p->speckle_sim_threshold; dp.speckle_size = p->speckle_size; dp.gap_width = p->ipol_gap_width;
  dp.add_corners = p->add_corners ? 1 : 0;
  if (h->sub) {                                         // the half-size maps' post-processing (elas.cpp:987-992, :1107-1112, :1292-1297, :1499-1504)
    h->dph = dp;
    h->dph.W = W / 2; h->dph.H = H / 2; h->dph.pitch = (W / 2 + 63) / 64 * 64;
    h->dph.speckle_size = (int32_t)(std::sqrt((float)p->speckle_size) * 2);
    h->dph.gap_width = p->ipol_gap_width / 2 + 1;
  }

  HostParams& hp = h->hp;
  hp.W = W; hp.H = H; hp.disp_max = p->disp_max; hp.step = dp.step; hp.incon_window_size = p->incon_window_size;
  hp.incon_threshold = p->incon_threshold; hp.incon_min_support = p->incon_min_support;
  hp.grid_size = p->grid_size; hp.gw = dp.gw; hp.gh = dp.gh; hp.cw = dp.cw; hp.ch = dp.ch;
  hp.add_corners = dp.add_corners;
  h->payload_cap = (HostWorker::payload_capacity(hp) + 255) / 256 * 256;
  h->tri_cap = 2 * (dp.cw * dp.ch + HostWorker::kCornerPoints) + 8;

  int nthreads = host_threads > 0 ? host_threads : usable_cpus();
  if (nthreads < 1) nthreads = 1;
  nthreads = std::min(nthreads, std::max(1, 8 * max_batch * slots));   // up to 2 sides x 4 parts per frame can run at once
  // latency-mode handles keep the pool threads that have just worked polling for 300 us (a lone pair's host stage is two 60 us tasks):
  // lone 640x480 pair 0.40 -> 0.35 ms.  (Running a synchronous call on the caller's thread instead of slot 0's worker was measured too: no gain.)
  int pool_spin = max_batch == 1 ? 300 : 0;
  if (const char* e = getenv("JN_POOL_SPIN_US")) pool_spin = atoi(e);
  h->pool.reset(new Pool(nthreads, hp, pool_spin));
  h->filter_min_batch = nthreads + 1;
  h->filters_fast = support_filters_fast(h->dp, p->incon_window_size, p->incon_min_support);
  // The plane data flow needs the LDS-staged forms of the two matching kernels; the parameter sets those do not take (support windows
  // beyond 2560 columns, grids below 8 pixels, priors beyond the keys' cost field) keep materialised descriptors and the kernels that read them.
  {
    const DescSrc probe{nullptr, 0, true};
    h->plane_flow = launch_support(nullptr, h->dp, max_batch, probe, nullptr, true) &&
                    launch_dense(nullptr, h->dp, max_batch, nullptr, nullptr, 0, nullptr, nullptr, nullptr, probe, nullptr, true);
    if (const char* e = getenv("JN_DESC_FLOW")) h->plane_flow = h->plane_flow && strcmp(e, "desc") != 0;
  }
  if (const char* e = getenv("JN_HOST_FILTERS")) h->filter_min_batch = atoi(e) ? (1 << 30) : 1;
  if (const char* e = getenv("JN_SPLIT_DELAUNAY")) h->split_delaunay = atoi(e) != 0;
  h->arr_cap = std::min(dp.cw * dp.ch, 8192);
  // Host route: sides with more vertices than k_arrange's 64-bit-key LDS form orders (8192; a 1920x1080 side has 11 k) are arranged on the
  // host (JN_ARRANGE_GLOBAL=1 in the hooks build sends them through the kernel's larger forms instead).  The GPU route (below) always
  // arranges on the device: 12288 vertices with compact keys in LDS (0.64 ms a 1080p batch), up to 16384 on global scratch (1.7 ms).
  h->arr_stride = (JN_HOOK_ENV("JN_ARRANGE_GLOBAL") && atoi(JN_HOOK_ENV("JN_ARRANGE_GLOBAL"))) ? std::min(dp.cw * dp.ch, 16384) : h->arr_cap;
  if (const char* e = JN_HOOK_ENV("JN_ARRANGE_SORTS")) h->arrange_sorts = atoi(e) != 0;
  h->gpu_arrange = !h->hp.add_corners;                     // the six corner points join the list on the host
  if (const char* e = getenv("JN_GPU_ARRANGE")) h->gpu_arrange = h->gpu_arrange && atoi(e) != 0;
  // Batch handles triangulate on the GPU as well (a latency-mode handle keeps the host stage: two pool threads finish a 640x480 pair's
  // two sides in 65 us, the kernel's serial top merges take longer than that); needs the device filters' list and the device arrangement.
  // (the kernels' FP64 predicates are exact for coordinates in (-2048, 2048); wider or taller images take their integer form)
  const bool gpu_dt_possible = max_batch > 1 && h->gpu_arrange && h->filters_fast;
  // Which of the two is faster depends on the host cores this process has (profiles/r05_gpu_delaunay_ab.txt, one MI355X): the kernel's top
  // merges are one thread each walking a seam through LDS (0.8-0.9 ms a batch, 42 bytes per vertex of every side held in LDS meanwhile):
  // 19.6 k pairs/s whatever the cores (0.3 busy); the host stage gives 22.0 k with ~10 busy cores where the scheduler may spread 16
  // threads over a whole socket, but 18.3 k pinned to 16 cores, 16.3 k to 12, 12.7 k to 8, 6.9 k to 4.  So: the GPU route for a process
  // PINNED to 16 cores or fewer (what a rank of a multi-GPU job gets: parallel.pin_rank) or with a CPU quota below 14, the host route
  // otherwise; JN_GPU_DELAUNAY=0/1 decides otherwise.
  {
    cpu_set_t set;
    const int pinned = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
    // (an explicit host_threads below 14 says the same thing — the caller's share of a quota that several ranks divide, which no rank can
    // see from its own affinity mask or cpu.max: bench.py passes quota / world)
    // Frames of 1920x1080 and beyond take the GPU route whatever the cores: their 11 k-point sides keep 12.8-14.4 host cores busy for
    // 4.9-5.7 k pairs/s, the kernels give 5.3-5.6 k with none (profiles/r06_full_hd_routes.txt, two boxes).
    h->gpu_delaunay = gpu_dt_possible && (pinned <= 16 || usable_cpus() < 14 || (host_threads > 0 && host_threads < 14) || (long long)W * H >= 1920LL * 1080);
  }
  if (const char* e = getenv("JN_GPU_DELAUNAY")) h->gpu_delaunay = gpu_dt_possible && atoi(e) != 0;
  // Sides with more support points than one workgroup's LDS holds (a 1920x1080 side has ~11 k) go through k_delaunay_sub / k_delaunay_top and
  // a global scratch (round 6); their arrangement then comes from k_arrange's compact-key LDS form (up to 12288 vertices a side) or its
  // global-scratch form (up to 16384).
  if (h->gpu_delaunay && dp.cw * dp.ch > delaunay_gpu_capacity(152 * 1024)) {
    h->dt_gcap = std::min(dp.cw * dp.ch, delaunay_gpu_max_points());
    h->arr_stride = std::max(h->arr_stride, std::min(dp.cw * dp.ch, 16384));
  }
  h->stage_events = max_batch > 1;
  h->wait_spin_us = max_batch > 1 ? 60 : 1000;
  if (const char* e = getenv("JN_WAIT_SPIN_US")) h->wait_spin_us = atoi(e);
  if (const char* e = getenv("JN_COMM_TIMEOUT_MS")) h->comm_timeout_ms = atoi(e);
  if (const char* e = JN_HOOK_ENV("JN_TEST_FAIL_SEQ")) h->test_fail_seq = atoll(e);
  if (const char* e = JN_HOOK_ENV("JN_TEST_SLOT_DELAY_US")) {
    for (const char* q = e; *q;) { h->test_slot_delay_us.push_back(atoi(q)); while (*q && *q != ',') q++; if (*q == ',') q++; }
  }
  h->pace = max_batch > 1 && slots > 1;
  if (const char* e = getenv("JN_PACE")) h->pace = atoi(e) != 0;
  h->zero_copy_payload = max_batch == 1;
  if (const char* e = getenv("JN_ZERO_COPY")) h->zero_copy_payload = atoi(e) != 0;
  if (const char* e = getenv("JN_STAGE_EVENTS")) h->stage_events = atoi(e) != 0;
  {
    int can_wait = 0;
    (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, device);
    h->gate_stage_b = max_batch == 1 && can_wait;
    if (const char* e = getenv("JN_GATE_STAGE_B")) h->gate_stage_b = atoi(e) != 0 && can_wait;
  }

  const size_t px = (size_t)W * H, B = (size_t)max_batch;
  for (int i = 0; i < slots; i++) {
    h->slots.emplace_back(new Slot());         // owned by the handle from the start: a failure below frees it too
    Slot* s = h->slots.back().get();
    CREATE_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    // Measured (profiles/r03_stage_a_priority_ab.txt): with stage A prioritised the pipelined 720p bench LOSES 12 % (17.5 k
    // against 20.2 k pairs/s) — the descriptor and support kernels of one slot then push the other slots' dense kernels
    // aside, and the GPU, not the host stage, is what the pipeline waits for.  Opt-in only: JN_STAGE_A_PRIORITY=1.
    if (getenv("JN_STAGE_A_PRIORITY") && atoi(getenv("JN_STAGE_A_PRIORITY")) != 0) {
      int least = 0, greatest = 0;
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
        CREATE_TRY(hipStreamCreateWithPriority(&s->stream_a, hipStreamNonBlocking, greatest));
    }
    // blocking-sync events: the slot worker sleeps while the GPU runs instead of spinning on a core that
    // the host stage (and, on a multi-GPU node, the other ranks) could use
    for (int e = 0; e < EV_COUNT; e++) CREATE_TRY(hipEventCreateWithFlags(&s->ev[e], hipEventBlockingSync));
    CREATE_TRY(hipEventCreate(&s->ev_scan)); CREATE_TRY(hipEventCreate(&s->ev_merged));
    CREATE_TRY(hipEventCreateWithFlags(&s->ev_head, hipEventDisableTiming));
    CREATE_TRY(hipEventCreate(&s->ev_owner));
    CREATE_TRY(dmalloc(&s->need_host, B));
    CREATE_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_need), B * sizeof(int32_t), hipHostMallocDefault));
    if (h->gate_stage_b) {                                   // no signal memory: the handle simply queues stage B after the host stage
      if (hipExtMallocWithFlags((void**)&s->gate, 8, hipMallocSignalMemory) == hipSuccess) { s->gate[0] = 0; s->gate[1] = 0; }
      else { (void)hipGetLastError(); s->gate = nullptr; }
    }
    if (h->plane_flow) CREATE_TRY(dmalloc(&s->planes, plane_bytes(W, H, 2 * (int)B) + 64));
    else CREATE_TRY(dmalloc(&s->desc, 2 * B * px));
    CREATE_TRY(dmalloc(&s->d_can, B * dp.cw * dp.ch));
    CREATE_TRY(dmalloc(&s->info, B)); CREATE_TRY(dmalloc(&s->payload, B * h->payload_cap));
    const size_t tiles = (size_t)((W + kTileW - 1) / kTileW) * ((H + kTileH - 1) / kTileH);
    CREATE_TRY(dmalloc(&s->bin_count, 2 * B * tiles)); CREATE_TRY(dmalloc(&s->bin_list, 2 * B * tiles * kBinCap));
    CREATE_TRY(dmalloc(&s->raw, 2 * B * px));
    CREATE_TRY(dmalloc(&s->tmp, B * px)); CREATE_TRY(dmalloc(&s->label, B * px)); CREATE_TRY(dmalloc(&s->size, B * px));
    CREATE_TRY(dmalloc(&s->scan_scratch, B * 4));
    CREATE_TRY(dmalloc(&s->d_flat, B * (1024 + 4)));
    const size_t grid_words = 2 * B * dp.gw * dp.gh * kGridWords;
    CREATE_TRY(dmalloc(&s->mark, grid_words)); CREATE_TRY(dmalloc(&s->gridbits, grid_words));
    CREATE_TRY(dmalloc(&s->recs, 2 * B * (size_t)h->tri_cap));
    s->scratch.resize(B);
    s->sides.resize(2 * B);
    CREATE_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_can), B * dp.cw * dp.ch * sizeof(int16_t), hipHostMallocDefault));
    CREATE_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_info), B * sizeof(FrameInfo), hipHostMallocDefault));
    CREATE_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_payload), B * h->payload_cap, hipHostMallocDefault));
    CREATE_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_list), B * dp.cw * dp.ch * 3 * sizeof(int16_t), hipHostMallocDefault));
    CREATE_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_cnt), B * sizeof(int32_t), hipHostMallocDefault));
    CREATE_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_arr), B * 2 * (size_t)h->arr_stride * sizeof(uint16_t), hipHostMallocDefault));
    if (h->arr_stride > h->arr_cap) CREATE_TRY(hipMalloc(&s->arr_scratch, arrange_scratch_bytes((int)B, h->arr_stride)));
    CREATE_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_arr_ok), B * 2 * sizeof(int32_t), hipHostMallocDefault));
    if (h->gpu_delaunay) {
      CREATE_TRY(dmalloc(&s->d_list, B * dp.cw * dp.ch * 3)); CREATE_TRY(dmalloc(&s->d_cnt, B));
      CREATE_TRY(dmalloc(&s->d_arr, B * 2 * (size_t)h->arr_stride)); CREATE_TRY(dmalloc(&s->d_arr_ok, B * 2));
      CREATE_TRY(hipMemset(s->payload, 0, B * h->payload_cap));
      if (h->dt_gcap) CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&s->dt_scratch), delaunay_gpu_scratch_bytes((int)B, h->dt_gcap)));      // (a side k_delaunay hands back leaves its part unwritten: never uninitialised memory)
    }
  }
  h->s_pitch = dp.pitch;
  CREATE_TRY(dmalloc(&h->s_img, 2 * (size_t)H * dp.pitch));
  CREATE_TRY(dmalloc(&h->s_D, 2 * px));
  for (auto& s : h->slots) s->th = std::thread(slot_loop, h.get(), s.get());
  *out = h.release();
  return JN_OK;
#undef CREATE_TRY
}

void jn_elas_destroy(jn_elas* h) {
  if (!h) return;
  hipSetDevice(h->device);
  for (auto& s : h->slots) {
    { std::unique_lock<std::mutex> l(s->m); s->cv.wait(l, [&] { return !s->busy; }); s->quit = true; }
    s->cv.notify_all();
    if (s->th.joinable()) s->th.join();
  }
  hipSetDevice(h->device);
  for (auto& s : h->slots) {
    hipFree(s->desc); hipFree(s->planes); hipFree(s->d_can); hipFree(s->info); hipFree(s->payload);
    hipFree(s->bin_count); hipFree(s->bin_list); hipFree(s->raw); hipFree(s->tmp); hipFree(s->label); hipFree(s->size); hipFree(s->scan_scratch); hipFree(s->d_flat); hipFree(s->st_img); hipFree(s->st_D); hipFree(s->arr_scratch);
    hipFree(s->mark); hipFree(s->gridbits); hipFree(s->recs);
    hipHostFree(s->h_can); hipHostFree(s->h_info); hipHostFree(s->h_payload); hipHostFree(s->h_list); hipHostFree(s->h_cnt); hipHostFree(s->h_arr); hipHostFree(s->h_arr_ok);
    for (int e = 0; e < EV_COUNT; e++) if (s->ev[e]) hipEventDestroy(s->ev[e]);
    if (s->ev_scan) hipEventDestroy(s->ev_scan);
    if (s->ev_merged) hipEventDestroy(s->ev_merged);
    if (s->ev_head) hipEventDestroy(s->ev_head);
    if (s->ev_owner) hipEventDestroy(s->ev_owner);
    hipFree(s->need_host); if (s->h_need) hipHostFree(s->h_need);
    hipFree(s->d_list); hipFree(s->d_cnt); hipFree(s->d_arr); hipFree(s->d_arr_ok); hipFree(s->dt_scratch);
    if (s->gate) hipFree(s->gate);
    if (s->stream_a) hipStreamDestroy(s->stream_a);
    if (s->stream) hipStreamDestroy(s->stream);
  }
  hipFree(h->s_img); hipFree(h->s_D);
  h->pool.reset();
  delete h;
}

static bool stride_ok(const jn_elas* h, int32_t n, int32_t pitch, int64_t image_stride) {
  return n == 1 || image_stride >= (int64_t)pitch * h->H;      // image b starts at base + b*image_stride: images must not overlap
}

jn_status jn_elas_submit(jn_elas* h, int32_t slot, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch,
                         int64_t image_stride, float* dD1, float* dD2, int32_t* status) {
  if (!h || slot < 0 || slot >= (int)h->slots.size() || n < 1 || n > h->max_batch || !dI1 || !dI2 || !dD1 || !dD2 ||
      pitch < h->W || !stride_ok(h, n, pitch, image_stride))
    return JN_ERR_INVALID;
  Slot& s = *h->slots[slot];
  {
    std::unique_lock<std::mutex> l(s.m);
    s.cv.wait(l, [&] { return !s.busy; });
    s.job = Job{n, dI1, dI2, pitch, image_stride, dD1, dD2, status};
    s.has_job = true; s.busy = true;
  }
  s.cv.notify_all();
  return JN_OK;
}

jn_status jn_elas_submit_host(jn_elas* h, int32_t slot, int32_t n, const uint8_t* I1, const uint8_t* I2, int32_t pitch,
                              int64_t image_stride, float* D1, float* D2, int32_t* status) {
  if (!h || slot < 0 || slot >= (int)h->slots.size() || n < 1 || n > h->max_batch || !I1 || !I2 || !D1 || !D2 || pitch < h->W ||
      !stride_ok(h, n, pitch, image_stride))
    return JN_ERR_INVALID;
  Slot& s = *h->slots[slot];
  {
    std::unique_lock<std::mutex> l(s.m);
    s.cv.wait(l, [&] { return !s.busy; });
    s.job = Job{};
    s.job.n = n; s.job.pitch = pitch; s.job.stride = image_stride; s.job.status = status;
    s.job.host = true; s.job.hI1 = I1; s.job.hI2 = I2; s.job.hD1 = D1; s.job.hD2 = D2;
    s.has_job = true; s.busy = true;
  }
  s.cv.notify_all();
  return JN_OK;
}

jn_status jn_elas_submit_scan(jn_elas* h, int32_t slot, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch,
                              int64_t image_stride, float* dD1, float* dD2, const jn_scan_params* sp, const uint8_t* dLut,
                              uint8_t* dDispU8, double* dBins, double* dMeta, int32_t* status) {
  if (!h || slot < 0 || slot >= (int)h->slots.size() || n < 1 || n > h->max_batch || !dI1 || !dI2 || !dD1 || !dD2 ||
      pitch < h->W || !stride_ok(h, n, pitch, image_stride) || !sp || !dLut || !dDispU8 || !dBins || !dMeta || sp->bins < 1 || sp->bins > 1024)
    return JN_ERR_INVALID;
  if (h->sub) return JN_ERR_UNSUPPORTED;                  // the node's tail works on full-size maps (its Q matrix and LUT are the image's)
  Slot& s = *h->slots[slot];
  {
    std::unique_lock<std::mutex> l(s.m);
    s.cv.wait(l, [&] { return !s.busy; });
    s.job = Job{n, dI1, dI2, pitch, image_stride, dD1, dD2, status};
    s.job.scan = true; s.job.sp = *sp; s.job.dLut = dLut; s.job.dDispU8 = dDispU8; s.job.dBins = dBins; s.job.dMeta = dMeta;
    {
      std::lock_guard<std::mutex> g(h->merge_m);             // the submitting thread numbers the batches: same order on every rank
      if (h->comm) { s.job.merge = true; s.job.seq = h->submit_seq++; }
    }
    s.has_job = true; s.busy = true;
  }
  s.cv.notify_all();
  return JN_OK;
}

jn_status jn_elas_set_comm(jn_elas* h, jn_comm* c) {
  if (!h) return JN_ERR_INVALID;
  if (c && comm_device(c) != h->device) return JN_ERR_INVALID;
  for (auto& sp_ : h->slots) {                              // no batch in flight
    std::unique_lock<std::mutex> l(sp_->m);
    sp_->cv.wait(l, [&] { return !sp_->busy; });
  }
  std::lock_guard<std::mutex> g(h->merge_m);
  if (c && comm_dead(c)) return JN_ERR_COMM;
  h->comm = c; h->submit_seq = 0; h->merge_seq = 0; h->merge_log.clear();
  return JN_OK;
}

int32_t jn_elas_merge_order(jn_elas* h, uint64_t* out, int32_t cap) {
  if (!h || !out || cap < 1) return 0;
  std::lock_guard<std::mutex> g(h->merge_m);
  const size_t k = std::min<size_t>(h->merge_log.size(), (size_t)cap);
  std::copy(h->merge_log.end() - k, h->merge_log.end(), out);
  return (int32_t)k;
}

jn_status jn_elas_route_stats(jn_elas* h, int32_t slot, int32_t out[3]) {
  if (!h || !out || slot < 0 || slot >= (int)h->slots.size()) return JN_ERR_INVALID;
  out[0] = h->gpu_delaunay ? 1 : 0; out[1] = (int32_t)h->slots[slot]->gpu_dt_fallbacks; out[2] = h->plane_flow ? 1 : 0;
  return JN_OK;
}

jn_status jn_elas_bin_stats(jn_elas* h, int32_t slot, int32_t out[3]) {
  if (!h || !out || slot < 0 || slot >= (int)h->slots.size()) return JN_ERR_INVALID;
  Slot& s = *h->slots[slot];
  out[0] = out[1] = out[2] = 0;
  if (s.last_n < 1) return JN_OK;
  HIP_TRY(hipSetDevice(h->device));
  const size_t tiles = (size_t)((h->W + kTileW - 1) / kTileW) * ((h->H + kTileH - 1) / kTileH), count = (size_t)s.last_n * 2 * tiles;
  std::vector<int32_t> c(count);
  HIP_TRY(hipMemcpy(c.data(), s.bin_count, count * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (int32_t x : c) { out[0] = std::max(out[0], x); out[1] += x > (int32_t)kBinLds; out[2] += x > (int32_t)kBinCap; }
  return JN_OK;
}

jn_status jn_elas_merge_time(jn_elas* h, int32_t slot, float* ms) {
  if (!h || !ms || slot < 0 || slot >= (int)h->slots.size()) return JN_ERR_INVALID;
  *ms = h->slots[slot]->merge_ms;
  return JN_OK;
}

jn_status jn_elas_wait(jn_elas* h, int32_t slot) {
  if (!h || slot < 0 || slot >= (int)h->slots.size()) return JN_ERR_INVALID;
  Slot& s = *h->slots[slot];
  std::unique_lock<std::mutex> l(s.m);
  s.cv.wait(l, [&] { return !s.busy; });
  return s.result;
}

jn_status jn_elas_process_batch(jn_elas* h, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch,
                                int64_t image_stride, float* dD1, float* dD2, int32_t* status) {
  const jn_status r = jn_elas_submit(h, 0, n, dI1, dI2, pitch, image_stride, dD1, dD2, status);
  if (r != JN_OK) return r;
  return jn_elas_wait(h, 0);
}

jn_status jn_elas_process(jn_elas* h, const uint8_t* I1, const uint8_t* I2, float* D1, float* D2, const int32_t dims[3]) {
  if (!h || !I1 || !I2 || !D1 || !D2 || !dims) return JN_ERR_INVALID;
  if (dims[0] != h->W || dims[1] != h->H || dims[2] < dims[0]) return JN_ERR_INVALID;
  std::lock_guard<std::mutex> guard(h->api_m);
  HIP_TRY(hipSetDevice(h->device));
  const size_t img = (size_t)h->H * h->s_pitch, px = (size_t)h->W * h->H;
  HIP_TRY(hipMemcpy2D(h->s_img, h->s_pitch, I1, dims[2], h->W, h->H, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy2D(h->s_img + img, h->s_pitch, I2, dims[2], h->W, h->H, hipMemcpyHostToDevice));
  int32_t st = JN_OK;
  const jn_status r = jn_elas_process_batch(h, 1, h->s_img, h->s_img + img, h->s_pitch, 0, h->s_D, h->s_D + px, &st);
  if (r != JN_OK) return r;
  if (st != JN_OK) {                        // elas.cpp:66-71: message, outputs untouched
    printf("ERROR: Need at least 3 support points!\n");
    return (jn_status)st;
  }
  const size_t opx = h->sub ? (size_t)(h->W / 2) * (h->H / 2) : px;          // elas.h:160-162: half-size maps with subsampling
  HIP_TRY(hipMemcpy(D1, h->s_D, opx * sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(D2, h->s_D + px, opx * sizeof(float), hipMemcpyDeviceToHost));
  return JN_OK;
}

jn_status jn_elas_last_times(jn_elas* h, int32_t slot, jn_stage_times* out) {
  if (!h || !out || slot < 0 || slot >= (int)h->slots.size()) return JN_ERR_INVALID;
  *out = h->slots[slot]->times;
  return JN_OK;
}

jn_status jn_elas_kernel_time(jn_elas* h, int32_t slot, const char* kernel, float* avg_ms, int32_t* launches) {
  if (!h || !kernel || !avg_ms || !launches || slot < 0 || slot >= (int)h->slots.size()) return JN_ERR_INVALID;
  const std::string k(kernel);
  if (k == "k_dense" || k == "k_dense_row") *avg_ms = h->slots[slot]->dense_ms;      // the dense matcher of this handle's data flow
  else if (k == "k_owner") *avg_ms = h->slots[slot]->owner_ms;
  else return JN_ERR_INVALID;
  *launches = h->slots[slot]->dense_launches;
  return JN_OK;
}

// ---- seam B2 ------------------------------------------------------------------------------------

void jn_scan_params_default(jn_scan_params* sp, int32_t W, int32_t H) {
  // K1 / T of calibration/amrl_jackal_webcam_stereo.yml (calibrated at 640x360, point_cloud.cpp:38),
  // scaled to the working size; Q in the zero-disparity form stereoRectify emits.
  const double sx = (double)W / 640.0, sy = (double)H / 360.0;
  const double f = 4.6417933392659904e+02 * sx, cx = 3.2479711799310849e+02 * sx, cy = 1.8685472713963392e+02 * sy;
  const double Tx = -9.4052586442980660e-02;
  const double Q[16] = {1, 0, 0, -cx, 0, 1, 0, -cy, 0, 0, 0, f, 0, 0, -1.0 / Tx, 0};
  memcpy(sp->Q, Q, sizeof(Q));
  const double XR[9] = {-0.0007962732853436516, -0.2675000227968607, 0.9635575706420958,
                        -0.9999984502796089, -0.001321509725770019, -0.00119326128710218,
                        0.001592547981999815, -0.9635569909380592, -0.2674985457970802};
  memcpy(sp->XR, XR, sizeof(XR));
  sp->XT[0] = 0; sp->XT[1] = 0; sp->XT[2] = 0.28;
  sp->crop_offset_x = 0; sp->crop_offset_y = 0;
  sp->gp_height_thresh = 0.05; sp->gp_angle_thresh = 4. * 3.1415 / 180.; sp->gp_dist_thresh = 1.0;
  sp->fov_deg = 90.; sp->bins = 90; sp->pi_approx = 3.1415;
}

jn_status jn_disparity_to_u8(int32_t device, const float* dD, uint8_t* dOut, int64_t n) {
  if (!dD || !dOut || n < 0) return JN_ERR_INVALID;
  HIP_TRY(hipSetDevice(device));
  if (n) launch_to_u8(nullptr, dD, dOut, n);
  HIP_TRY(hipStreamSynchronize(nullptr));
  HIP_TRY(hipGetLastError());
  return JN_OK;
}

jn_status jn_build_valid_disp_lut(int32_t device, const jn_scan_params* sp, int32_t W, int32_t H, uint8_t* dLut) {
  if (!sp || !dLut || W < 1 || H < 1) return JN_ERR_INVALID;
  HIP_TRY(hipSetDevice(device));
  launch_valid_lut(nullptr, *sp, W, H, dLut);
  HIP_TRY(hipStreamSynchronize(nullptr));
  HIP_TRY(hipGetLastError());
  return JN_OK;
}

static jn_status scan_common(int32_t device, const jn_scan_params* sp, int32_t n, const float* dD, uint8_t* dDisp,
                             const uint8_t* dLut, int32_t W, int32_t H, double* dBins, double* dMeta) {
  if (!sp || !dDisp || !dBins || !dMeta || n < 1 || sp->bins < 1 || sp->bins > 1024) return JN_ERR_INVALID;
  HIP_TRY(hipSetDevice(device));
  // grow-only scratch per device and calling thread: no hipMalloc/hipFree (a device-wide sync) per call
  struct Scratch { unsigned long long* p = nullptr; int cap = 0; int dev = -1; };
  static thread_local Scratch sc;
  if (sc.dev != device || sc.cap < n) {
    if (sc.p) { hipSetDevice(sc.dev); hipFree(sc.p); hipSetDevice(device); sc.p = nullptr; }
    const int cap = n > 64 ? n : 64;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sc.p), sizeof(unsigned long long) * 4 * cap));
    sc.cap = cap; sc.dev = device;
  }
  launch_scan(nullptr, *sp, n, dD, dDisp, dLut, W, H, dBins, dMeta, sc.p);
  HIP_TRY(hipStreamSynchronize(nullptr));
  HIP_TRY(hipGetLastError());
  return JN_OK;
}

jn_status jn_obstacle_scan(int32_t device, const jn_scan_params* sp, int32_t n, const uint8_t* dDisp, const uint8_t* dLut,
                           int32_t W, int32_t H, double* dBins, double* dMeta) {
  if (!dLut) return JN_ERR_INVALID;
  return scan_common(device, sp, n, nullptr, const_cast<uint8_t*>(dDisp), dLut, W, H, dBins, dMeta);
}

jn_status jn_obstacle_scan_cloud(int32_t device, const jn_scan_params* sp, int32_t n, const uint8_t* dDisp, int32_t W, int32_t H,
                                 double* dBins, double* dMeta) {
  return scan_common(device, sp, n, nullptr, const_cast<uint8_t*>(dDisp), nullptr, W, H, dBins, dMeta);
}

jn_status jn_disparity_scan(int32_t device, const jn_scan_params* sp, int32_t n, const float* dD, const uint8_t* dLut,
                            int32_t W, int32_t H, uint8_t* dDispU8, double* dBins, double* dMeta) {
  if (!dD || !dLut) return JN_ERR_INVALID;
  return scan_common(device, sp, n, dD, dDispU8, dLut, W, H, dBins, dMeta);
}

int32_t jn_compact_ranges(const double* bins, int32_t nbins, float* ranges) {
  int32_t k = 0;
  for (int i = nbins - 1; i >= 0; i--)                      // point_cloud.cpp:278-282
    if (bins[i] < JN_SCAN_EMPTY - 1) ranges[k++] = (float)bins[i];
  return k;
}

jn_status jn_point_cloud(int32_t device, const jn_scan_params* sp, const uint8_t* dDisp, int32_t W, int32_t H, float* dXyz,
                         int64_t* count) {
  if (!sp || !dDisp || !dXyz || !count || W < 1 || H < 1) return JN_ERR_INVALID;
  HIP_TRY(hipSetDevice(device));
  long long* cols = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&cols), sizeof(long long) * (W + 1)));
  launch_point_cloud(nullptr, *sp, dDisp, W, H, dXyz, cols);
  long long total = 0;
  hipError_t e = hipMemcpy(&total, cols + W, sizeof(long long), hipMemcpyDeviceToHost);
  hipFree(cols);
  HIP_TRY(e);
  HIP_TRY(hipGetLastError());
  *count = total;
  return JN_OK;
}

// ---- rectification front end -----------------------------------------------------------------------
jn_status jn_init_undistort_rectify_map(int32_t device, const double K[9], const double D[5], const double R[9], const double P[12],
                                        int32_t W, int32_t H, float* dMapX, float* dMapY) {
  if (!K || !D || !R || !P || !dMapX || !dMapY || W < 1 || H < 1) return JN_ERR_INVALID;
  // iR = inverse(P[:, :3] * R), by cofactors
  double M[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M[3 * i + j] = P[4 * i] * R[j] + P[4 * i + 1] * R[3 + j] + P[4 * i + 2] * R[6 + j];
  const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
  const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
  if (det == 0.0) return JN_ERR_INVALID;
  const double id = 1.0 / det;
  const double iR[9] = {c00 * id, (M[2] * M[7] - M[1] * M[8]) * id, (M[1] * M[5] - M[2] * M[4]) * id,
                        c01 * id, (M[0] * M[8] - M[2] * M[6]) * id, (M[2] * M[3] - M[0] * M[5]) * id,
                        c02 * id, (M[1] * M[6] - M[0] * M[7]) * id, (M[0] * M[4] - M[1] * M[3]) * id};
  HIP_TRY(hipSetDevice(device));
  launch_undistort_map(nullptr, iR, K, D, W, H, dMapX, dMapY);
  HIP_TRY(hipStreamSynchronize(nullptr));
  HIP_TRY(hipGetLastError());
  return JN_OK;
}

jn_status jn_remap_bilinear(int32_t device, int32_t n, const uint8_t* dSrc, int32_t sw, int32_t sh, int32_t spitch, int64_t sstride,
                            const float* dMapX, const float* dMapY, uint8_t* dDst, int32_t W, int32_t H, int32_t dpitch, int64_t dstride) {
  if (!dSrc || !dMapX || !dMapY || !dDst || n < 1 || sw < 1 || sh < 1 || W < 1 || H < 1 || spitch < sw || dpitch < W) return JN_ERR_INVALID;
  HIP_TRY(hipSetDevice(device));
  launch_remap(nullptr, n, dSrc, sw, sh, spitch, sstride, dMapX, dMapY, dDst, W, H, dpitch, dstride);
  HIP_TRY(hipStreamSynchronize(nullptr));
  HIP_TRY(hipGetLastError());
  return JN_OK;
}

// ---- host-stage hooks -----------------------------------------------------------------------------
int32_t jn_host_triangulate(const int32_t* x, const int32_t* y, int32_t n, int32_t* tri) {
  if (!x || !y || !tri || n < 0) return -1;
  Delaunay dt;
  return dt.run(x, y, n, tri);
}

int32_t jn_host_triangulate_parts(const int32_t* x, const int32_t* y, int32_t n, int32_t* tri, int32_t parts) {
  if (!x || !y || !tri || n < 0) return -1;
  Delaunay dt;
  const int got = dt.prepare(x, y, n, parts);
  if (got == 0) return -1;
  std::vector<std::thread> th;                               // the parts really run concurrently
  for (int i = 1; i < got; i++) th.emplace_back([&dt, i] { dt.subtree(i); });
  dt.subtree(0);
  for (auto& t : th) t.join();
  return dt.finish(tri);
}

int32_t jn_host_arrangement(const int32_t* x, const int32_t* y, int32_t n, uint16_t* out) {
  if (!x || !y || !out || n < 0) return -1;
  Delaunay dt;
  return dt.arrangement(x, y, n, out) ? 1 : 0;
}

// bounds of a list of (uc, vc, d) triples for k_arrange's rank form (the product passes what the handle's lattice and disparity range give)
static bool arrange_by_sorts() { const char* e = JN_HOOK_ENV("JN_ARRANGE_SORTS"); return e && atoi(e) != 0; }   // hooks build: the sort forms where the rank form would run
static ArrBounds bounds_of(const int16_t* t, int n, int step) {
  if (n <= 0 || arrange_by_sorts()) return ArrBounds{0, 0, 0, 0};
  int ucm = 0, vcm = 0, xlo = 1 << 30, xhi = -(1 << 30);
  for (int i = 0; i < n; i++) {
    const int uc = t[3 * i], vc = t[3 * i + 1], x = uc * step - t[3 * i + 2];
    if (uc < 0 || vc < 0) return ArrBounds{0, 0, 0, 0};
    ucm = std::max(ucm, uc); vcm = std::max(vcm, vc); xlo = std::min(xlo, x); xhi = std::max(xhi, x);
  }
  return ArrBounds{vcm + 1, ucm + 1, xlo, xhi - xlo + 1};
}
jn_status jn_device_arrangement(int32_t device, const int16_t* triples, int32_t n, int32_t step, uint16_t* left, uint16_t* right,
                                int32_t ok[2]) {
  if (!triples || !left || !right || !ok || n < 0 || step < 1) return JN_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return JN_ERR_NO_DEVICE;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(configure_device_kernels());
  const int cap = std::max(n, 1), arr_cap = std::min(cap, 8192), g_cap = std::min(cap, 16384);
  int16_t* d_list = nullptr; int32_t* d_cnt = nullptr; uint16_t* d_arr = nullptr; int32_t* d_ok = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_list), (size_t)cap * 3 * sizeof(int16_t));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_cnt), sizeof(int32_t));
  void* d_scratch = nullptr;
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_arr), (size_t)2 * g_cap * sizeof(uint16_t));
  if (e == hipSuccess && g_cap > arr_cap) e = hipMalloc(&d_scratch, arrange_scratch_bytes(1, g_cap));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_ok), 2 * sizeof(int32_t));
  if (e == hipSuccess && n) e = hipMemcpy(d_list, triples, (size_t)n * 3 * sizeof(int16_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_cnt, &n, sizeof(int32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) { launch_arrange(nullptr, 1, d_list, d_cnt, cap, step, arr_cap, g_cap, d_arr, d_ok, d_scratch, d_scratch ? g_cap : 0, bounds_of(triples, n, step)); e = hipStreamSynchronize(nullptr); }
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(ok, d_ok, 2 * sizeof(int32_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess && ok[0]) e = hipMemcpy(left, d_arr, (size_t)n * sizeof(uint16_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess && ok[1]) e = hipMemcpy(right, d_arr + g_cap, (size_t)n * sizeof(uint16_t), hipMemcpyDeviceToHost);
  hipFree(d_list); hipFree(d_cnt); hipFree(d_arr); hipFree(d_ok); hipFree(d_scratch);
  HIP_TRY(e);
  return JN_OK;
}

jn_status jn_device_triangulate(int32_t device, const int16_t* triples, int32_t n, int32_t step, int32_t* tri_left, int32_t* tri_right, int32_t ntri[2],
                                int32_t* need_host) {
  if (!triples || !tri_left || !tri_right || !ntri || !need_host || n < 0 || step < 1) return JN_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return JN_ERR_NO_DEVICE;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(configure_device_kernels());
  const int cap = std::max(n, 1), arr_cap = std::min(cap, 8192), g_cap = std::min(cap, 16384);
  const int whole = delaunay_gpu_capacity(152 * 1024);
  const size_t pay = (size_t)cap * 12 + 2 * (2 * (size_t)cap + 8) * 12 + 256;
  int16_t* d_list = nullptr; int32_t* d_cnt = nullptr; uint16_t* d_arr = nullptr; int32_t* d_ok = nullptr; uint8_t* d_pay = nullptr; FrameInfo* d_info = nullptr; int32_t* d_need = nullptr;
  void* d_ascr = nullptr; uint8_t* d_dscr = nullptr;            // more vertices than the LDS forms take: the arrangement's and the triangulation's global scratch
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_list), (size_t)cap * 3 * sizeof(int16_t));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_cnt), sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_arr), (size_t)2 * g_cap * sizeof(uint16_t));
  if (e == hipSuccess && g_cap > arr_cap) e = hipMalloc(&d_ascr, arrange_scratch_bytes(1, g_cap));
  if (e == hipSuccess && n > whole) e = hipMalloc(reinterpret_cast<void**>(&d_dscr), delaunay_gpu_scratch_bytes(1, g_cap));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_ok), 2 * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_pay), pay);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_info), sizeof(FrameInfo));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_need), sizeof(int32_t));
  if (e == hipSuccess && n) e = hipMemcpy(d_list, triples, (size_t)n * 3 * sizeof(int16_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_cnt, &n, sizeof(int32_t), hipMemcpyHostToDevice);
  FrameInfo fi;
  memset(&fi, 0, sizeof(fi));
  if (e == hipSuccess) {
    launch_arrange(nullptr, 1, d_list, d_cnt, cap, step, arr_cap, g_cap, d_arr, d_ok, d_ascr, d_ascr ? g_cap : 0, bounds_of(triples, n, step));
    long long* d_clk = nullptr;
    const bool want_clk = JN_HOOK_ENV("JN_DT_CLOCKS") != nullptr;
    if (want_clk && hipMalloc(reinterpret_cast<void**>(&d_clk), 64 * sizeof(long long)) == hipSuccess) hipMemset(d_clk, 0, 64 * sizeof(long long));
    bool wide = false;                                       // coordinates beyond (-2048, 2048): the integer predicates
    for (int i = 0; i < n; i++) wide |= triples[3 * i] * step >= 2048 || triples[3 * i + 1] * step >= 2048;
    launch_delaunay(nullptr, 1, d_list, d_cnt, cap, step, d_arr, d_ok, g_cap, d_dscr ? n : whole, d_pay, (long long)pay, d_info, d_need, d_clk, d_dscr, d_dscr ? g_cap : 0, 0, wide);
    e = hipStreamSynchronize(nullptr);
    if (d_clk) {                                             // JN_DT_CLOCKS: microseconds per tree level (leaves first) of both sides, to stderr
      long long clk[64];
      if (hipMemcpy(clk, d_clk, sizeof(clk), hipMemcpyDeviceToHost) == hipSuccess)
        for (int sd = 0; sd < 2; sd++) {
          fprintf(stderr, "k_delaunay n=%d side %d, us per level from the leaves up:", n, sd);
          long long prev = clk[sd * 32 + 31];
          for (int k = 30; k >= 0; k--) if (clk[sd * 32 + k]) { fprintf(stderr, " %.1f", (clk[sd * 32 + k] - prev) / 100.0); prev = clk[sd * 32 + k]; }
          fprintf(stderr, "  total %.1f\n", (prev - clk[sd * 32 + 31]) / 100.0);
        }
      hipFree(d_clk);
    }
  }
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(&fi, d_info, sizeof(fi), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(need_host, d_need, sizeof(int32_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess) {
    ntri[0] = fi.ntri[0]; ntri[1] = fi.ntri[1];
    if (fi.ntri[0] > 0) e = hipMemcpy(tri_left, d_pay + fi.corner_offset[0], (size_t)fi.ntri[0] * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess && fi.ntri[1] > 0) e = hipMemcpy(tri_right, d_pay + fi.corner_offset[1], (size_t)fi.ntri[1] * 12, hipMemcpyDeviceToHost);
  }
  hipFree(d_list); hipFree(d_cnt); hipFree(d_arr); hipFree(d_ok); hipFree(d_pay); hipFree(d_info); hipFree(d_need); hipFree(d_ascr); hipFree(d_dscr);
  HIP_TRY(e);
  return JN_OK;
}

jn_status jn_device_support_filters(int32_t device, const jn_elas_params* p, int32_t W, int32_t H, int32_t n, int16_t* d_can,
                                    int32_t form) {
  if (!p || !d_can || n < 1 || W < 1 || H < 1 || p->candidate_stepsize < 1 || form < 0 || form > 2) return JN_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return JN_ERR_NO_DEVICE;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(configure_device_kernels());
  DevParams dp;
  memset(&dp, 0, sizeof(dp));
  dp.W = W; dp.H = H; dp.step = p->candidate_stepsize;
  dp.cw = (W + dp.step - 1) / dp.step; dp.ch = (H + dp.step - 1) / dp.step;
  const size_t cells = (size_t)n * dp.cw * dp.ch;
  if (form == 2 && !support_filters_fast(dp, p->incon_window_size, p->incon_min_support)) return JN_ERR_UNSUPPORTED;
  int16_t* d = nullptr; uint8_t* scratch = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), cells * sizeof(int16_t)));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&scratch), cells);
  if (e != hipSuccess) { hipFree(d); HIP_TRY(e); }
  e = hipMemcpy(d, d_can, cells * sizeof(int16_t), hipMemcpyHostToDevice);
  bool ran = false;
  if (e == hipSuccess) {
    ran = launch_support_filters(nullptr, dp, n, p->incon_window_size, p->incon_threshold, p->incon_min_support, d,
                                 form == 1 ? nullptr : scratch);       // no scratch: only the wavefront kernel can run
    if (ran) e = hipMemcpy(d_can, d, cells * sizeof(int16_t), hipMemcpyDeviceToHost);
  }
  hipFree(d); hipFree(scratch);
  HIP_TRY(e);
  HIP_TRY(hipGetLastError());
  return ran ? JN_OK : JN_ERR_UNSUPPORTED;
}

static_assert(sizeof(jn_host_frame_info) == sizeof(FrameInfo), "jn_host_frame_info mirrors FrameInfo");

int64_t jn_host_stage(const jn_elas_params* p, int32_t W, int32_t H, int16_t* d_can, uint8_t* payload, int64_t payload_cap,
                      jn_host_frame_info* info) {
  if (!p || !d_can || !payload || !info || p->candidate_stepsize < 1 || p->grid_size < 1) return -1;
  HostParams hp;
  hp.W = W; hp.H = H; hp.disp_max = p->disp_max; hp.step = p->candidate_stepsize;
  hp.incon_window_size = p->incon_window_size; hp.incon_threshold = p->incon_threshold;
  hp.incon_min_support = p->incon_min_support; hp.grid_size = p->grid_size;
  hp.gw = (int)std::ceil((float)W / (float)p->grid_size); hp.gh = (int)std::ceil((float)H / (float)p->grid_size);
  hp.cw = (W + hp.step - 1) / hp.step; hp.ch = (H + hp.step - 1) / hp.step;
  hp.add_corners = p->add_corners ? 1 : 0;
  if ((int64_t)HostWorker::payload_capacity(hp) > payload_cap) return -1;
  HostWorker w(hp);
  FrameInfo fi;
  FrameScratch fs;
  w.filter_and_list(d_can, &fi, &fs);
  HostWorker::place(&fi, 0);
  w.triangulate_side(0, fs, payload, &fi);
  w.triangulate_side(1, fs, payload, &fi);
  memcpy(info, &fi, sizeof(fi));
  return fi.ok ? fi.corner_offset[1] + (int64_t)fi.ntri[1] * 3 * (int64_t)sizeof(int32_t) : 0;
}

// ---- device helpers -------------------------------------------------------------------------------
jn_status jn_device_malloc(int32_t device, int64_t bytes, void** out) {
  if (!out || bytes < 0) return JN_ERR_INVALID;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMalloc(out, (size_t)bytes));
  return JN_OK;
}
jn_status jn_device_free(int32_t device, void* p) { HIP_TRY(hipSetDevice(device)); HIP_TRY(hipFree(p)); return JN_OK; }
jn_status jn_memcpy_h2d(int32_t device, void* dst, const void* src, int64_t bytes) {
  HIP_TRY(hipSetDevice(device)); HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice)); return JN_OK;
}
jn_status jn_memcpy_d2h(int32_t device, void* dst, const void* src, int64_t bytes) {
  HIP_TRY(hipSetDevice(device)); HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost)); return JN_OK;
}
jn_status jn_device_synchronize(int32_t device) { HIP_TRY(hipSetDevice(device)); HIP_TRY(hipDeviceSynchronize()); return JN_OK; }

}  // extern "C"
