// sgm.hip — the semi-global-matching mode (include/jn_sgm.h): its C ABI.  Product code.
//
// No reference counterpart (the reference's only matcher is libelas); the definition is in jn_sgm.h and its scalar
// restatement (the checker, test infrastructure only) lives outside the product.  Everything is integer arithmetic, so the bar is bit-exactness.
// The kernels are sgm_sweep.hip's four sweeps (lanes = pixels).  Round 2's one-wave-per-line kernels (lanes = disparities, eight u8 volumes,
// 16 W H D bytes of traffic, 1.3 k pairs/s) lived here behind JN_SGM_IMPL=0 until round 5; their description and numbers: DESIGN_HISTORY.md.
#include <hip/hip_runtime.h>
#include "hooks.h"
#include <stdint.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../include/jn_sgm.h"
#include "sgm_sweep.h"
#include "kernels.h"            // launch_scan: the node's tail on a slot's stream (jn_sgm_submit_scan)

namespace {

__global__ void __launch_bounds__(256) k_sgm_to_u8(const int16_t* __restrict__ d, int subpixel, uint8_t* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int v = d[i];
  if (v < 0) { out[i] = 0; return; }
  if (subpixel) { const int q = v >> 4, r = v & 15; v = q + ((r > 8 || (r == 8 && (q & 1))) ? 1 : 0); }   // half to even
  out[i] = (uint8_t)min(v, 255);
}

}  // namespace

struct jn_sgm {
  jn_sgm_params p;
  int W = 0, H = 0, max_batch = 0, device = 0;
  jnav_sgm::SwDev sw = {};
  jnav_sgm::SweepSizes sizes = {};
  jnav_sgm::SweepBuffers sb = {};
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {};
  hipEvent_t ev_end[8] = {};   // per slot: recorded behind EVERYTHING a submit queued (sweeps + the scan tail); what jn_sgm_wait waits for
  jn_sgm_times times = {};
  // Pipelined form (jn_sgm_submit_scan / jn_sgm_wait): slot 0 is the set above, slots 1 .. kSgmSlots-1 get their own stream, events and
  // buffers the first time they are used.  Batches on different slots overlap on the GPU: the upward sweep's tail (the last blocks of
  // its parallelogram run alone) is filled by the next batch's horizontal and downward sweeps.
  struct Extra { jnav_sgm::SweepBuffers sb = {}; hipStream_t stream = nullptr; hipEvent_t ev[4] = {}; jn_sgm_times times = {}; bool ready = false, shared = false; };
  enum { kSgmSlots = 8 };
  Extra extra[kSgmSlots - 1];
  unsigned long long* scan_scratch[kSgmSlots] = {};   // [max_batch][4] per slot, the scan tail's extrema
  bool pending[kSgmSlots] = {};
  static_assert(kSgmSlots == sizeof(ev_end) / sizeof(ev_end[0]), "one end event per slot");
};

#define SGM_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess) {                                                                \
      fprintf(stderr, "libjn_stereo: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return JN_ERR_NO_DEVICE;                                                              \
    }                                                                                       \
  } while (0)

extern "C" {

void jn_sgm_params_default(jn_sgm_params* p) {
  p->num_disparities = 128; p->P1 = 10; p->P2 = 60; p->prefilter_cap = 31; p->lr_max_diff = 1; p->subpixel = 0;
}

void jn_sgm_destroy(jn_sgm* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  for (auto& x : h->extra) {
    if (x.stream && !x.shared) hipStreamSynchronize(x.stream);
    jnav_sgm::sweep_release(x.sb);
    hipFree(x.sb.gm); hipFree(x.sb.volF); hipFree(x.sb.volH0); hipFree(x.sb.volH1); hipFree(x.sb.gx); hipFree(x.sb.flags); hipFree(x.sb.minr); hipFree(x.sb.dl);
    for (auto& e : x.ev) if (e) hipEventDestroy(e);
    if (x.stream && !x.shared) hipStreamDestroy(x.stream);
  }
  for (auto& q : h->scan_scratch) hipFree(q);
  jnav_sgm::sweep_release(h->sb);
  hipFree(h->sb.gm); hipFree(h->sb.volF); hipFree(h->sb.volH0); hipFree(h->sb.volH1); hipFree(h->sb.gx); hipFree(h->sb.flags); hipFree(h->sb.minr); hipFree(h->sb.dl);
  for (auto& e : h->ev) if (e) hipEventDestroy(e);
  for (auto& e : h->ev_end) if (e) hipEventDestroy(e);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

jn_status jn_sgm_create(const jn_sgm_params* p, int32_t W, int32_t H, int32_t max_batch, int32_t device, jn_sgm** out) {
  if (!p || !out || W < 8 || H < 8 || W > 8192 || H > 8192 || max_batch < 1) return JN_ERR_INVALID;
  *out = nullptr;
  const int D = p->num_disparities;
  if ((D != 64 && D != 128 && D != 256) || p->prefilter_cap < 1 || p->prefilter_cap > 31 || p->P1 < 0 || p->P2 < p->P1 ||
      6 * p->prefilter_cap + p->P2 > 255)
    return JN_ERR_UNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return JN_ERR_NO_DEVICE;
  SGM_TRY(hipSetDevice(device));
  jn_sgm* h = new jn_sgm();
  h->p = *p; h->W = W; h->H = H; h->max_batch = max_batch; h->device = device;
#define SGM_CREATE_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { fprintf(stderr, "libjn_stereo: %s failed: %s\n", #expr, hipGetErrorString(e__)); jn_sgm_destroy(h); return JN_ERR_NO_DEVICE; } } while (0)
  {
    jnav_sgm::SweepSizes& z = h->sizes;
    jnav_sgm::sweep_geometry(W, H, D, p->P1, p->P2, p->prefilter_cap, p->lr_max_diff, p->subpixel, &h->sw, &z, max_batch);
    SGM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->sb.gm), z.gm));
    SGM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->sb.volF), z.vol * (h->sw.wide ? 2 : 1)));
    SGM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->sb.volH0), z.vol));
    SGM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->sb.volH1), z.vol));
    SGM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->sb.gx), z.gx));
    SGM_CREATE_TRY(hipMemset(h->sb.gx, 0, z.gx));               // tag 0 = "never written" (sgm_sweep.hip, k_sw_w)
    h->sb.gx_bytes = z.gx;
    if (const char* e = JN_HOOK_ENV("JN_SGM_EPOCH_START")) h->sb.epoch = (uint32_t)atoi(e) & 0xFFFFu;   // test hook: start next to the tag's wrap-around
    SGM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->sb.flags), z.flags));
    SGM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->sb.minr), z.minr));
    SGM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->sb.dl), z.dl));
  }
  SGM_CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  for (auto& e : h->ev) SGM_CREATE_TRY(hipEventCreate(&e));
#undef SGM_CREATE_TRY
  *out = h;
  return JN_OK;
}

jn_status jn_sgm_process_batch(jn_sgm* h, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch, int64_t image_stride, int16_t* dDisp) {
  if (!h || n < 1 || n > h->max_batch || !dI1 || !dI2 || !dDisp || pitch < h->W) return JN_ERR_INVALID;
  if (h->pending[0]) return JN_ERR_INVALID;                     // slot 0's buffers carry a submitted batch: jn_sgm_wait(h, 0) first
  SGM_TRY(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  SGM_TRY(jnav_sgm::sweep_run(h->sw, n, dI1, dI2, pitch, (long long)image_stride, dDisp, st, h->sb, h->ev, true));
  SGM_TRY(hipStreamSynchronize(st));
  SGM_TRY(hipGetLastError());
  hipEventElapsedTime(&h->times.prefilter, h->ev[0], h->ev[1]);
  hipEventElapsedTime(&h->times.paths, h->ev[1], h->ev[2]);
  hipEventElapsedTime(&h->times.wta, h->ev[2], h->ev[3]);
  hipEventElapsedTime(&h->times.total, h->ev[0], h->ev[3]);
  return JN_OK;
}

// A slot beyond the first: its own buffers, stream and events, allocated when it is first used.
static jn_status sgm_ensure_slot(jn_sgm* h, int slot) {
  if (slot == 0) return JN_OK;
  jn_sgm::Extra& x = h->extra[slot - 1];
  if (x.ready) return JN_OK;
  const jnav_sgm::SweepSizes& z = h->sizes;
  SGM_TRY(hipMalloc(reinterpret_cast<void**>(&x.sb.gm), z.gm));
  SGM_TRY(hipMalloc(reinterpret_cast<void**>(&x.sb.volF), z.vol * (h->sw.wide ? 2 : 1)));
  SGM_TRY(hipMalloc(reinterpret_cast<void**>(&x.sb.volH0), z.vol));
  SGM_TRY(hipMalloc(reinterpret_cast<void**>(&x.sb.volH1), z.vol));
  SGM_TRY(hipMalloc(reinterpret_cast<void**>(&x.sb.gx), z.gx));
  SGM_TRY(hipMemset(x.sb.gx, 0, z.gx));
  x.sb.gx_bytes = z.gx;
  x.sb.epoch = h->sb.epoch;                                     // (tests start it next to the tag's wrap-around)
  SGM_TRY(hipMalloc(reinterpret_cast<void**>(&x.sb.flags), z.flags));
  SGM_TRY(hipMalloc(reinterpret_cast<void**>(&x.sb.minr), z.minr));
  SGM_TRY(hipMalloc(reinterpret_cast<void**>(&x.sb.dl), z.dl));
  {
    static const int share = getenv("JN_SGM_STREAMS") ? atoi(getenv("JN_SGM_STREAMS")) : 0;     // experiment: slot s queues on the stream of slot s % share
    if (share > 0 && slot >= share) {
      const int lower = slot % share;
      const jn_status el = sgm_ensure_slot(h, lower);
      if (el != JN_OK) return el;
      x.stream = lower == 0 ? h->stream : h->extra[lower - 1].stream; x.shared = true;
    } else SGM_TRY(hipStreamCreateWithFlags(&x.stream, hipStreamNonBlocking));
  }
  for (auto& e : x.ev) SGM_TRY(hipEventCreate(&e));
  x.ready = true;
  return JN_OK;
}

jn_status jn_sgm_submit_scan(jn_sgm* h, int32_t slot, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch, int64_t image_stride, int16_t* dDisp,
                             const jn_scan_params* sp, const uint8_t* dLut, uint8_t* dDispU8, double* dBins, double* dMeta) {
  if (!h || slot < 0 || slot >= jn_sgm::kSgmSlots || n < 1 || n > h->max_batch || !dI1 || !dI2 || !dDisp || pitch < h->W) return JN_ERR_INVALID;
  if (sp && (!dLut || !dDispU8 || !dBins || !dMeta || sp->bins < 1 || sp->bins > 1024)) return JN_ERR_INVALID;
  if (h->pending[slot]) return JN_ERR_INVALID;                  // one batch per slot: jn_sgm_wait first
  SGM_TRY(hipSetDevice(h->device));
  const jn_status es = sgm_ensure_slot(h, slot);
  if (es != JN_OK) return es;
  if (sp && !h->scan_scratch[slot]) SGM_TRY(hipMalloc(reinterpret_cast<void**>(&h->scan_scratch[slot]), sizeof(unsigned long long) * 4 * h->max_batch));
  jnav_sgm::SweepBuffers& sb = slot == 0 ? h->sb : h->extra[slot - 1].sb;
  hipStream_t st = slot == 0 ? h->stream : h->extra[slot - 1].stream;
  hipEvent_t* ev = slot == 0 ? h->ev : h->extra[slot - 1].ev;
  // With a scan: ONE tail kernel applies the L/R check, writes the int16 map and the mono8 map (point_cloud.cpp:422 semantics) and scans
  // (JN_SGM_TAIL=3: the three kernels k_sw_lr, k_sgm_to_u8, k_scan one after the other, for A/B).
  static const bool fused_tail = !(getenv("JN_SGM_TAIL") && atoi(getenv("JN_SGM_TAIL")) == 3);
  const bool fuse = sp && fused_tail;
  SGM_TRY(jnav_sgm::sweep_run(h->sw, n, dI1, dI2, pitch, (long long)image_stride, dDisp, st, sb, ev, false, !fuse));
  if (fuse) {
    jnav::SgmWinners w;
    w.dl = sb.dl; w.minr = sb.minr; w.disp = dDisp; w.lr = h->sw.lr; w.subpixel = h->sw.subpixel;
    jnav::launch_scan(st, *sp, n, nullptr, dDispU8, dLut, h->W, h->H, dBins, dMeta, h->scan_scratch[slot], nullptr, &w);
  } else if (sp) {
    const long long px = (long long)n * h->W * h->H;
    hipLaunchKernelGGL(k_sgm_to_u8, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, st, dDisp, h->p.subpixel ? 1 : 0, dDispU8, px);
    jnav::launch_scan(st, *sp, n, nullptr, dDispU8, dLut, h->W, h->H, dBins, dMeta, h->scan_scratch[slot]);
  }
  // the batch's end: behind the scan tail, not behind the sweeps (ev[3] stays the end of the winner-takes-all timing)
  if (!h->ev_end[slot]) SGM_TRY(hipEventCreateWithFlags(&h->ev_end[slot], hipEventDisableTiming));
  SGM_TRY(hipEventRecord(h->ev_end[slot], st));
  SGM_TRY(hipGetLastError());
  h->pending[slot] = true;
  return JN_OK;
}

jn_status jn_sgm_wait(jn_sgm* h, int32_t slot) {
  if (!h || slot < 0 || slot >= jn_sgm::kSgmSlots) return JN_ERR_INVALID;
  if (!h->pending[slot]) return JN_OK;
  SGM_TRY(hipSetDevice(h->device));
  hipEvent_t* ev = slot == 0 ? h->ev : h->extra[slot - 1].ev;
  jn_sgm_times& t = slot == 0 ? h->times : h->extra[slot - 1].times;
  h->pending[slot] = false;
  SGM_TRY(hipEventSynchronize(h->ev_end[slot]));                // the slot's own end, scan tail included (its stream may carry a later slot's batch)
  SGM_TRY(hipGetLastError());
  hipEventElapsedTime(&t.prefilter, ev[0], ev[1]);
  hipEventElapsedTime(&t.paths, ev[1], ev[2]);
  hipEventElapsedTime(&t.wta, ev[2], ev[3]);
  hipEventElapsedTime(&t.total, ev[0], ev[3]);
  if (slot != 0) h->times = t;                                  // jn_sgm_last_times: the batch waited for last
  return JN_OK;
}

const void* jn_sgm_debug_ptr(jn_sgm* h, int32_t which, int32_t info[5]) {
  if (!h) return nullptr;
  if (info) { info[0] = h->sw.wide; info[1] = h->sw.Wp; info[2] = h->sw.padl; info[3] = h->sw.NB; info[4] = 1; }
  switch (which) {
    case 0: return h->sb.volF; case 1: return h->sb.volH0; case 2: return h->sb.volH1;
    case 3: return h->sb.minr; case 4: return h->sb.dl; case 5: return h->sb.gm;
  }
  return nullptr;
}

jn_status jn_sgm_last_times(jn_sgm* h, jn_sgm_times* out) {
  if (!h || !out) return JN_ERR_INVALID;
  *out = h->times;
  return JN_OK;
}

jn_status jn_sgm_disparity_to_u8(int32_t device, const int16_t* dDisp, int32_t subpixel, uint8_t* dOut, int64_t n) {
  if (!dDisp || !dOut || n < 0) return JN_ERR_INVALID;
  SGM_TRY(hipSetDevice(device));
  if (n) hipLaunchKernelGGL(k_sgm_to_u8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, dDisp, subpixel ? 1 : 0, dOut, (long long)n);
  SGM_TRY(hipStreamSynchronize(nullptr));
  SGM_TRY(hipGetLastError());
  return JN_OK;
}

}  // extern "C"
