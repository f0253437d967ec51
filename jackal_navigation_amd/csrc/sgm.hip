// sgm.hip — the semi-global-matching mode (include/jn_sgm.h): gfx950 kernels and their C ABI.  Product code.
//
// No reference counterpart (the reference's only matcher is libelas); the definition is in jn_sgm.h and its scalar
// restatement (the checker, test infrastructure only) lives outside the product.  Everything is integer arithmetic, so the bar is bit-exactness.
//
// Decomposition.  An SGM path L_r(p, .) depends only on the previous pixel of ITS line, so the lines of one direction
// are independent 1-D recurrences: one wave64 per line, lanes = disparities (D/64 per lane), walking the line pixel by
// pixel.  Per pixel a wave
//   * forms the 1x3 SAD cost of its disparities from one unaligned dword of the prefiltered right row (rows are stored
//     with replicated borders, so there are no clamps) and one wave-uniform dword of the left row: one v_sad_u8 per d;
//   * gets L(p-r, d-1) / L(p-r, d+1) of the neighbouring lanes with two DPP wave shifts, the minimum over d with a
//     DPP prefix-min (row_shr 1,2,4,8 + row_bcast 15,31), and writes its D bytes of L_r as one contiguous store.
// The eight directions run as ONE launch (the wave index selects direction and line).  Each direction writes its own
// u8 volume; k_sgm_wta then streams the eight volumes once (16 bytes per lane and volume), sums them to S, takes the
// winner per pixel with a DPP group minimum, the right image's winner with LDS atomic minima on packed keys
// (S << 8 | d), and applies the L/R check and the optional 1/16-pixel refinement.
// HBM traffic: 8 W H D written + 8 W H D read (twice SURVEY 8d's B_sgm lower bound of 4 W H D, which assumes
// that all eight paths of a pixel meet in one sweep — on a GPU the in-row dependency of the horizontal paths and the
// in-column one of the vertical paths cannot both be walked by one decomposition without exchanging the whole state).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../include/jn_sgm.h"
#include "sgm_sweep.h"
#include "kernels.h"            // launch_scan: the node's tail on a slot's stream (jn_sgm_submit_scan)

namespace {

struct SgmDev { int W, H, D, P1, P2, cap, lr, subpixel, Wp, off, dbg; };   // dbg: JN_SGM_DBG profiling hook (bit 0: path kernel without its stores — results are then WRONG)

#define DEV static __device__ __forceinline__

// ---- prefilter: g = clamp(Sobel_x, -cap, cap) + cap with replicated borders, rows padded (off bytes left, >= 8 right) ----
__global__ void __launch_bounds__(256) k_sgm_prefilter(SgmDev s, const uint8_t* __restrict__ I1, const uint8_t* __restrict__ I2, int pitch,
                                                       long long stride, int n, uint8_t* __restrict__ g) {
  const int xp = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, img = blockIdx.z;
  if (xp >= s.Wp) return;
  const uint8_t* I = img < n ? I1 + (long long)img * stride : I2 + (long long)(img - n) * stride;
  const int x = min(max(xp - s.off, 0), s.W - 1);
  const int xm = max(x - 1, 0), xq = min(x + 1, s.W - 1), ym = max(y - 1, 0), yq = min(y + 1, s.H - 1);
  const uint8_t* r0 = I + (size_t)ym * pitch; const uint8_t* r1 = I + (size_t)y * pitch; const uint8_t* r2 = I + (size_t)yq * pitch;
  const int sx = ((int)r0[xq] - (int)r0[xm]) + 2 * ((int)r1[xq] - (int)r1[xm]) + ((int)r2[xq] - (int)r2[xm]);
  g[((size_t)img * s.H + y) * s.Wp + xp] = (uint8_t)(min(max(sx, -s.cap), s.cap) + s.cap);
}

DEV uint32_t load_u32_unaligned(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// wave-wide minimum of v (all 64 lanes active), result wave-uniform
DEV unsigned wave_min(unsigned v) {
  const unsigned inf = 0xFFFFFFFFu;
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(inf, v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(inf, v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(inf, v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(inf, v, 0x118, 0xf, 0xf, false));   // row_shr:8  -> lane 15 of each row = row minimum
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(inf, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1 and 3
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(inf, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2 and 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- the eight path directions, one wave per line ----
// Lines of direction (dx, dy) start at the pixels whose predecessor lies outside the image; they are numbered: the
// H rows, the H rows (reverse), the W columns, the W columns (reverse), then per diagonal direction W starts on the
// top / bottom row followed by H - 1 starts on the left / right column.
template <int DPL>
__global__ void __launch_bounds__(256) k_sgm_path(SgmDev s, int n, const uint8_t* __restrict__ g, uint8_t* __restrict__ Lr) {
  const int W = s.W, H = s.H, D = s.D;
  const int lane = threadIdx.x & 63;
  int id = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int frame = blockIdx.y;
  const int ndiag = W + H - 1;
  int dir, x0, y0, dx, dy;
  if (id < H) { dir = 0; dx = 1; dy = 0; x0 = 0; y0 = id; }
  else if ((id -= H) < H) { dir = 1; dx = -1; dy = 0; x0 = W - 1; y0 = id; }
  else if ((id -= H) < W) { dir = 2; dx = 0; dy = 1; x0 = id; y0 = 0; }
  else if ((id -= W) < W) { dir = 3; dx = 0; dy = -1; x0 = id; y0 = H - 1; }
  else {
    id -= W;
    const int q = id / ndiag, i = id - q * ndiag;
    if (q > 3) return;
    dir = 4 + q;
    dx = (q == 0 || q == 3) ? 1 : -1;                        // (1,1), (-1,-1), (-1,1), (1,-1)
    dy = (q == 0 || q == 2) ? 1 : -1;
    if (i < W) { x0 = i; y0 = dy > 0 ? 0 : H - 1; }
    else { x0 = dx > 0 ? 0 : W - 1; y0 = dy > 0 ? 1 + (i - W) : H - 2 - (i - W); }
  }
  int len = 1 << 30;
  if (dx > 0) len = min(len, W - x0); else if (dx < 0) len = min(len, x0 + 1);
  if (dy > 0) len = min(len, H - y0); else if (dy < 0) len = min(len, y0 + 1);

  const size_t gimg = (size_t)H * s.Wp;
  const uint8_t* gl = g + (size_t)frame * gimg + (size_t)y0 * s.Wp + s.off + x0 - 1;                      // left window x-1 .. x+1 (wave-uniform)
  // right bytes of this lane's disparities: x - DPL*lane - DPL ...; kept as a wave-uniform pointer (at lane 63's bytes) plus
  // a non-negative per-lane offset, so that the pointer walks on the scalar unit and the loads use base + offset addressing
  const uint8_t* gr = g + (size_t)(n + frame) * gimg + (size_t)y0 * s.Wp + s.off + x0 - DPL * 63 - DPL;
  const unsigned roff = (unsigned)(DPL * (63 - lane));
  const long long gstep = (long long)dy * s.Wp + dx;
  uint8_t* out = Lr + (((size_t)dir * n + frame) * H * W + (size_t)y0 * W + x0) * D;
  const unsigned ooff = (unsigned)(DPL * lane);
  const long long ostep = ((long long)dy * W + dx) * D;

  const unsigned big = 0xFFFFu;
  unsigned Lp[DPL];
#pragma unroll
  for (int j = 0; j < DPL; j++) Lp[j] = 0u;                  // with L = 0 and min = 0 the recurrence yields L = C at the first pixel
  unsigned min_prev = 0u;
  const unsigned P1 = (unsigned)s.P1, P2 = (unsigned)s.P2;
  // The neighbours across lanes arrive by DPP wave shifts into these two registers; lane 0 of `up` and lane 63 of `dn`
  // receive nothing and keep the "does not exist" value they start with, so the registers are never re-initialised.
  unsigned up = big, dn = big;

  // One pixel of the line: costs from the loaded bytes (the SAD's accumulator operand carries -min, so C - min costs nothing),
  // the recurrence, the minimum over d, the store.
  auto pixel = [&](uint32_t a_raw, uint32_t b0, uint32_t b1) {
    const uint32_t a = a_raw & 0x00FFFFFFu;
    const unsigned nm = 0u - min_prev;
    // 1x3 SAD of the prefiltered rows: disparity DPL*lane + j matches right bytes [DPL-1-j, DPL+1-j] of the loaded run
    unsigned C[DPL];
    if (DPL == 1) C[0] = __builtin_amdgcn_sad_u8(a, b0 & 0x00FFFFFFu, nm);
    if (DPL == 2) { C[0] = __builtin_amdgcn_sad_u8(a, b0 >> 8, nm); C[1] = __builtin_amdgcn_sad_u8(a, b0 & 0x00FFFFFFu, nm); }
    if (DPL == 4) {
      C[3] = __builtin_amdgcn_sad_u8(a, b0 & 0x00FFFFFFu, nm);
      C[2] = __builtin_amdgcn_sad_u8(a, b0 >> 8, nm);
      C[1] = __builtin_amdgcn_sad_u8(a, __builtin_amdgcn_alignbit(b1, b0, 16) & 0x00FFFFFFu, nm);
      C[0] = __builtin_amdgcn_sad_u8(a, __builtin_amdgcn_alignbit(b1, b0, 24) & 0x00FFFFFFu, nm);
    }
    // neighbours in d across lanes: d-1 of this lane's first disparity, d+1 of its last
    up = (unsigned)__builtin_amdgcn_update_dpp((int)up, (int)Lp[DPL - 1], 0x138, 0xf, 0xf, false);   // wave_shr:1
    dn = (unsigned)__builtin_amdgcn_update_dpp((int)dn, (int)Lp[0], 0x130, 0xf, 0xf, false);         // wave_shl:1
    const unsigned far = min_prev + P2;
    unsigned Ln[DPL];
    unsigned mn = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < DPL; j++) {
      const unsigned lo = j == 0 ? up : Lp[j - 1], hi = j == DPL - 1 ? dn : Lp[j + 1];
      const unsigned m = min(min(Lp[j], min(lo, hi) + P1), far);
      Ln[j] = C[j] + m;                                      // C already holds cost - min_prev
      mn = min(mn, Ln[j]);
    }
    min_prev = wave_min(mn);
    if (s.dbg & 1) { if (Ln[0] == 0xFFFFFFF0u) out[ooff] = 1; }            // profiling: keep the arithmetic alive, never store
    else if (DPL == 1) out[ooff] = (uint8_t)Ln[0];
    else if (DPL == 2) { const uint16_t v = (uint16_t)(Ln[0] | (Ln[1] << 8)); __builtin_memcpy(out + ooff, &v, 2); }
    else if (DPL == 4) { const uint32_t v = Ln[0] | (Ln[1] << 8) | (Ln[2] << 16) | (Ln[3] << 24); __builtin_memcpy(out + ooff, &v, 4); }
    out += ostep;
#pragma unroll
    for (int j = 0; j < DPL; j++) Lp[j] = Ln[j];
  };
  auto load = [&](uint32_t& a, uint32_t& b0, uint32_t& b1) {
    a = load_u32_unaligned(gl); b0 = load_u32_unaligned(gr + roff); b1 = DPL == 4 ? load_u32_unaligned(gr + roff + 4) : 0u;
    gl += gstep; gr += gstep;
  };

  // loads run one pixel ahead of the arithmetic; two pixels per turn so that the two sets of load registers swap roles
  // instead of being copied
  uint32_t a0 = 0, b00 = 0, b01 = 0, a1 = 0, b10 = 0, b11 = 0;
  load(a0, b00, b01);
  int i = 0;
  for (; i + 2 <= len; i += 2) {
    load(a1, b10, b11);                                      // pixel i + 1 exists
    pixel(a0, b00, b01);
    if (i + 2 < len) load(a0, b00, b01);
    pixel(a1, b10, b11);
  }
  if (i < len) pixel(a0, b00, b01);
}

// ---- sum of the eight volumes, winner-takes-all for both images, L/R check, sub-pixel: one workgroup per image row ----
// LPP lanes share a pixel (16 disparities each); a wave takes 64 / LPP pixels per step.
template <int LPP>
DEV unsigned group_min(unsigned v) {
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));                 // quad_perm [1,0,3,2]
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));                 // quad_perm [2,3,0,1]
  if (LPP >= 8) v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false));  // row_half_mirror
  if (LPP >= 16) v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false)); // row_mirror
  return v;
}
template <int LPP>
__global__ void __launch_bounds__(256) k_sgm_wta(SgmDev s, int n, const uint8_t* __restrict__ Lr, int16_t* __restrict__ disp) {
  extern __shared__ uint32_t lds[];                          // [W] right-image keys | [W] left winners | [4 waves][64 lanes][16] S scratch (sub-pixel)
  const int W = s.W, H = s.H, D = s.D;
  uint32_t* s_minR = lds;
  uint32_t* s_dl = lds + W;
  uint16_t* s_S = reinterpret_cast<uint16_t*>(lds + 2 * W);
  const int y = blockIdx.x, frame = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int x = tid; x < W; x += 256) s_minR[x] = 0xFFFFFFFFu;
  __syncthreads();
  constexpr int PXW = 64 / LPP;
  const int sub = lane % LPP, dbase = sub * 16;
  const size_t vol = (size_t)n * H * W * D;                  // one direction's volume
  const uint8_t* row = Lr + (((size_t)frame * H + y) * W) * D + dbase;
  uint16_t* myS = s_S + (size_t)(wave * 64 + lane) * 16;
  for (int xb = wave * PXW; xb < W; xb += 4 * PXW) {
    const int x = xb + lane / LPP;
    const bool in = x < W;
    uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};     // sums of bytes 0,2 / 1,3 of each dword as 16-bit halves
    if (in) {
      uint4 v[8];
#pragma unroll
      for (int r = 0; r < 8; r++) v[r] = *reinterpret_cast<const uint4*>(row + (size_t)r * vol + (size_t)x * D);
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const uint32_t w[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
        for (int k = 0; k < 4; k++) { lo[k] += w[k] & 0x00FF00FFu; hi[k] += (w[k] >> 8) & 0x00FF00FFu; }
      }
    }
    unsigned S[16];
#pragma unroll
    for (int k = 0; k < 4; k++) { S[4 * k] = lo[k] & 0xFFFFu; S[4 * k + 1] = hi[k] & 0xFFFFu; S[4 * k + 2] = lo[k] >> 16; S[4 * k + 3] = hi[k] >> 16; }
    // left image: smallest key (S << 8 | d) of the pixel; right image: the same keys, minimised per column x - d
    unsigned best = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const unsigned key = (S[j] << 8) | (unsigned)(dbase + j);
      best = min(best, key);
      const int xr = x - dbase - j;
      if (in && xr >= 0) atomicMin(&s_minR[xr], key);
    }
    best = group_min<LPP>(in ? best : 0xFFFFFFFFu);
    if (s.subpixel) {
#pragma unroll
      for (int j = 0; j < 16; j++) myS[j] = (uint16_t)S[j];   // the pixel's S[0..D) = 16 consecutive values per lane, LPP lanes in a row
    }
    if (in && sub == 0) {
      const int d = (int)(best & 255u);
      int d16 = 16 * d;
      if (s.subpixel && d > 0 && d < D - 1) {
        const uint16_t* ps = s_S + (size_t)(wave * 64 + lane) * 16;     // lane = first lane of the pixel's group
        const int sm = ps[d - 1], sc = ps[d], sp = ps[d + 1];
        const int den = max(sm + sp - 2 * sc, 1);
        d16 = 16 * d + (16 * (sm - sp) + den) / (2 * den);
      }
      s_dl[x] = (uint32_t)d | ((uint32_t)(uint16_t)d16 << 16);
    }
  }
  __syncthreads();
  int16_t* out = disp + ((size_t)frame * H + y) * W;
  const int scale = s.subpixel ? 16 : 1;
  for (int x = tid; x < W; x += 256) {
    const uint32_t e = s_dl[x];
    const int d = (int)(e & 0xFFFFu);
    bool ok = true;
    if (s.lr >= 0) ok = x - d >= 0 && abs(d - (int)(s_minR[max(x - d, 0)] & 255u)) <= s.lr;
    out[x] = (int16_t)(ok ? (s.subpixel ? (int)(int16_t)(e >> 16) : d) : -scale);
  }
}

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
  SgmDev dev;
  int W = 0, H = 0, max_batch = 0, device = 0;
  int impl = 1;                // 1: the sweep kernels of sgm_sweep.hip (default); 0: the round-2 one-wave-per-line kernels below (JN_SGM_IMPL=0, kept for A/B)
  uint8_t* g = nullptr;        // impl 0: prefiltered rows [2 * max_batch][H][Wp]
  uint8_t* Lr = nullptr;       // impl 0: path volumes [8][max_batch][H][W][D]
  jnav_sgm::SwDev sw = {};     // impl 1
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
  hipFree(h->g); hipFree(h->Lr);
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
  SgmDev& s = h->dev;
  s.W = W; s.H = H; s.D = D; s.P1 = p->P1; s.P2 = p->P2; s.cap = p->prefilter_cap; s.lr = p->lr_max_diff; s.subpixel = p->subpixel ? 1 : 0;
  s.off = D + 8; s.Wp = s.off + W + 8;
  s.dbg = getenv("JN_SGM_DBG") ? atoi(getenv("JN_SGM_DBG")) : 0;
#define SGM_CREATE_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { fprintf(stderr, "libjn_stereo: %s failed: %s\n", #expr, hipGetErrorString(e__)); jn_sgm_destroy(h); return JN_ERR_NO_DEVICE; } } while (0)
  h->impl = getenv("JN_SGM_IMPL") ? atoi(getenv("JN_SGM_IMPL")) : 1;
  if (h->impl == 0 && W > 7168) { jn_sgm_destroy(h); return JN_ERR_UNSUPPORTED; }   // k_sgm_wta's 2 W dwords + 8 KB of LDS would pass the 64 KB a launch gets by default
  if (h->impl == 0) {
    SGM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->g), (size_t)2 * max_batch * H * s.Wp + 64));
    SGM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->Lr), (size_t)8 * max_batch * H * W * D));
  } else {
    jnav_sgm::SweepSizes& z = h->sizes;
    jnav_sgm::sweep_geometry(W, H, D, p->P1, p->P2, p->prefilter_cap, p->lr_max_diff, p->subpixel, &h->sw, &z, max_batch);
    SGM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->sb.gm), z.gm));
    SGM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->sb.volF), z.vol * (h->sw.wide ? 2 : 1)));
    SGM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->sb.volH0), z.vol));
    SGM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->sb.volH1), z.vol));
    SGM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->sb.gx), z.gx));
    SGM_CREATE_TRY(hipMemset(h->sb.gx, 0, z.gx));               // tag 0 = "never written" (sgm_sweep.hip, k_sw_w)
    h->sb.gx_bytes = z.gx;
    if (const char* e = getenv("JN_SGM_EPOCH_START")) h->sb.epoch = (uint32_t)atoi(e) & 0xFFFFu;   // test hook: start next to the tag's wrap-around
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
  const SgmDev& s = h->dev;
  hipStream_t st = h->stream;
  if (h->impl != 0) {
    SGM_TRY(jnav_sgm::sweep_run(h->sw, n, dI1, dI2, pitch, (long long)image_stride, dDisp, st, h->sb, h->ev, true));
    SGM_TRY(hipStreamSynchronize(st));
    SGM_TRY(hipGetLastError());
    hipEventElapsedTime(&h->times.prefilter, h->ev[0], h->ev[1]);
    hipEventElapsedTime(&h->times.paths, h->ev[1], h->ev[2]);
    hipEventElapsedTime(&h->times.wta, h->ev[2], h->ev[3]);
    hipEventElapsedTime(&h->times.total, h->ev[0], h->ev[3]);
    return JN_OK;
  }
  SGM_TRY(hipEventRecord(h->ev[0], st));
  hipLaunchKernelGGL(k_sgm_prefilter, dim3((s.Wp + 255) / 256, s.H, 2 * n), dim3(256), 0, st, s, dI1, dI2, pitch, (long long)image_stride, n, h->g);
  SGM_TRY(hipEventRecord(h->ev[1], st));
  const int lines = 2 * s.H + 2 * s.W + 4 * (s.W + s.H - 1);
  const dim3 pg((lines + 3) / 4, n);
  if (s.D == 64) hipLaunchKernelGGL(k_sgm_path<1>, pg, dim3(256), 0, st, s, n, h->g, h->Lr);
  else if (s.D == 128) hipLaunchKernelGGL(k_sgm_path<2>, pg, dim3(256), 0, st, s, n, h->g, h->Lr);
  else hipLaunchKernelGGL(k_sgm_path<4>, pg, dim3(256), 0, st, s, n, h->g, h->Lr);
  SGM_TRY(hipEventRecord(h->ev[2], st));
  const size_t lds = (size_t)2 * s.W * sizeof(uint32_t) + (size_t)4 * 64 * 16 * sizeof(uint16_t);
  const dim3 wg(s.H, n);
  // the volumes are laid out [8][max_batch][H][W][D]; the kernels index them with the batch actually submitted
  if (s.D == 64) hipLaunchKernelGGL(k_sgm_wta<4>, wg, dim3(256), lds, st, s, n, h->Lr, dDisp);
  else if (s.D == 128) hipLaunchKernelGGL(k_sgm_wta<8>, wg, dim3(256), lds, st, s, n, h->Lr, dDisp);
  else hipLaunchKernelGGL(k_sgm_wta<16>, wg, dim3(256), lds, st, s, n, h->Lr, dDisp);
  SGM_TRY(hipEventRecord(h->ev[3], st));
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
  if (h->impl == 0) return JN_ERR_UNSUPPORTED;                  // the round-2 kernels exist for A/B through jn_sgm_process_batch only
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
  if (info) { info[0] = h->sw.wide; info[1] = h->sw.Wp; info[2] = h->sw.padl; info[3] = h->sw.NB; info[4] = h->impl; }
  if (h->impl == 0) return nullptr;
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
