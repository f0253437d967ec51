// bm.hip — the block-matching mode (include/jn_bm.h): gfx950 kernels and their C ABI.  Product code.
//
// No reference counterpart (the reference's only matcher is libelas); the definition is in jn_bm.h and its scalar
// restatement (the checker, test infrastructure only) lives outside the product.  Integer arithmetic: the bar is bit-exactness.
//
// Decomposition.  The cost of a (2r+1)^2 block is a vertical running sum of row costs
//   H(x, y, d) = sum_i |a(x+i, y) - b(x+i -/+ d, y)|,     C(x, y, d) = C(x, y-1, d) + H(x, y+r, d) - H(x, y-r-1, d).
// A workgroup owns 64 columns x a band of rows; both prefiltered images are stored with replicated borders wide enough
// that no window ever needs a clamp in x, and the band's rows (+ r above and below, clamped in y) sit in LDS.  A wave's
// lanes are the 64 columns; the wave takes 8 consecutive disparities at a time and walks down the band with the
// last 2r+1 row costs of each of them in registers (the ring index is static: the row loop is unrolled by 2r+1).  Per
// row the lane reads the 16-20 bytes of b that its 8 windows span ONCE and normalises them with v_alignbyte_b32 to start
// at its first window; window k then starts at the compile-time byte offset k (or 7-k) of that span: two or three more
// v_alignbyte_b32 and one v_sad_u8 per dword, the SADs chained through their accumulator operand.  ≈10 vector
// instructions per (pixel, disparity) at 9x9 — the 81 absolute differences and the box filter included.
// The winner per pixel is a packed key (cost << 8 | d) minimised through LDS (ds_min, 4 waves x D/32 rounds); with the
// sub-pixel option a wave evaluates d0-1 and d0+8 too and the key carries the winner's two neighbouring costs in its low
// half (64-bit ds_min; keys differ in the high half, so the low half never decides).  The right-referenced pass is the same
// kernel with the roles of the images swapped and the shift direction reversed; k_bm_finish applies the L/R check and
// the 1/16-pixel formula.
// HBM traffic is ~4 bytes per pixel in and 8-16 out; the kernel is bound by vector issue (W H D x 10 instructions / 64 lanes).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "../../include/jn_bm.h"

namespace {

struct BmDev { int W, H, D, r, cap, lr, subpixel, Wp, padx; };

#define DEV static __device__ __forceinline__

constexpr int kBmMaxBand = 32;        // rows of a band (the launch may choose fewer)
constexpr int kBmPad = 96;            // LDS row of the shifted image: D + kBmPad bytes; image rows padded by D + kBmPad columns each side

// ---- prefilter: g = clamp(Sobel_x, -cap, cap) + cap with replicated borders, rows padded by padx columns on both sides ----
__global__ void __launch_bounds__(256) k_bm_prefilter(BmDev s, const uint8_t* __restrict__ I1, const uint8_t* __restrict__ I2, int pitch,
                                                      long long stride, int n, uint8_t* __restrict__ g) {
  const int xp = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, img = blockIdx.z;
  if (xp >= s.Wp) return;
  const uint8_t* I = img < n ? I1 + (long long)img * stride : I2 + (long long)(img - n) * stride;
  const int x = min(max(xp - s.padx, 0), s.W - 1);
  const int xm = max(x - 1, 0), xq = min(x + 1, s.W - 1), ym = max(y - 1, 0), yq = min(y + 1, s.H - 1);
  const uint8_t* r0 = I + (size_t)ym * pitch; const uint8_t* r1 = I + (size_t)y * pitch; const uint8_t* r2 = I + (size_t)yq * pitch;
  const int sx = ((int)r0[xq] - (int)r0[xm]) + 2 * ((int)r1[xq] - (int)r1[xm]) + ((int)r2[xq] - (int)r2[xm]);
  g[((size_t)img * s.H + y) * s.Wp + xp] = (uint8_t)(min(max(sx, -s.cap), s.cap) + s.cap);
}

DEV uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t shift) { return __builtin_amdgcn_alignbyte(hi, lo, shift); }
DEV uint32_t sad_u8(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }

// ---- matching: one side.  SIDE 0: a = left, b = right at x - d;  SIDE 1: a = right, b = left at x + d. ----
// keys: [n][H][W] of uint32 (cost << 8 | d) or, SUB, uint64 (that << 32 | cost(d-1) << 16 | cost(d+1)).
template <int R, int SIDE, bool SUB>
__global__ void __launch_bounds__(256) k_bm(BmDev s, int n, int band, const uint8_t* __restrict__ g, void* __restrict__ keys_out) {
  constexpr int WB = 2 * R + 1, RING = WB, NDW = (WB + 3) / 4, LASTB = WB - 4 * (NDW - 1);
  constexpr int NK = SUB ? 10 : 8, E = SUB ? 1 : 0, NS = (NK - 1 + WB + 3) / 4;
  constexpr uint32_t kLastMask = LASTB == 4 ? 0xFFFFFFFFu : ((1u << (8 * LASTB)) - 1u);
  constexpr int PA = 80;
  extern __shared__ uint32_t s_mem[];
  const int PB = s.D + kBmPad, rows_tot = band + 2 * R;
  uint8_t* sA = reinterpret_cast<uint8_t*>(s_mem);                        // [rows_tot][PA]
  uint8_t* sB = sA + (size_t)rows_tot * PA;                               // [rows_tot][PB]
  typedef typename std::conditional<SUB, unsigned long long, uint32_t>::type Key;
  Key* sKey = reinterpret_cast<Key*>(sB + (size_t)rows_tot * PB);         // [band][64]   (offset is a multiple of 8: PA, PB are multiples of 16... of 8)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * band, img = blockIdx.z;
  const uint8_t* gA = g + (size_t)(SIDE == 0 ? img : n + img) * s.H * s.Wp;
  const uint8_t* gB = g + (size_t)(SIDE == 0 ? n + img : img) * s.H * s.Wp;
  // stage the rows (aligned dwords: padx, x0, Wp are multiples of 4)
  {
    const int colA = s.padx + x0 - 4, colB = s.padx + (SIDE == 0 ? x0 - s.D - 8 : x0 - 8);
    const int dwA = PA / 4, dwB = PB / 4;
    for (int i = tid; i < rows_tot * dwA; i += 256) {
      const int t = i / dwA, c = i - t * dwA, yy = min(max(y0 - R + t, 0), s.H - 1);
      reinterpret_cast<uint32_t*>(sA)[t * dwA + c] = *reinterpret_cast<const uint32_t*>(gA + (size_t)yy * s.Wp + colA + 4 * c);
    }
    for (int i = tid; i < rows_tot * dwB; i += 256) {
      const int t = i / dwB, c = i - t * dwB, yy = min(max(y0 - R + t, 0), s.H - 1);
      reinterpret_cast<uint32_t*>(sB)[t * dwB + c] = *reinterpret_cast<const uint32_t*>(gB + (size_t)yy * s.Wp + colB + 4 * c);
    }
    for (int i = tid; i < band * 64; i += 256) sKey[i] = ~(Key)0;
  }
  __syncthreads();
  // the reference-side window of this lane: bytes lane + 4 - R ... of the staged row
  const int cA = lane + 4 - R, offA = cA & ~3, shA = cA & 3;
  const int chunks = s.D >> 3;
  for (int chunk = wave; chunk < chunks; chunk += 4) {
    const int d0 = chunk * 8;
    // first byte of the span of b this lane's NK windows cover, in the staged row
    const int c_min = SIDE == 0 ? lane - (d0 + NK - 1 - E) - R + s.D + 8 : lane + d0 - E - R + 8;
    const int offB = c_min & ~3, shB = c_min & 3;
    uint32_t ring[RING][NK];
    uint32_t C[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) {
      C[k] = 0;
#pragma unroll
      for (int q = 0; q < RING; q++) ring[q][k] = 0;
    }
    for (int tb = 0; tb < rows_tot; tb += RING) {
#pragma unroll
      for (int q = 0; q < RING; q++) {
        const int t = tb + q;
        if (t < rows_tot) {
          const uint32_t* pa = reinterpret_cast<const uint32_t*>(sA + t * PA + offA);
          const uint32_t* pb = reinterpret_cast<const uint32_t*>(sB + t * PB + offB);
          uint32_t rawA[NDW + 1], rawB[NS + 1], wa[NDW], span[NS];
#pragma unroll
          for (int j = 0; j <= NDW; j++) rawA[j] = pa[j];
#pragma unroll
          for (int j = 0; j <= NS; j++) rawB[j] = pb[j];
#pragma unroll
          for (int j = 0; j < NDW; j++) wa[j] = alignbyte(rawA[j + 1], rawA[j], shA);
          wa[NDW - 1] &= kLastMask;
#pragma unroll
          for (int j = 0; j < NS; j++) span[j] = alignbyte(rawB[j + 1], rawB[j], shB);
#pragma unroll
          for (int k = 0; k < NK; k++) {
            const int o = SIDE == 0 ? NK - 1 - k : k;             // byte offset of window k inside the span (compile-time after unrolling)
            const int qd = o >> 2, sh = o & 3;
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < NDW; j++) {
              uint32_t w;
              if (j == NDW - 1 && LASTB == 1) {
                const int byte = o + 4 * j;                       // the one byte of the last dword that counts
                w = (span[byte >> 2] >> (8 * (byte & 3))) & 0xFFu;
              } else {
                const uint32_t lo = span[qd + j], hi = qd + j + 1 < NS ? span[qd + j + 1] : 0u;
                w = sh ? alignbyte(hi, lo, (uint32_t)sh) : lo;
                if (j == NDW - 1) w &= kLastMask;
              }
              acc = sad_u8(w, wa[j], acc);
            }
            C[k] = C[k] - ring[q][k] + acc;
            ring[q][k] = acc;
          }
          if (t >= 2 * R) {
            const int ry = t - 2 * R;
            const int dbase = d0 - E;
            uint32_t key = 0xFFFFFFFFu;
#pragma unroll
            for (int k = E; k < E + 8; k++) key = min(key, (C[k] << 8) + (uint32_t)(dbase + k));
            if constexpr (SUB) {
              const int bk = (int)(key & 0xFFu) - dbase;          // 1..8
              uint32_t prev = C[0], next = C[2];
#pragma unroll
              for (int k = 2; k <= 8; k++) { const bool m = bk == k; prev = m ? C[k - 1] : prev; next = m ? C[k + 1] : next; }
              const unsigned long long k64 = ((unsigned long long)key << 32) | (prev << 16) | next;
              atomicMin(&sKey[ry * 64 + lane], k64);
            } else {
              atomicMin(&sKey[ry * 64 + lane], key);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  Key* out = reinterpret_cast<Key*>(keys_out) + (size_t)img * s.H * s.W;
  for (int i = tid; i < band * 64; i += 256) {
    const int ry = i >> 6, l = i & 63, x = x0 + l, y = y0 + ry;
    if (x < s.W && y < s.H) out[(size_t)y * s.W + x] = sKey[i];
  }
}

// ---- L/R check, sub-pixel formula, output ----
template <bool SUB>
__global__ void __launch_bounds__(256) k_bm_finish(BmDev s, const void* __restrict__ keysL, const void* __restrict__ keysR, int16_t* __restrict__ disp) {
  typedef typename std::conditional<SUB, unsigned long long, uint32_t>::type Key;
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, img = blockIdx.z;
  if (x >= s.W) return;
  const size_t row = ((size_t)img * s.H + y) * s.W;
  const Key kl = reinterpret_cast<const Key*>(keysL)[row + x];
  const uint32_t hi = SUB ? (uint32_t)((unsigned long long)kl >> 32) : (uint32_t)kl;
  const int d = (int)(hi & 0xFFu);
  bool ok = true;
  if (s.lr >= 0) {
    ok = x - d >= 0;
    if (ok) {
      const Key kr = reinterpret_cast<const Key*>(keysR)[row + x - d];
      const int dr = (int)((SUB ? (uint32_t)((unsigned long long)kr >> 32) : (uint32_t)kr) & 0xFFu);
      ok = abs(d - dr) <= s.lr;
    }
  }
  int out = SUB ? -16 : -1;
  if (ok) {
    out = SUB ? 16 * d : d;
    if (SUB && d > 0 && d < s.D - 1) {
      const uint32_t lo = (uint32_t)(unsigned long long)kl;
      const int cm = (int)(lo >> 16), cp = (int)(lo & 0xFFFFu), c0 = (int)(hi >> 8);
      const int den = max(cm + cp - 2 * c0, 1);
      out = 16 * d + (16 * (cm - cp) + den) / (2 * den);
    }
  }
  disp[row + x] = (int16_t)out;
}

}  // namespace

struct jn_bm {
  jn_bm_params p;
  BmDev dev;
  int W = 0, H = 0, max_batch = 0, device = 0;
  uint8_t* g = nullptr;        // prefiltered rows [2 * max_batch][H][Wp]
  void* keys = nullptr;        // winners [2][max_batch][H][W], 4 or 8 bytes each
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {};
  jn_bm_times times = {};
};

#define BM_TRY(expr)                                                                        \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess) {                                                                \
      fprintf(stderr, "libjn_stereo: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return JN_ERR_NO_DEVICE;                                                              \
    }                                                                                       \
  } while (0)

namespace {

size_t bm_lds_bytes(const BmDev& s, int band, bool sub) {
  const int rows_tot = band + 2 * s.r;
  return (size_t)rows_tot * (80 + s.D + kBmPad) + (size_t)band * 64 * (sub ? 8 : 4);
}

template <int R, int SIDE, bool SUB>
hipError_t launch_bm_one(hipStream_t st, const BmDev& s, int n, int band, const uint8_t* g, void* keys) {
  static bool configured[16] = {};
  int dev = 0; hipGetDevice(&dev);
  if (dev < 16 && !configured[dev]) {             // dynamic LDS stays below the 64 KB default; nothing to raise, kept for symmetry with kernels.hip
    configured[dev] = true;
  }
  const dim3 grid((s.W + 63) / 64, (s.H + band - 1) / band, n);
  hipLaunchKernelGGL((k_bm<R, SIDE, SUB>), grid, dim3(256), bm_lds_bytes(s, band, SUB), st, s, n, band, g, keys);
  return hipGetLastError();
}

template <int SIDE>
hipError_t launch_bm(hipStream_t st, const BmDev& s, int n, int band, const uint8_t* g, void* keys) {
  const bool sub = s.subpixel != 0;
  switch (s.r) {
    case 2: return sub ? launch_bm_one<2, SIDE, true>(st, s, n, band, g, keys) : launch_bm_one<2, SIDE, false>(st, s, n, band, g, keys);
    case 3: return sub ? launch_bm_one<3, SIDE, true>(st, s, n, band, g, keys) : launch_bm_one<3, SIDE, false>(st, s, n, band, g, keys);
    default: return sub ? launch_bm_one<4, SIDE, true>(st, s, n, band, g, keys) : launch_bm_one<4, SIDE, false>(st, s, n, band, g, keys);
  }
}

}  // namespace

extern "C" {

void jn_bm_params_default(jn_bm_params* p) {
  p->num_disparities = 64; p->block_radius = 4; p->prefilter_cap = 31; p->lr_max_diff = 1; p->subpixel = 0;
}

void jn_bm_destroy(jn_bm* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  hipFree(h->g); hipFree(h->keys);
  for (auto& e : h->ev) if (e) hipEventDestroy(e);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

jn_status jn_bm_create(const jn_bm_params* p, int32_t W, int32_t H, int32_t max_batch, int32_t device, jn_bm** out) {
  if (!p || !out || W < 8 || H < 8 || W > 8192 || H > 8192 || max_batch < 1) return JN_ERR_INVALID;
  *out = nullptr;
  const int D = p->num_disparities;
  if (D < 8 || D > 256 || (D & 7) || p->block_radius < 2 || p->block_radius > 4 || p->prefilter_cap < 1 || p->prefilter_cap > 31)
    return JN_ERR_UNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return JN_ERR_NO_DEVICE;
  BM_TRY(hipSetDevice(device));
  jn_bm* h = new jn_bm();
  h->p = *p; h->W = W; h->H = H; h->max_batch = max_batch; h->device = device;
  BmDev& s = h->dev;
  s.W = W; s.H = H; s.D = D; s.r = p->block_radius; s.cap = p->prefilter_cap; s.lr = p->lr_max_diff; s.subpixel = p->subpixel ? 1 : 0;
  s.padx = D + kBmPad; s.Wp = (W + 2 * s.padx + 3) & ~3;
#define BM_CREATE_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { fprintf(stderr, "libjn_stereo: %s failed: %s\n", #expr, hipGetErrorString(e__)); jn_bm_destroy(h); return JN_ERR_NO_DEVICE; } } while (0)
  BM_CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->g), (size_t)2 * max_batch * H * s.Wp + 64));
  BM_CREATE_TRY(hipMalloc(&h->keys, (size_t)2 * max_batch * H * W * (s.subpixel ? 8 : 4)));
  BM_CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  for (auto& e : h->ev) BM_CREATE_TRY(hipEventCreate(&e));
#undef BM_CREATE_TRY
  *out = h;
  return JN_OK;
}

jn_status jn_bm_process_batch(jn_bm* h, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch, int64_t image_stride, int16_t* dDisp) {
  if (!h || n < 1 || n > h->max_batch || !dI1 || !dI2 || !dDisp || pitch < h->W) return JN_ERR_INVALID;
  BM_TRY(hipSetDevice(h->device));
  const BmDev& s = h->dev;
  hipStream_t st = h->stream;
  BM_TRY(hipEventRecord(h->ev[0], st));
  hipLaunchKernelGGL(k_bm_prefilter, dim3((s.Wp + 255) / 256, s.H, 2 * n), dim3(256), 0, st, s, dI1, dI2, pitch, (long long)image_stride, n, h->g);
  BM_TRY(hipEventRecord(h->ev[1], st));
  // rows per band: 32, shorter while the launch would leave most of the 256 CUs idle (a lone pair)
  int band = kBmMaxBand;
  if (const char* e = getenv("JN_BM_BAND")) band = std::min(std::max(atoi(e), 1), kBmMaxBand);
  else
    while (band > 8 && (long long)((s.W + 63) / 64) * ((s.H + band - 1) / band) * n < 1024) band >>= 1;
  const size_t key_bytes = (size_t)h->max_batch * s.H * s.W * (s.subpixel ? 8 : 4);
  void* keysL = h->keys;
  void* keysR = static_cast<uint8_t*>(h->keys) + key_bytes;
  BM_TRY(launch_bm<0>(st, s, n, band, h->g, keysL));
  if (s.lr >= 0) BM_TRY(launch_bm<1>(st, s, n, band, h->g, keysR));
  BM_TRY(hipEventRecord(h->ev[2], st));
  const dim3 fg((s.W + 255) / 256, s.H, n);
  if (s.subpixel) hipLaunchKernelGGL(k_bm_finish<true>, fg, dim3(256), 0, st, s, keysL, keysR, dDisp);
  else hipLaunchKernelGGL(k_bm_finish<false>, fg, dim3(256), 0, st, s, keysL, keysR, dDisp);
  BM_TRY(hipEventRecord(h->ev[3], st));
  BM_TRY(hipStreamSynchronize(st));
  BM_TRY(hipGetLastError());
  hipEventElapsedTime(&h->times.prefilter, h->ev[0], h->ev[1]);
  hipEventElapsedTime(&h->times.match, h->ev[1], h->ev[2]);
  hipEventElapsedTime(&h->times.finish, h->ev[2], h->ev[3]);
  hipEventElapsedTime(&h->times.total, h->ev[0], h->ev[3]);
  return JN_OK;
}

jn_status jn_bm_last_times(jn_bm* h, jn_bm_times* out) {
  if (!h || !out) return JN_ERR_INVALID;
  *out = h->times;
  return JN_OK;
}

}  // extern "C"
