// bm.hip — the block-matching mode (include/jn_bm.h): gfx950 kernels and their C ABI.  Product code.
//
// No reference counterpart (the reference's only matcher is libelas); the definition is in jn_bm.h and its scalar
// restatement (the checker, test infrastructure only) lives outside the product.  Integer arithmetic: the bar is bit-exactness.
//
// Decomposition.  The cost of a (2r+1)^2 block is a vertical running sum of row costs
//   H(x, y, d) = sum_i |a(x+i, y) - b(x+i -/+ d, y)|,     C(x, y, d) = C(x, y-1, d) + H(x, y+r, d) - H(x, y-r-1, d).
// A workgroup owns 64 columns x a band of rows; both prefiltered images are stored with replicated borders wide enough
// that no window ever needs a clamp in x, and the band's rows (+ r above and below, clamped in y) sit in LDS.  A wave's
// lanes are the 64 columns; the wave takes 16 consecutive disparities at a time and walks down the band with the last
// 2r+1 row costs of each of them in registers, four 16-bit costs per register pair (the ring index is static: the row
// loop is unrolled by 2r+1).  Per row the lane reads the 24 bytes of b that its 16 windows span ONCE and normalises them
// with v_alignbyte_b32 (per-lane shift) to start at its first window.  From there the hardware slides: v_qsad_pk_u16_u8
// takes 8 bytes of b and 4 bytes of a and returns the SADs of the four windows at byte offsets 0..3, accumulated into
// four packed 16-bit sums — the row costs of four consecutive disparities per instruction and dword of the window.  The
// odd last byte(s) of the window go through v_mqsad_pk_u16_u8, which skips the bytes of a that are zero: the prefilter
// stores g + 1, so a zero byte only ever is one this kernel put there.  Running sums and keys stay packed
// (v_pk_sub/add_u16, v_pk_mad_u16, v_pk_min_u16): (cost << 3 | k) of eight disparities fits 16 bits.
// Measured (scripts/probes/qsad_probe.hip): the quad SADs issue at 1/5 of v_sad_u8's rate for 4x the work and no per-window
// alignment.  The winner per pixel is a key (cost << 8 | d) minimised through LDS (4 waves x D/64 rounds).  The
// right-referenced pass is the same kernel with the images swapped and the shift reversed; k_bm_finish applies the L/R
// check; with the sub-pixel option it stages the band again and evaluates the two costs next to each winner directly.
// HBM traffic is ~13 bytes per pixel; the kernel is bound by vector issue.
#include <hip/hip_runtime.h>
#include "hooks.h"
#include <stdint.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../include/jn_bm.h"
#include "kernels.h"          // launch_scan: the node's tail on the same stream (jn_bm_process_scan)
#include "bm_mfma.h"
#include "prefilter.h"          // JN_BM_COST_SSD: the matrix-core kernels

namespace {

struct BmDev { int W, H, D, r, cap, lr, subpixel, Wp, padx; };

#define DEV static __device__ __forceinline__

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

constexpr int kBmMaxBand = 64;        // rows of a band (the launch may choose fewer)
constexpr int kBmPad = 112;           // LDS row of the shifted image: D + kBmPad bytes; image rows padded by D + kBmPad columns each side
constexpr int kBmPA = 80;             // LDS row of the reference-side image: columns x0 - 4 ... x0 + 75

// ---- prefilter: 4 (g + 1), g = clamp(Sobel_x, -cap, cap) + cap, with replicated borders, rows padded by padx columns.  + 1: never zero (see
// above); x 4: every cost is then a multiple of 4 and the two low bits of a packed 16-bit sum are free to carry the position
// of its disparity inside its quad (4 (81 x 62) + 3 < 2^16). ----
__global__ void __launch_bounds__(256) k_bm_prefilter(BmDev s, const uint8_t* __restrict__ I1, const uint8_t* __restrict__ I2, int pitch,
                                                      long long stride, int n, uint8_t* __restrict__ g) {
  const int xp = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y, img = blockIdx.z;      // four padded columns per thread (Wp is a multiple of 4)
  if (xp >= s.Wp) return;
  const uint8_t* I = img < n ? I1 + (long long)img * stride : I2 + (long long)(img - n) * stride;
  const int ym = max(y - 1, 0), yq = min(y + 1, s.H - 1);
  const uint8_t* r0 = I + (size_t)ym * pitch; const uint8_t* r1 = I + (size_t)y * pitch; const uint8_t* r2 = I + (size_t)yq * pitch;
  int v[4];
  const int x0 = xp - s.padx;
  if (x0 >= 0 && x0 + 3 <= s.W - 1) jnav_pre::sobel4<1>(r0, r1, r2, x0, s.W, s.cap, v);
  else {
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = min(max(jnav_pre::sobel_x_clamped(r0, r1, r2, min(max(x0 + k, 0), s.W - 1), s.W), -s.cap), s.cap);
  }
  const uint32_t c1 = (uint32_t)(s.cap + 1);
  *reinterpret_cast<uint32_t*>(g + ((size_t)img * s.H + y) * s.Wp + xp) =
      (4u * ((uint32_t)v[0] + c1)) | ((4u * ((uint32_t)v[1] + c1)) << 8) | ((4u * ((uint32_t)v[2] + c1)) << 16) | ((4u * ((uint32_t)v[3] + c1)) << 24);
}

DEV uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t shift) { return __builtin_amdgcn_alignbyte(hi, lo, shift); }
DEV uint32_t sad_u8(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }
DEV uint64_t pack64(uint32_t hi, uint32_t lo) { return ((uint64_t)hi << 32) | lo; }
DEV us2 as_us2(uint32_t v) { return __builtin_bit_cast(us2, v); }
DEV uint32_t as_u32(us2 v) { return __builtin_bit_cast(uint32_t, v); }

// Both images' rows of a band into LDS (aligned dwords: padx, x0, Wp are multiples of 4).  SIDE 0: b columns start at
// x0 - D - 16 (window of disparity d at lane - d - R + D + 16); SIDE 1: at x0 - 8 (window at lane + d - R + 8).
template <int R, int SIDE>
DEV void bm_stage(const BmDev& s, int n, int img, int x0, int y0, int rows_tot, const uint8_t* __restrict__ g, uint8_t* sA, uint8_t* sB) {
  const int PB = s.D + kBmPad, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint8_t* gA = g + (size_t)(SIDE == 0 ? img : n + img) * s.H * s.Wp;
  const uint8_t* gB = g + (size_t)(SIDE == 0 ? n + img : img) * s.H * s.Wp;
  const int colA = s.padx + x0 - 4, colB = s.padx + (SIDE == 0 ? x0 - s.D - 16 : x0 - 8);
  const int dwA = kBmPA / 4, dwB = PB / 4;                 // 20 and <= 92 dwords per row
  // one wave per row: lanes take the row's dwords of b (one or two rounds), the first 20 lanes those of a
  for (int t = wave; t < rows_tot; t += 4) {
    const int yy = min(max(y0 - R + t, 0), s.H - 1);
    const uint32_t* srcB = reinterpret_cast<const uint32_t*>(gB + (size_t)yy * s.Wp + colB);
    const uint32_t* srcA = reinterpret_cast<const uint32_t*>(gA + (size_t)yy * s.Wp + colA);
    uint32_t* dstB = reinterpret_cast<uint32_t*>(sB + (size_t)t * PB);
    uint32_t* dstA = reinterpret_cast<uint32_t*>(sA + (size_t)t * kBmPA);
    const uint32_t b0 = lane < dwB ? srcB[lane] : 0u;
    const uint32_t b1 = lane + 64 < dwB ? srcB[lane + 64] : 0u;
    const uint32_t a0 = lane < dwA ? srcA[lane] : 0u;
    if (lane < dwB) dstB[lane] = b0;
    if (lane + 64 < dwB) dstB[lane + 64] = b1;
    if (lane < dwA) dstA[lane] = a0;
  }
}

// ---- matching: one side.  SIDE 0: a = left, b = right at x - d;  SIDE 1: a = right, b = left at x + d. ----
// keys: [n][H][W] uint32 (cost << 8 | d).
template <int R, int SIDE>
__global__ void __launch_bounds__(256) k_bm(BmDev s, int n, int band, const uint8_t* __restrict__ g, uint32_t* __restrict__ keys_out) {
  constexpr int WB = 2 * R + 1, RING = WB, NDW = (WB + 3) / 4, LASTB = WB - 4 * (NDW - 1);
  constexpr int CH = 16, NQ = CH / 4, NS = (CH - 1 + WB + 3) / 4;
  constexpr uint32_t kLastMask = LASTB == 4 ? 0xFFFFFFFFu : ((1u << (8 * LASTB)) - 1u);
  constexpr int PA = kBmPA;
  extern __shared__ uint32_t s_mem[];
  // rows staged: the band + 2R, rounded up to whole turns of the ring so that the unrolled row loop needs no "row exists"
  // test (the extra rows are real image rows, clamped; what they produce is never written)
  const int PB = s.D + kBmPad, rows_tot = (band + 2 * R + RING - 1) / RING * RING;
  uint8_t* sA = reinterpret_cast<uint8_t*>(s_mem);                        // [rows_tot][PA]
  uint8_t* sB = sA + (size_t)rows_tot * PA;                               // [rows_tot][PB]
  uint32_t* sKey = reinterpret_cast<uint32_t*>(sB + (size_t)rows_tot * PB);   // [band][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * band, img = blockIdx.z;
  bm_stage<R, SIDE>(s, n, img, x0, y0, rows_tot, g, sA, sB);
  for (int i = tid; i < band * 64; i += 256) sKey[i] = 0xFFFFFFFFu;
  __syncthreads();
  // the reference-side window of this lane: bytes lane + 4 - R ... of the staged row
  const int cA = lane + 4 - R, offA = cA & ~3, shA = cA & 3;
  const int chunks = (s.D + CH - 1) / CH;
  for (int chunk = wave; chunk < chunks; chunk += 4) {
    const int d0 = chunk * CH;
    const bool second = d0 + 8 < s.D;                       // D is a multiple of 8: the last chunk may hold 8 disparities only
    // first byte of the span of b this lane's CH windows cover, in the staged row; window of d0 + k at byte 15 - k (SIDE 0) / k (SIDE 1)
    const int c_min = SIDE == 0 ? lane - (d0 + CH - 1) - R + s.D + 16 : lane + d0 - R + 8;
    const int offB = c_min & ~3, shB = c_min & 3;
    uint32_t ring[RING][NQ][2];                             // row costs: quad q = windows 4q .. 4q+3, four u16
    uint32_t C[NQ][2];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      // sums start at (k & 3) of their disparity d0 + k and stay congruent to it mod 4: window 4q + i holds k = 15 - 4q - i (SIDE 0) or 4q + i (SIDE 1)
      C[q][0] = SIDE == 0 ? (3u | 2u << 16) : (0u | 1u << 16);
      C[q][1] = SIDE == 0 ? (1u | 0u << 16) : (2u | 3u << 16);
#pragma unroll
      for (int i = 0; i < RING; i++) ring[i][q][0] = ring[i][q][1] = 0;
    }
    const uint8_t* rowA = sA + offA;                        // walked by addition: no per-row multiplies
    const uint8_t* rowB = sB + offB;
    uint32_t* rowK = sKey + lane - 2 * R * 64;
    for (int tb = 0; tb < rows_tot; tb += RING) {
#pragma unroll
      for (int ri = 0; ri < RING; ri++) {
        const int t = tb + ri;
        {
          const uint32_t* pa = reinterpret_cast<const uint32_t*>(rowA);
          const uint32_t* pb = reinterpret_cast<const uint32_t*>(rowB);
          rowA += PA; rowB += PB;
          uint32_t rawA[NDW + 1], rawB[NS + 1], wa[NDW], span[NS];
#pragma unroll
          for (int j = 0; j <= NDW; j++) rawA[j] = pa[j];
#pragma unroll
          for (int j = 0; j <= NS; j++) rawB[j] = pb[j];
#pragma unroll
          for (int j = 0; j < NDW; j++) wa[j] = alignbyte(rawA[j + 1], rawA[j], shA);
          wa[NDW - 1] &= kLastMask;                           // zero = "skip" for the masked quad SAD
#pragma unroll
          for (int j = 0; j < NS; j++) span[j] = alignbyte(rawB[j + 1], rawB[j], shB);
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            uint64_t acc = 0;
#pragma unroll
            for (int j = 0; j < NDW; j++) {
              const uint64_t src = pack64(q + j + 1 < NS ? span[q + j + 1] : 0u, span[q + j]);
              acc = (j == NDW - 1 && LASTB != 4) ? __builtin_amdgcn_mqsad_pk_u16_u8(src, wa[j], acc) : __builtin_amdgcn_qsad_pk_u16_u8(src, wa[j], acc);
            }
            const uint32_t n0 = (uint32_t)acc, n1 = (uint32_t)(acc >> 32);
            C[q][0] = as_u32(as_us2(C[q][0]) - as_us2(ring[ri][q][0]) + as_us2(n0));
            C[q][1] = as_u32(as_us2(C[q][1]) - as_us2(ring[ri][q][1]) + as_us2(n1));
            ring[ri][q][0] = n0; ring[ri][q][1] = n1;
          }
          if (t >= 2 * R && t - 2 * R < band) {
            // per quad: the smaller of its four sums (4 cost | k & 3: ties go to the smaller disparity), then the global key
            uint32_t key = 0xFFFFFFFFu;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
              const int k_base = SIDE == 0 ? CH - 4 - 4 * q : 4 * q;          // smallest k of the quad
              if (k_base >= 8 && !second) continue;
              const us2 m = __builtin_elementwise_min(as_us2(C[q][0]), as_us2(C[q][1]));
              const uint32_t mm = min((uint32_t)m.x, (uint32_t)m.y);
              key = min(key, ((mm & ~3u) << 6) + (mm & 3u) + (uint32_t)(d0 + k_base));
            }
            atomicMin(rowK, key);
          }
          rowK += 64;
        }
      }
    }
  }
  __syncthreads();
  uint32_t* out = keys_out + (size_t)img * s.H * s.W;
  for (int i = tid; i < band * 64; i += 256) {
    const int ry = i >> 6, l = i & 63, x = x0 + l, y = y0 + ry;
    if (x < s.W && y < s.H) out[(size_t)y * s.W + x] = sKey[i];
  }
}

// ---- L/R check and output, integer disparities ----
// u8 (optional): the node's mono8 depth map of the same value (point_cloud.cpp:422 semantics: invalid -> 0, saturate at 255, 1/16 pixel rounded half to even)
DEV uint8_t bm_u8(int v, bool sub) {
  if (v < 0) return 0;
  if (sub) { const int q = v >> 4, r = v & 15; v = q + ((r > 8 || (r == 8 && (q & 1))) ? 1 : 0); }
  return (uint8_t)min(v, 255);
}
__global__ void __launch_bounds__(256) k_bm_finish(BmDev s, const uint32_t* __restrict__ keysL, const uint32_t* __restrict__ keysR, int16_t* __restrict__ disp,
                                                   uint8_t* __restrict__ u8) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, img = blockIdx.z;
  if (x >= s.W) return;
  const size_t row = ((size_t)img * s.H + y) * s.W;
  const int d = (int)(keysL[row + x] & 0xFFu);
  bool ok = true;
  if (s.lr >= 0) ok = x - d >= 0 && abs(d - (int)(keysR[row + x - d] & 0xFFu)) <= s.lr;
  disp[row + x] = (int16_t)(ok ? d : -1);
  if (u8) u8[row + x] = bm_u8(ok ? d : -1, false);
}

// ---- L/R check, the two costs next to the winner, 1/16-pixel formula ----
// The winner's cost is in its key; C(d-1) and C(d+1) are evaluated from the staged rows: per block row one window of a
// and two of b at per-lane byte offsets (aligned dword reads + v_alignbyte_b32), three v_sad_u8 each.
template <int R>
__global__ void __launch_bounds__(256) k_bm_finish_sub(BmDev s, int n, int band, const uint8_t* __restrict__ g, const uint32_t* __restrict__ keysL,
                                                       const uint32_t* __restrict__ keysR, int16_t* __restrict__ disp, uint8_t* __restrict__ u8) {
  constexpr int WB = 2 * R + 1, NDW = (WB + 3) / 4, LASTB = WB - 4 * (NDW - 1);
  constexpr uint32_t kLastMask = LASTB == 4 ? 0xFFFFFFFFu : ((1u << (8 * LASTB)) - 1u);
  constexpr int PA = kBmPA;
  extern __shared__ uint32_t s_mem[];
  const int PB = s.D + kBmPad, rows_tot = band + 2 * R;
  uint8_t* sA = reinterpret_cast<uint8_t*>(s_mem);
  uint8_t* sB = sA + (size_t)rows_tot * PA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * band, img = blockIdx.z;
  bm_stage<R, 0>(s, n, img, x0, y0, rows_tot, g, sA, sB);
  __syncthreads();
  const int x = x0 + lane;
  const int cA = lane + 4 - R, offA = cA & ~3, shA = cA & 3;
  for (int ry = wave; ry < band; ry += 4) {
    const int y = y0 + ry;
    if (x >= s.W || y >= s.H) continue;
    const size_t row = ((size_t)img * s.H + y) * s.W;
    const uint32_t kl = keysL[row + x];
    const int d = (int)(kl & 0xFFu);
    bool ok = true;
    if (s.lr >= 0) ok = x - d >= 0 && abs(d - (int)(keysR[row + x - d] & 0xFFu)) <= s.lr;
    int out = -16;
    if (ok) {
      out = 16 * d;
      if (d > 0 && d < s.D - 1) {
        uint32_t cm = 0, cp = 0;
        const int cM = lane - (d - 1) - R + s.D + 16, cP = cM - 2;             // windows of d - 1 and d + 1
        const int offM = cM & ~3, shM = cM & 3, offP = cP & ~3, shP = cP & 3;
#pragma unroll
        for (int j = 0; j < WB; j++) {
          const int t = ry + j;
          const uint32_t* pa = reinterpret_cast<const uint32_t*>(sA + t * PA + offA);
          const uint32_t* pm = reinterpret_cast<const uint32_t*>(sB + t * PB + offM);
          const uint32_t* pp = reinterpret_cast<const uint32_t*>(sB + t * PB + offP);
#pragma unroll
          for (int w = 0; w < NDW; w++) {
            uint32_t a = alignbyte(pa[w + 1], pa[w], shA), bm_ = alignbyte(pm[w + 1], pm[w], shM), bp = alignbyte(pp[w + 1], pp[w], shP);
            if (w == NDW - 1) { a &= kLastMask; bm_ &= kLastMask; bp &= kLastMask; }
            cm = sad_u8(a, bm_, cm); cp = sad_u8(a, bp, cp);
          }
        }
        const int c0 = (int)(kl >> 8), cm1 = (int)(cm >> 2), cp1 = (int)(cp >> 2);   // the staged bytes are 4 (g + 1)
        const int den = max(cm1 + cp1 - 2 * c0, 1);
        out = 16 * d + (16 * (cm1 - cp1) + den) / (2 * den);
      }
    }
    disp[row + x] = (int16_t)out;
    if (u8) u8[row + x] = bm_u8(out, true);
  }
}

}  // namespace

// One slot = one batch in flight: its stream, events and scratch.  Slot 0 exists from jn_bm_create and serves the synchronous calls;
// slots 1 .. kBmSlots-1 (jn_bm_submit_scan / jn_bm_wait) are allocated when first used.  Batches on different slots overlap on the GPU:
// the memory-bound prefilter / finish / scan kernels of one run next to the issue-bound matching of another.
struct BmSlot {
  uint8_t* g = nullptr;        // prefiltered rows [2 * max_batch][H][Wp]
  uint32_t* keys = nullptr;    // winners [2][max_batch][H][W]: cost << 8 | d
  unsigned long long* scan_scratch = nullptr;   // [max_batch][4], the scan tail's extrema
  int32_t* q = nullptr;        // JN_BM_COST_SSD: key halves / squared patch norms [2 * max_batch][H][Wp]
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {};
  jn_bm_times times = {};
  bool ready = false, pending = false;
};
struct jn_bm {
  jn_bm_params p;
  BmDev dev;
  int W = 0, H = 0, max_batch = 0, device = 0;
  jnav_bmq::QDev qdev = {};    // JN_BM_COST_SSD: geometry of the matrix-core path (g then holds its plain prefiltered rows)
  size_t g_bytes = 0, q_bytes = 0;
  enum { kBmSlots = 6 };
  BmSlot slot[kBmSlots];
  jn_bm_times times = {};      // of the batch waited for last
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

size_t bm_lds_bytes(const BmDev& s, int band, bool keys) {
  const int ring = 2 * s.r + 1;
  const int rows_tot = keys ? (band + 2 * s.r + ring - 1) / ring * ring : band + 2 * s.r;     // k_bm stages whole turns of its ring
  return (size_t)rows_tot * (kBmPA + s.D + kBmPad) + (keys ? (size_t)band * 64 * 4 : 0);
}

template <int R, int SIDE>
hipError_t launch_bm_one(hipStream_t st, const BmDev& s, int n, int band, const uint8_t* g, uint32_t* keys) {
  const dim3 grid((s.W + 63) / 64, (s.H + band - 1) / band, n);
  hipLaunchKernelGGL((k_bm<R, SIDE>), grid, dim3(256), bm_lds_bytes(s, band, true), st, s, n, band, g, keys);
  return hipGetLastError();
}

template <int SIDE>
hipError_t launch_bm(hipStream_t st, const BmDev& s, int n, int band, const uint8_t* g, uint32_t* keys) {
  switch (s.r) {
    case 2: return launch_bm_one<2, SIDE>(st, s, n, band, g, keys);
    case 3: return launch_bm_one<3, SIDE>(st, s, n, band, g, keys);
    default: return launch_bm_one<4, SIDE>(st, s, n, band, g, keys);
  }
}

hipError_t launch_bm_finish(hipStream_t st, const BmDev& s, int n, const uint8_t* g, const uint32_t* keysL, const uint32_t* keysR, int16_t* disp, uint8_t* u8) {
  if (!s.subpixel) {
    hipLaunchKernelGGL(k_bm_finish, dim3((s.W + 255) / 256, s.H, n), dim3(256), 0, st, s, keysL, keysR, disp, u8);
    return hipGetLastError();
  }
  const int band = 16;
  const dim3 grid((s.W + 63) / 64, (s.H + band - 1) / band, n);
  const size_t lds = bm_lds_bytes(s, band, false);
  switch (s.r) {
    case 2: hipLaunchKernelGGL(k_bm_finish_sub<2>, grid, dim3(256), lds, st, s, n, band, g, keysL, keysR, disp, u8); break;
    case 3: hipLaunchKernelGGL(k_bm_finish_sub<3>, grid, dim3(256), lds, st, s, n, band, g, keysL, keysR, disp, u8); break;
    default: hipLaunchKernelGGL(k_bm_finish_sub<4>, grid, dim3(256), lds, st, s, n, band, g, keysL, keysR, disp, u8); break;
  }
  return hipGetLastError();
}

}  // namespace

static jn_status bm_ensure_slot(jn_bm* h, int k) {
  BmSlot& x = h->slot[k];
  if (x.ready) return JN_OK;
  if (h->q_bytes) BM_TRY(hipMalloc(reinterpret_cast<void**>(&x.q), h->q_bytes));
  BM_TRY(hipMalloc(reinterpret_cast<void**>(&x.g), h->g_bytes));
  BM_TRY(hipMalloc(reinterpret_cast<void**>(&x.keys), (size_t)2 * h->max_batch * h->H * h->W * sizeof(uint32_t)));
  BM_TRY(hipMalloc(reinterpret_cast<void**>(&x.scan_scratch), sizeof(unsigned long long) * 4 * h->max_batch));
  BM_TRY(hipStreamCreateWithFlags(&x.stream, hipStreamNonBlocking));
  for (auto& e : x.ev) BM_TRY(hipEventCreate(&e));
  x.ready = true;
  return JN_OK;
}

extern "C" {

void jn_bm_params_default(jn_bm_params* p) {
  p->num_disparities = 64; p->block_radius = 4; p->prefilter_cap = 31; p->lr_max_diff = 1; p->subpixel = 0; p->cost_function = JN_BM_COST_SAD;
}

void jn_bm_destroy(jn_bm* h) {
  if (!h) return;
  hipSetDevice(h->device);
  for (auto& x : h->slot) {
    if (x.stream) hipStreamSynchronize(x.stream);
    hipFree(x.g); hipFree(x.keys); hipFree(x.scan_scratch); hipFree(x.q);
    for (auto& e : x.ev) if (e) hipEventDestroy(e);
    if (x.stream) hipStreamDestroy(x.stream);
  }
  delete h;
}

jn_status jn_bm_create(const jn_bm_params* p, int32_t W, int32_t H, int32_t max_batch, int32_t device, jn_bm** out) {
  if (!p || !out || W < 8 || H < 8 || W > 8192 || H > 8192 || max_batch < 1) return JN_ERR_INVALID;
  *out = nullptr;
  const int D = p->num_disparities;
  if (D < 8 || D > 256 || (D & 7) || p->block_radius < 2 || p->block_radius > 4 || p->prefilter_cap < 1 || p->prefilter_cap > 31)
    return JN_ERR_UNSUPPORTED;
  const bool ssd = p->cost_function == JN_BM_COST_SSD;
  if (p->cost_function != JN_BM_COST_SAD && !ssd) return JN_ERR_UNSUPPORTED;
  if (ssd && (D & 31)) return JN_ERR_UNSUPPORTED;              // the matrix-core path covers the band with whole tiles of 32 candidates
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return JN_ERR_NO_DEVICE;
  BM_TRY(hipSetDevice(device));
  jn_bm* h = new jn_bm();
  h->p = *p; h->W = W; h->H = H; h->max_batch = max_batch; h->device = device;
  BmDev& s = h->dev;
  s.W = W; s.H = H; s.D = D; s.r = p->block_radius; s.cap = p->prefilter_cap; s.lr = p->lr_max_diff; s.subpixel = p->subpixel ? 1 : 0;
  s.padx = D + kBmPad; s.Wp = (W + 2 * s.padx + 3) & ~3;
#define BM_CREATE_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { fprintf(stderr, "libjn_stereo: %s failed: %s\n", #expr, hipGetErrorString(e__)); jn_bm_destroy(h); return JN_ERR_NO_DEVICE; } } while (0)
  h->g_bytes = (size_t)2 * max_batch * H * s.Wp + 64;
  if (ssd) {
    jnav_bmq::Sizes z;
    jnav_bmq::geometry(W, H, D, p->block_radius, p->prefilter_cap, p->lr_max_diff, p->subpixel, &h->qdev, &z, max_batch);
    h->g_bytes = z.g; h->q_bytes = z.q;
  }
#undef BM_CREATE_TRY
  const jn_status es = bm_ensure_slot(h, 0);
  if (es != JN_OK) { jn_bm_destroy(h); return es; }
  *out = h;
  return JN_OK;
}

static jn_status bm_submit(jn_bm* h, int k, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch, int64_t image_stride, int16_t* dDisp,
                           const jn_scan_params* sp, const uint8_t* dLut, uint8_t* dU8, double* dBins, double* dMeta) {
  if (!h || k < 0 || k >= jn_bm::kBmSlots || n < 1 || n > h->max_batch || !dI1 || !dI2 || !dDisp || pitch < h->W) return JN_ERR_INVALID;
  if (h->slot[k].pending) return JN_ERR_INVALID;               // one batch per slot: jn_bm_wait first
  BM_TRY(hipSetDevice(h->device));
  const jn_status es = bm_ensure_slot(h, k);
  if (es != JN_OK) return es;
  BmSlot& x = h->slot[k];
  const BmDev& s = h->dev;
  hipStream_t st = x.stream;
  BM_TRY(hipEventRecord(x.ev[0], st));
  uint32_t* keysL = x.keys;
  uint32_t* keysR = x.keys + (size_t)h->max_batch * s.H * s.W;
  if (h->p.cost_function == JN_BM_COST_SSD) {                  // squared differences: the banded int8 contraction on the matrix cores (bm_mfma.hip)
    BM_TRY(jnav_bmq::run(h->qdev, n, dI1, dI2, pitch, (long long)image_stride, x.g, x.q, keysL, keysR, dDisp, dU8, st, x.ev));
  } else {
    hipLaunchKernelGGL(k_bm_prefilter, dim3((s.Wp / 4 + 255) / 256, s.H, 2 * n), dim3(256), 0, st, s, dI1, dI2, pitch, (long long)image_stride, n, x.g);
    BM_TRY(hipEventRecord(x.ev[1], st));
    // Rows per band: whole turns of the kernel's ring (band + 2r = k (2r+1)) so that no staged row is wasted, as many as
    // fit 64 rows (halo overhead 2r / band), fewer turns while the launch would leave most of the 256 CUs idle (a lone pair).
    const int ring = 2 * s.r + 1;
    int turns = (kBmMaxBand + 2 * s.r) / ring;
    while (turns > 2 && (long long)((s.W + 63) / 64) * ((s.H + turns * ring - 2 * s.r - 1) / (turns * ring - 2 * s.r)) * n < 1024) turns--;
    int band = turns * ring - 2 * s.r;
    if (const char* e = JN_HOOK_ENV("JN_BM_BAND")) band = std::min(std::max(atoi(e), 1), kBmMaxBand);
    BM_TRY(launch_bm<0>(st, s, n, band, x.g, keysL));
    if (s.lr >= 0) BM_TRY(launch_bm<1>(st, s, n, band, x.g, keysR));
    BM_TRY(hipEventRecord(x.ev[2], st));
    BM_TRY(launch_bm_finish(st, s, n, x.g, keysL, keysR, dDisp, dU8));
  }
  if (sp) jnav::launch_scan(st, *sp, n, nullptr, dU8, dLut, s.W, s.H, dBins, dMeta, x.scan_scratch);   // the node's tail, same stream
  BM_TRY(hipEventRecord(x.ev[3], st));
  BM_TRY(hipGetLastError());
  x.pending = true;
  return JN_OK;
}
static jn_status bm_wait(jn_bm* h, int k) {
  if (!h || k < 0 || k >= jn_bm::kBmSlots) return JN_ERR_INVALID;
  BmSlot& x = h->slot[k];
  if (!x.pending) return JN_OK;
  BM_TRY(hipSetDevice(h->device));
  x.pending = false;
  BM_TRY(hipStreamSynchronize(x.stream));
  BM_TRY(hipGetLastError());
  hipEventElapsedTime(&x.times.prefilter, x.ev[0], x.ev[1]);
  hipEventElapsedTime(&x.times.match, x.ev[1], x.ev[2]);
  hipEventElapsedTime(&x.times.finish, x.ev[2], x.ev[3]);
  hipEventElapsedTime(&x.times.total, x.ev[0], x.ev[3]);
  h->times = x.times;
  return JN_OK;
}

jn_status jn_bm_process_batch(jn_bm* h, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch, int64_t image_stride, int16_t* dDisp) {
  const jn_status r = bm_submit(h, 0, n, dI1, dI2, pitch, image_stride, dDisp, nullptr, nullptr, nullptr, nullptr, nullptr);
  return r != JN_OK ? r : bm_wait(h, 0);
}

jn_status jn_bm_process_scan(jn_bm* h, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch, int64_t image_stride, int16_t* dDisp,
                             const jn_scan_params* sp, const uint8_t* dLut, uint8_t* dDispU8, double* dBins, double* dMeta) {
  if (!sp || !dLut || !dDispU8 || !dBins || !dMeta || sp->bins < 1 || sp->bins > 1024) return JN_ERR_INVALID;
  const jn_status r = bm_submit(h, 0, n, dI1, dI2, pitch, image_stride, dDisp, sp, dLut, dDispU8, dBins, dMeta);
  return r != JN_OK ? r : bm_wait(h, 0);
}

jn_status jn_bm_submit_scan(jn_bm* h, int32_t slot, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch, int64_t image_stride, int16_t* dDisp,
                            const jn_scan_params* sp, const uint8_t* dLut, uint8_t* dDispU8, double* dBins, double* dMeta) {
  if (sp && (!dLut || !dDispU8 || !dBins || !dMeta || sp->bins < 1 || sp->bins > 1024)) return JN_ERR_INVALID;
  return bm_submit(h, slot, n, dI1, dI2, pitch, image_stride, dDisp, sp, dLut, dDispU8, dBins, dMeta);
}

jn_status jn_bm_wait(jn_bm* h, int32_t slot) { return bm_wait(h, slot); }

jn_status jn_bm_last_times(jn_bm* h, jn_bm_times* out) {
  if (!h || !out) return JN_ERR_INVALID;
  *out = h->times;
  return JN_OK;
}

}  // extern "C"
