// sgm_sweep.hip — the semi-global-matching mode (include/jn_sgm.h) as four sweeps with lanes = pixels.  Product code.
//
// No reference counterpart (the reference's only matcher is libelas); the definition is in jn_sgm.h, everything is integer
// arithmetic and the bar is bit-exactness against its scalar restatement (the checker, outside the product).
//
// Why this decomposition (round 3; round 2's one-wave-per-line kernels are described in DESIGN_HISTORY.md).  With lanes = disparities
// a path pixel costs ~27 wave-instructions (a 6-step DPP prefix-min and two wave shifts per pixel) and every direction
// writes and re-reads its own W*H*D volume (16 W H D bytes of traffic).  Here a lane owns a PIXEL and DPL = D/4
// consecutive disparities of it (four lanes per pixel), the path values live in registers as packed u16 pairs and all the
// arithmetic is v_pk_min_u16 / v_pk_add_u16, two cells per lane-instruction:
//   * the minimum over d is in-lane (plus two cross-lane steps over the four lanes of the pixel), L(p-r, d+-1) are register
//     neighbours (one v_alignbit per register), the 1x3 SAD costs of four consecutive disparities come from ONE
//     v_mqsad_pk_u16_u8 on the prefiltered right row (the prefilter stores g+1, so a zero byte in the left reference
//     masks the fourth tap), already packed as the recurrence wants them;
//   * the recurrence is carried NORMALISED: Lq = L(p-r, .) - min L(p-r, .), so L(p, d) = C + min(Lq[d], Lq[d+-1] + P1, P2)
//     and what a path contributes beyond the cost, m = L - C, lies in [0, P2]: S = sum_r L_r = 8 C + sum_r m_r.  The
//     sweeps store sums of m (bytes), and the last sweep adds 8 C, which it computes anyway;
//   * k_sw_h: the two horizontal paths, lanes = 16 image rows x 4 disparity quarters, walking along x (one volume each);
//   * row sweep <FINAL=false>: the three downward paths (0,1), (1,1), (-1,1) in ONE top-to-bottom sweep, summed in
//     registers, one byte volume out;  row sweep <FINAL=true>: the three upward paths in one bottom-to-top sweep which
//     also reads the three stored volumes, forms S, takes the left winner (keys S << 16 | j), the right image's
//     winners (LDS atomic minima, flushed per row with global atomic minima) and the sub-pixel offset;
//     k_sw_lr applies the L/R check (or, with a scan behind the mode, k_scan<false, true> in kernels.hip while it scans).
// HBM traffic: 3 volumes written + 3 read = 6 W H D (+ images), against 16 W H D before; SURVEY 8d's bound is 4 W H D.
//
// The row sweeps and their one-directional pipeline.  A pixel's three downward paths need the previous row at x, x-1 and
// x+1, so 16-pixel strips of a row sweep cannot be independent.  In the SHEARED coordinate x' = x - y (a lane keeps x' and
// so walks along the (1,1) diagonal) the three predecessors sit at x'+0, x'+1 and x'+2: all on ONE side.  A strip then
// depends only on its right neighbour's first two columns of the previous row — a pipeline, not a lock-step.  k_sw_w (see
// there): no workgroup barrier and no communication wave — an LDS ring with row counters between the strips of a block,
// self-validating tagged columns between blocks.  (Round 3's first form, k_sw_v — one barrier per row, the columns between
// blocks fetched by a wave that did no arithmetic, 14 % slower — left the library in round 5: DESIGN_HISTORY.md.)
// Workgroups take a ticket when they start, and tickets
// are numbered so that a block's producer always holds a smaller one: whatever the dispatch order, a waiting block's
// producer is running or done (placement-independent, no co-residency assumption).
// The kernels work in x-mirrored image space (x_k = W-1-x), where the right-image tap x - d becomes x_k + d and the bytes a
// lane needs ascend with d; the upward sweep is the downward one on the row-flipped image.
#include <hip/hip_runtime.h>
#include "hooks.h"
#include <stdint.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "sgm_sweep.h"
#include "prefilter.h"

namespace jnav_sgm {

#define DEV static __device__ __forceinline__

constexpr int NQ = 4;            // lanes per pixel

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
DEV u16x2 v2(uint32_t a) { return __builtin_bit_cast(u16x2, a); }
DEV uint32_t u1(u16x2 a) { return __builtin_bit_cast(uint32_t, a); }
DEV uint32_t pk_min(uint32_t a, uint32_t b) { return u1(__builtin_elementwise_min(v2(a), v2(b))); }
DEV uint32_t pk_max(uint32_t a, uint32_t b) { return u1(__builtin_elementwise_max(v2(a), v2(b))); }
DEV uint32_t pk_add(uint32_t a, uint32_t b) { return u1((u16x2)(v2(a) + v2(b))); }
DEV uint32_t pk_sub(uint32_t a, uint32_t b) { return u1((u16x2)(v2(a) - v2(b))); }
DEV uint32_t pk_subsat(uint32_t a, uint32_t b) { return u1(__builtin_elementwise_sub_sat(v2(a), v2(b))); }   // max(a - b, 0) per half
DEV uint32_t pk_shl3(uint32_t a) { return u1((u16x2)(v2(a) << (u16x2)(3))); }
DEV uint32_t bperm(int src_lane, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v); }
DEV uint32_t load_u32_unaligned(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
DEV uint64_t load_u64_unaligned(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

// REGISTER LAYOUT.  A lane owns DPL = 2 NR disparities j = 0 .. DPL-1 of its quarter (d = DPL q + j); register r holds the
// pair (j = r, j = r + NR) in its low / high half.  With this "stride-NR" pairing both halves of register r have their
// disparity neighbours in registers r-1 and r+1, in the same halves: max(X[r-1], X[r+1]) needs no shuffling; only r = 0 and
// r = NR-1 wrap (the low half of register 0 continues the quarter below, the high half of register NR-1 the quarter above).
//
// ARITHMETIC.  The recurrence is carried normalised, clamped and negated: X = P2 - min(L(p-r, .) - min L(p-r, .), P2), so
//   Y = max(X[d], max(X[d-1], X[d+1]) (-) P1)      ((-) saturates at 0; a neighbour that does not exist is X = 0)
//   L(p, d) - C = P2 - Y,   Ln = (C + P2) - Y,   X_new = (P2 + min Ln) (-) Ln
// is exactly include/jn_sgm.h's L = C + min(Lq[d], Lq[d+-1] + P1, P2) with Lq = L(p-r, .) - min: seven packed
// instructions per register and path (max, sub, max, add into the sum of Y, sub, min, sub).  A path entering the image has
// Lq = 0, i.e. X = P2.  The sweeps store sums of Y (bytes; Y <= P2); S = sum_r L_r = 8 (C + P2) - sum_r Y_r.

// 1x3 SAD costs + P2.  run: right-row bytes from column x_k - 1 + DPL q on (NR/2 + 1 dwords), ref: left-row bytes x_k-1, x_k,
// x_k+1 (top byte 0 = masked by v_mqsad).  One v_mqsad gives 4 consecutive disparities; a v_perm per register re-pairs them.
template <int NR>
DEV void costs(const uint32_t (&w)[NR / 2 + 1], uint32_t ref, uint32_t P2pk, uint32_t (&Cp)[NR]) {
  uint32_t a[NR];                                              // a[2k] = (c[4k], c[4k+1]), a[2k+1] = (c[4k+2], c[4k+3])
  const uint64_t p2 = (uint64_t)P2pk | ((uint64_t)P2pk << 32);
#pragma unroll
  for (int k = 0; k < NR / 2; k++) {
    const uint64_t r = __builtin_amdgcn_mqsad_pk_u16_u8((uint64_t)w[k] | ((uint64_t)w[k + 1] << 32), ref, p2);
    a[2 * k] = (uint32_t)r; a[2 * k + 1] = (uint32_t)(r >> 32);
  }
#pragma unroll
  for (int r = 0; r < NR; r++)                                 // (c[r], c[r + NR]): c[j] is half (j & 1) of a[j >> 1]
    Cp[r] = __builtin_amdgcn_perm(a[(r + NR) >> 1], a[r >> 1], (r & 1) ? 0x07060302u : 0x05040100u);
}

// the same with the 8-byte windows already paired (ww[k] = bytes 4k .. 4k+7 of the run)
template <int NR>
DEV void costs64(const uint64_t (&ww)[NR / 2], uint32_t ref, uint32_t P2pk, uint32_t (&Cp)[NR]) {
  uint32_t a[NR];
  const uint64_t p2 = (uint64_t)P2pk | ((uint64_t)P2pk << 32);
#pragma unroll
  for (int k = 0; k < NR / 2; k++) {
    const uint64_t r = __builtin_amdgcn_mqsad_pk_u16_u8(ww[k], ref, p2);
    a[2 * k] = (uint32_t)r; a[2 * k + 1] = (uint32_t)(r >> 32);
  }
#pragma unroll
  for (int r = 0; r < NR; r++) Cp[r] = __builtin_amdgcn_perm(a[(r + NR) >> 1], a[r >> 1], (r & 1) ? 0x07060302u : 0x05040100u);
}

// Three-input packed maxima / minima.  gfx950's v_pk_maximum3_f16 / v_pk_minimum3_f16 compare IEEE halves; every value the recurrence
// holds is a u16 below 0x7C00 (a positive finite half, denormals included, which these instructions neither flush nor quieten), and on
// those the order of the halves IS the unsigned order: scripts/probes/pk3_probe.hip checks all 2^30 pairs of such patterns on the device.
// (Written with the element-wise builtins, which the back end folds into the three-input forms: behind inline assembly the hazard
// recogniser pads every use of the result with an s_nop.)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
DEV uint32_t pk_max3(uint32_t a, uint32_t b, uint32_t c) {
  const f16x2 x = __builtin_bit_cast(f16x2, a), y = __builtin_bit_cast(f16x2, b), z = __builtin_bit_cast(f16x2, c);
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(x, y), z));
}
DEV uint32_t pk_min3(uint32_t a, uint32_t b, uint32_t c) {
  const f16x2 x = __builtin_bit_cast(f16x2, a), y = __builtin_bit_cast(f16x2, b), z = __builtin_bit_cast(f16x2, c);
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_minimum(__builtin_elementwise_minimum(x, y), z));
}

// The adjoining pairs of the neighbouring quarters, already reduced by P1 (T = X (-) P1: max(a, b) (-) P1 = max(a (-) P1, b (-) P1)):
// up = the quarter below's j = DPL-1 (high half of its register NR-1) for this lane's j = 0; dn = the quarter above's j = 0 for this
// lane's j = DPL-1.  Issued ahead of the cells that use them.
template <int NR, int LQ = 4>
DEV void path_neighbours(const uint32_t (&X)[NR], int lane, int q, uint32_t P1pk, uint32_t& up, uint32_t& dn) {
  constexpr int NQ = LQ, PX = 64 / LQ;                         // lanes per pixel, pixels per wave (the file-scope values are the 4-lane layout's)
  const uint32_t a = bperm((lane - PX) & 63, pk_subsat(X[NR - 1], P1pk));
  const uint32_t b = bperm((lane + PX) & 63, pk_subsat(X[0], P1pk));
  up = q == 0 ? 0u : a;
  dn = q == NQ - 1 ? 0u : b;
}
// 5.5 packed instructions per register and path: T = X (-) P1, Y = max3(X[r], T[r-1], T[r+1]), sum of Y, Ln = Cp - Y, half a min3 for
// the minimum of Ln, and (path_normalise) X' = (P2 + min Ln) (-) Ln.
// SUB: the sum is carried negated (acc -= Y): the final sweep starts acc at 8 (C + P2) - (the five stored paths) and ends with S.
template <int NR, bool SUB = false>
DEV void path_cells(const uint32_t (&X)[NR], uint32_t up, uint32_t dn, const uint32_t (&Cp)[NR], uint32_t (&acc)[NR], uint32_t (&Ln)[NR], uint32_t& mn,
                    uint32_t P1pk) {
  uint32_t T[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) T[r] = pk_subsat(X[r], P1pk);
  auto cell = [&](int r, uint32_t lo, uint32_t hi) __attribute__((always_inline)) {
    const uint32_t y = pk_max3(X[r], lo, hi);
    acc[r] = SUB ? pk_sub(acc[r], y) : pk_add(acc[r], y);
    Ln[r] = pk_sub(Cp[r], y);
  };
  // register 0: j = 0 has j-1 in the quarter below, j = NR has j-1 = NR-1 in the low half of register NR-1
  cell(0, __builtin_amdgcn_perm(T[NR - 1], up, 0x05040302u), T[1]);
#pragma unroll
  for (int r = 1; r < NR - 1; r++) cell(r, T[r - 1], T[r + 1]);
  // register NR-1: j = NR-1 has j+1 = NR in the high half of register 0, j = DPL-1 has j+1 in the quarter above
  cell(NR - 1, T[NR - 2], __builtin_amdgcn_perm(dn, T[0], 0x05040302u));
  mn = pk_min(Ln[0], Ln[1]);
#pragma unroll
  for (int r = 2; r < NR; r += 2) mn = pk_min3(mn, Ln[r], Ln[r + 1]);
}
// minimum over the pixel's four lanes (and both halves) with the gfx950 row / half swaps: pure VALU, no LDS round trip
template <int LQ = 4>
DEV uint32_t pixel_min(uint32_t mn) {
  uint32_t m = min(mn & 0xFFFFu, mn >> 16);
  if constexpr (LQ == 8) m = min(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x128, 0xf, 0xf, false));   // row_ror:8: the lane 8 further, inside the 16-lane DPP row
  const auto a = __builtin_amdgcn_permlane16_swap(m, m, false, false);
  m = min(a[0], a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(m, m, false, false);
  return min(b[0], b[1]);
}
template <int NR>
DEV void path_normalise(uint32_t (&X)[NR], const uint32_t (&Ln)[NR], uint32_t m, uint32_t P2) {
  const uint32_t t = (m + P2) * 0x10001u;
#pragma unroll
  for (int r = 0; r < NR; r++) X[r] = pk_subsat(t, Ln[r]);
}

// register pairs with values <= 255 -> 4 bytes (lo a, hi a, lo b, hi b); and back
DEV uint32_t pack4(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x06040200u); }
DEV uint32_t unpack_lo(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c010c00u); }
DEV uint32_t unpack_hi(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c030c02u); }
// VOLUME LAYOUT.  A pixel's D bytes are stored as 16-byte pieces: piece c of quarter q at byte 64 c + 16 q, so that the four
// lanes of a pixel write (and read) 64 contiguous bytes per instruction.  Within a lane, piece c holds registers 8c .. 8c+7
// as pack4 pairs.  Only these kernels read the volumes, so the order of the disparities inside a pixel is theirs to choose.
template <int NR, int LQ = 4>
DEV void store_bytes(uint8_t* pixel_base, int q, const uint32_t (&acc)[NR]) {
#pragma unroll
  for (int c = 0; c < NR / 8; c++)
    *reinterpret_cast<uint4*>(pixel_base + 16 * LQ * c + 16 * q) = make_uint4(pack4(acc[8 * c], acc[8 * c + 1]), pack4(acc[8 * c + 2], acc[8 * c + 3]),
                                                                          pack4(acc[8 * c + 4], acc[8 * c + 5]), pack4(acc[8 * c + 6], acc[8 * c + 7]));
}
template <int NR, int LQ = 4>
DEV void load_bytes(const uint8_t* pixel_base, int q, uint32_t (&f)[NR / 2]) {
#pragma unroll
  for (int c = 0; c < NR / 8; c++) {
    const uint4 v = *reinterpret_cast<const uint4*>(pixel_base + 16 * LQ * c + 16 * q);
    f[4 * c] = v.x; f[4 * c + 1] = v.y; f[4 * c + 2] = v.z; f[4 * c + 3] = v.w;
  }
}
// 16-bit form (three-path sums beyond 255): piece c of quarter q holds registers 4c .. 4c+3 as they are
template <int NR, int LQ = 4>
DEV void store_words(uint8_t* pixel_base, int q, const uint32_t (&acc)[NR]) {
#pragma unroll
  for (int c = 0; c < NR / 4; c++) *reinterpret_cast<uint4*>(pixel_base + 16 * LQ * c + 16 * q) = make_uint4(acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]);
}
template <int NR, int LQ = 4>
DEV void load_words(const uint8_t* pixel_base, int q, uint32_t (&f)[NR]) {
#pragma unroll
  for (int c = 0; c < NR / 4; c++) {
    const uint4 v = *reinterpret_cast<const uint4*>(pixel_base + 16 * LQ * c + 16 * q);
    f[4 * c] = v.x; f[4 * c + 1] = v.y; f[4 * c + 2] = v.z; f[4 * c + 3] = v.w;
  }
}

// ---- prefilter: mirrored, padded, +1 ----
// gm[img][y][padl + x_k] = clamp(Sobel_x(I, W-1-cl(x_k), y), -cap, cap) + cap + 1, cl = clamp to [0, W-1] (replicated borders:
// the cost's coordinate clamps become plain reads).  +1 keeps every byte non-zero for v_mqsad's mask.
// A thread owns four padded columns of kPreRows consecutive rows and walks down them: the horizontal differences h(y) = I(x+1, y) - I(x-1, y)
// of a row are formed once and used by three output rows (Sobel_x = h(y-1) + 2 h(y) + h(y+1)), one 8-byte load per row instead of three
// (round 4's form — one thread per four bytes of ONE row, 92 k workgroups a batch — took 0.12 ms, a quarter of the chip's copy rate).
constexpr int kPreRows = 8;
__global__ void __launch_bounds__(256) k_sw_prefilter(SwDev s, const uint8_t* __restrict__ I1, const uint8_t* __restrict__ I2, int pitch,
                                                      long long stride, int n, uint8_t* __restrict__ gm) {
  const int xp = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * kPreRows, img = blockIdx.z;
  if (xp >= s.Wp || y0 >= s.H) return;                          // (Wp is a multiple of 16)
  const uint8_t* I = img < n ? I1 + (long long)img * stride : I2 + (long long)(img - n) * stride;
  const int W = s.W, H = s.H;
  const int xk0 = xp - s.padl;                                  // mirrored column of the first of the four; the image column DEscends with it
  const int hi = W - 1 - xk0, lo = hi - 3;                      // image columns of the four outputs when none needs a clamp: output k is column hi - k
  const bool fast = xk0 >= 0 && xk0 + 3 <= W - 1 && lo >= 1 && hi <= W - 2 && lo + 6 <= W - 1;    // the 8-byte window lo-1 .. lo+6 lies inside the row
  int xq[4], xm[4];                                             // otherwise: per output the clamped columns x+1 / x-1 (the row's padding replicates the outermost columns)
#pragma unroll
  for (int k = 0; k < 4; k++) { const int x = W - 1 - min(max(xk0 + k, 0), W - 1); xq[k] = min(x + 1, W - 1); xm[k] = max(x - 1, 0); }
  auto hrow = [&](int y, int (&h)[4]) __attribute__((always_inline)) {
    const uint8_t* r = I + (size_t)min(max(y, 0), H - 1) * pitch;
    if (fast) {
      uint64_t a;
      __builtin_memcpy(&a, r + lo - 1, 8);
#pragma unroll
      for (int k = 0; k < 4; k++) { const int i = 4 - k; h[k] = (int)((a >> (8 * (i + 1))) & 255u) - (int)((a >> (8 * (i - 1))) & 255u); }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) h[k] = (int)r[xq[k]] - (int)r[xm[k]];
    }
  };
  int ha[4], hb[4], hc[4];
  hrow(y0 - 1, ha); hrow(y0, hb);
  const uint32_t c1 = (uint32_t)(s.cap + 1);
  uint8_t* out = gm + ((size_t)img * H + y0) * s.Wp + xp;
#pragma unroll
  for (int r = 0; r < kPreRows; r++) {
    if (y0 + r >= H) break;
    hrow(y0 + r + 1, hc);
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int g = min(max(ha[k] + 2 * hb[k] + hc[k], -s.cap), s.cap);
      w |= ((uint32_t)g + c1) << (8 * k);
      ha[k] = hb[k]; hb[k] = hc[k];
    }
    *reinterpret_cast<uint32_t*>(out + (size_t)r * s.Wp) = w;
  }
}

// ---- horizontal paths: lanes = 16 rows x 4 disparity quarters; blockIdx.z = 0 walks x_k upwards, 1 downwards ----
// The bytes a lane needs at x_k + 1 are those of x_k shifted by one: the lane keeps an ALIGNED window of its row in registers
// (rows start 16-byte aligned and every lane of the wave is at the same x_k, so the byte phase is wave-uniform), forms each
// pixel's dwords with v_alignbyte, and loads one new dword every fourth pixel — four pixels ahead of its use.
// Stores: 64-byte pieces in 16 different image rows per instruction.  Collecting 512 contiguous bytes per row in LDS first (two
// rows per store instruction) was built and measured: no change (2 824 against 2 827 pairs/s) — the kernel writes 7.5 GB in 2.0 ms,
// it is the write rate itself, not the shape of the writes, that bounds it next to its 1.25 ms of arithmetic.
// Where the entering dwords come from.  Loaded from memory under the wrap condition (as in the first form of this kernel) the loop needs a
// vector-memory wait, and since the compiler cannot count operations issued under conditions that wait is s_waitcnt vmcnt(0) before EVERY
// pixel: every store of the previous pixel was drained first.  Now the wave stages the next 64 bytes of its 16 rows (+ the other quarters'
// offsets) in LDS once per 16 wraps — coalesced loads, one full wait per 64 pixels — and the wrap reads its two dwords from there: LDS reads
// count in lgkmcnt, so nothing in the pixel loop waits for memory and the stores stay in flight.
// (Also measured, bit-exact, slower: every v_mqsad operand loaded per pixel as the row sweeps do — 4.6 ms instead of 2.1, the lanes are 16
// different image rows here and each load instruction touches dozens of cache lines; the entering dwords reloaded after every pixel — 3.1 ms.)
constexpr int HCH = 16;                                        // wraps (dwords per row and quarter) staged at a time
template <int NR, int DIR, int LQ>
DEV void h_sweep(const SwDev& s, const uint8_t* __restrict__ gmL, const uint8_t* __restrict__ gmR, int y0, uint8_t* __restrict__ vol, bool valid, int lane, int q,
                 uint32_t* __restrict__ sR, uint32_t* __restrict__ sL) {
  constexpr int NW = NR / 2 + 1;                               // dwords of one pixel's run
  constexpr int NQ = LQ, PX = 64 / LQ;                         // lanes per pixel, image rows per wave
  constexpr int DPL = 2 * NR, SPAN = HCH + (NQ - 1) * DPL / 4; // staged dwords per row of the right image
  const int W = s.W, p = lane & (PX - 1);
  const uint32_t P1pk = (uint32_t)s.P1 * 0x10001u, P2pk = (uint32_t)s.P2 * 0x10001u;
  // gmL / gmR: column x_k = 0 of image row 0 of this frame's left / right prefiltered image; this lane's row is min(y0 + p, H - 1)
  const int yc = min(y0 + p, s.H - 1);
  const uint8_t* rowL = gmL + (size_t)yc * s.Wp;
  const uint8_t* rowR = gmR + (size_t)yc * s.Wp + DPL * q;
  uint32_t X[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) X[r] = P2pk;
  // rowL / rowR point at column x_k = 0 of this lane's row (rowR at its quarter's first disparity); the run of pixel x_k starts
  // at byte x_k - 1.  Window: aligned dwords Rw[0 .. NW+1] from aligned byte address `al`; Lw[0 .. 2] likewise for the left row.
  int xk = DIR ? W - 1 : 0;
  int al = (xk - 1) & ~3;                                      // relative to the row pointers (may be -4)
  uint32_t Rw[NW + 2], Lw[3];
#pragma unroll
  for (int k = 0; k < NW + 2; k++) Rw[k] = *reinterpret_cast<const uint32_t*>(rowR + al + 4 * (DIR ? k - 1 : k));
#pragma unroll
  for (int k = 0; k < 3; k++) Lw[k] = *reinterpret_cast<const uint32_t*>(rowL + al + 4 * (DIR ? k - 1 : k));
  // upwards: Rw[k] = dword at al + 4k (k = NW+1 is the one fetched ahead); downwards: Rw[k] = dword at al + 4(k-1) (k = 0 fetched ahead)
  // the window is complete before the loop starts (uses the compiler must wait for): a wait for it inside the loop would be executed before
  // every pixel and drain the previous pixel's stores
#pragma unroll
  for (int k = 0; k < NW + 2; k++) asm volatile("" : : "v"(Rw[k]));
#pragma unroll
  for (int k = 0; k < 3; k++) asm volatile("" : : "v"(Lw[k]));
  const int lo_byte = -s.padl, hi_byte = s.Wp - s.padl - 4;    // a row's own bytes (staging near the ends clamps into them; those dwords are never used)
  int wraps_left = 0;                                          // entering dwords still staged
  int soff = 0;                                                // dword index of the next wrap's entering dword within the staged span
  const uint32_t* myR = sR + p * SPAN + (DPL / 4) * q;
  const uint32_t* myL = sL + p * HCH;
  for (int t = 0; t < W; t++) {
    const int sh = (xk - 1) & 3;
    if (wraps_left == 0) {
      // the next HCH wraps fetch (upwards) al + 4 (i + 1) + 4 (NW + 1) resp. al + 4 (i + 1) + 8, (downwards) al - 4 (i + 1) - 4 for both rows
      const int baseR = DIR ? al - 4 * HCH - 4 : al + 4 + 4 * (NW + 1), baseL = DIR ? al - 4 * HCH - 4 : al + 12;
      constexpr int KR = (PX * SPAN + 63) / 64, KL = PX * HCH / 64;
      uint32_t tR[KR], tL[KL];                                 // all loads first, then all LDS writes
#pragma unroll
      for (int k = 0; k < KR; k++) {
        const int i = min(lane + 64 * k, PX * SPAN - 1), row = i / SPAN, col = i - row * SPAN;
        tR[k] = *reinterpret_cast<const uint32_t*>(gmR + (size_t)min(y0 + row, s.H - 1) * s.Wp + min(max(baseR + 4 * col, lo_byte), hi_byte));
      }
#pragma unroll
      for (int k = 0; k < KL; k++) {
        const int i = lane + 64 * k, row = i / HCH, col = i % HCH;
        tL[k] = *reinterpret_cast<const uint32_t*>(gmL + (size_t)min(y0 + row, s.H - 1) * s.Wp + min(max(baseL + 4 * col, lo_byte), hi_byte));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < KR; k++) { const int i = lane + 64 * k; if ((PX * SPAN) % 64 == 0 || i < PX * SPAN) sR[i] = tR[k]; }
#pragma unroll
      for (int k = 0; k < KL; k++) sL[lane + 64 * k] = tL[k];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      wraps_left = HCH;
      soff = DIR ? HCH - 1 : 0;
    }
    uint32_t w[NW], Cp[NR], acc[NR], Ln[NR], up, dn, mn;
#pragma unroll
    for (int k = 0; k < NW; k++) w[k] = __builtin_amdgcn_alignbyte(Rw[k + 1 + DIR], Rw[k + DIR], sh);
    const uint32_t ref = __builtin_amdgcn_alignbyte(Lw[1 + DIR], Lw[DIR], sh) & 0x00FFFFFFu;
    costs<NR>(w, ref, P2pk, Cp);
#pragma unroll
    for (int r = 0; r < NR; r++) acc[r] = 0u;
    path_neighbours<NR, LQ>(X, lane, q, P1pk, up, dn);
    path_cells<NR>(X, up, dn, Cp, acc, Ln, mn, P1pk);
    path_normalise<NR>(X, Ln, pixel_min<LQ>(mn), (uint32_t)s.P2);
    if (valid) store_bytes<NR, LQ>(vol + (size_t)xk * s.D, q, acc);
    // next pixel: slide the window when the byte phase wraps
    if (!DIR) {
      xk++;
      if (sh == 3) {
        al += 4;
#pragma unroll
        for (int k = 0; k < NW + 1; k++) Rw[k] = Rw[k + 1];
        Lw[0] = Lw[1]; Lw[1] = Lw[2];
        Rw[NW + 1] = myR[soff]; Lw[2] = myL[soff];
        soff++; wraps_left--;
      }
    } else {
      xk--;
      if (sh == 0) {
        al -= 4;
#pragma unroll
        for (int k = NW + 1; k > 0; k--) Rw[k] = Rw[k - 1];
        Lw[2] = Lw[1]; Lw[1] = Lw[0];
        Rw[0] = myR[soff]; Lw[0] = myL[soff];
        soff--; wraps_left--;
      }
    }
  }
}
template <int NR, int LQ>
__global__ void __launch_bounds__(256, NR <= 16 ? 4 : 2) k_sw_h(SwDev s, int n, const uint8_t* __restrict__ gm, uint8_t* __restrict__ vol0, uint8_t* __restrict__ vol1) {
  constexpr int NQ = LQ, PX = 64 / LQ;
  constexpr int DPL = 2 * NR, SPAN = HCH + (NQ - 1) * DPL / 4;
  __shared__ uint32_t sR[4][PX * SPAN], sL[4][PX * HCH];       // per wave: the staged dwords of its PX rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane / PX, p = lane & (PX - 1);
  const int y0 = (blockIdx.x * 4 + wave) * PX, frame = blockIdx.y, dir = blockIdx.z;
  if (y0 >= s.H) return;
  const int y = y0 + p;
  const bool valid = y < s.H;
  const int yc = min(y, s.H - 1);
  const uint8_t* gmL = gm + (size_t)frame * s.H * s.Wp + s.padl;
  const uint8_t* gmR = gm + (size_t)(n + frame) * s.H * s.Wp + s.padl;
  const size_t row_px = ((size_t)frame * s.H + yc) * s.W;
  if (dir == 0) h_sweep<NR, 0, LQ>(s, gmL, gmR, y0, vol0 + row_px * s.D, valid, lane, q, sR[wave], sL[wave]);
  else h_sweep<NR, 1, LQ>(s, gmL, gmR, y0, vol1 + row_px * s.D, valid, lane, q, sR[wave], sL[wave]);
}

// ---- the three paths of one vertical direction, sheared strips ----

// ---- the row sweeps: no workgroup barrier, no communication wave ----
// (The first form kept a block's strips in lock-step — one s_barrier per row: every row took as long as the slowest strip — and spent a
// quarter of the wave slots and of the register file on communication waves.)  Every wave computes, and a strip runs as
// far ahead of its left neighbour as a ring of RING rows in LDS allows:
//   * inside a block: strip w publishes its boundary columns of row y in ring[y mod RING][w] and then prog[w] = y + 1; strip w-1
//     polls prog[w] only when its cached copy is too old, reads the columns and acknowledges with cons[w] = y + 1 (back pressure:
//     row y + RING overwrites that slot).  LDS executes a wave's instructions in order, so data-then-counter needs no wait; the
//     fences are "local" ones (lgkmcnt only — an ordinary workgroup release would drain the volume stores every row);
//   * between blocks: the columns travel as SELF-VALIDATING dwords.  Path values are <= P2 <= 255, so bytes 1 and 3 of every packed
//     pair are free: they carry a 16-bit launch tag.  The producer (strip 0) stores without draining anything, the consumer (last
//     strip) loads its producer's row one row AHEAD of its use, checks the tags when it needs the row and only then — the row
//     was not complete yet — polls.  No progress flags, no store drains, no ordering assumptions between different addresses
//     (every aligned dword is its own message).  Tags change with every launch on a buffer (SweepBuffers::epoch); the buffer is
//     zeroed when it is allocated and whenever the 16-bit tag wraps, so a stale row can never carry the current tag.
//   * the right image's winners: each strip owns a row of LDS minima per quarter and flushes it itself.
// The ticket order (a block's producer holds the smaller ticket) makes the inter-block wait placement-independent as before.
DEV int lds_load_relaxed(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV void lds_store_relaxed(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// a piece of a row of boundary columns as it crosses between blocks: 16 or 8 bytes, write-through (aux 16 = sc1)
template <int PW> struct Piece;
template <> struct Piece<4> {
  typedef u32x4 T;
  DEV T load(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); }
  DEV void store(T v, __amdgpu_buffer_rsrc_t r, int off) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16); }
  DEV bool all_tagged(T v, uint32_t tag) { return ((v.x & 0xFF00FF00u) == tag) & ((v.y & 0xFF00FF00u) == tag) & ((v.z & 0xFF00FF00u) == tag) & ((v.w & 0xFF00FF00u) == tag); }
  DEV T zero() { return (T){0u, 0u, 0u, 0u}; }
};
template <> struct Piece<2> {
  typedef u32x2 T;
  DEV T load(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 16); }
  DEV void store(T v, __amdgpu_buffer_rsrc_t r, int off) { __builtin_amdgcn_raw_buffer_store_b64(v, r, off, 0, 16); }
  DEV bool all_tagged(T v, uint32_t tag) { return ((v.x & 0xFF00FF00u) == tag) & ((v.y & 0xFF00FF00u) == tag); }
  DEV T zero() { return (T){0u, 0u}; }
};
// Profiling switches of k_sw_w (results are then WRONG; attribution only): compiled in with -DJN_SGM_PROFILE (make ... EXTRA=-DJN_SGM_PROFILE),
// absent from the product build — a switch that is only tested at run time still keeps registers alive across the code it guards.
//   JN_SGM_DBG bits: 1 no wait for / load of the producer block's columns, 2 no right-image minima (LDS atomics + flush), 4 no volume loads /
//   stores, 16 no per-row input fetch, 32 no waiting on the neighbour strip's LDS counters
#ifdef JN_SGM_PROFILE
#define SW_DBG(bit) (s.dbg & (bit))
#else
#define SW_DBG(bit) false
#endif
DEV uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
// N ds_read_b128 from addrV into tv and N from addrM into tm in the lanes whose address is not 0xFFFFFFFF (the others are switched off
// with the exec mask around the reads and keep what their registers held: tv / tm are in/out operands), then the wait for them — ONE
// asm statement, so that the compiler never sees (and never copies or spills) the registers while the reads are in flight.  (s_nop 4: a VALU write of EXEC — the
// v_cmpx — must be five wait states away from the DPP instructions that follow the statement.)
template <int N>
DEV void lds_read_lanes(u32x4 (&tv)[N], u32x4 (&tm)[N], uint32_t addrV, uint32_t addrM);
template <>
DEV void lds_read_lanes<2>(u32x4 (&tv)[2], u32x4 (&tm)[2], uint32_t addrV, uint32_t addrM) {
  asm volatile("s_mov_b64 s[2:3], exec\n\tv_cmpx_ne_u32 -1, %5\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:16\n\t"
               "s_mov_b64 exec, s[2:3]\n\tv_cmpx_ne_u32 -1, %4\n\tds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\t"
               "s_mov_b64 exec, s[2:3]\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 4"
               : "+v"(tv[0]), "+v"(tv[1]), "+v"(tm[0]), "+v"(tm[1]) : "v"(addrV), "v"(addrM) : "s2", "s3", "vcc", "memory");
}
template <>
DEV void lds_read_lanes<4>(u32x4 (&tv)[4], u32x4 (&tm)[4], uint32_t addrV, uint32_t addrM) {
  asm volatile("s_mov_b64 s[2:3], exec\n\tv_cmpx_ne_u32 -1, %9\n\tds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:16\n\tds_read_b128 %6, %9 offset:32\n\tds_read_b128 %7, %9 offset:48\n\t"
               "s_mov_b64 exec, s[2:3]\n\tv_cmpx_ne_u32 -1, %8\n\tds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
               "s_mov_b64 exec, s[2:3]\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 4"
               : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]), "+v"(tm[0]), "+v"(tm[1]), "+v"(tm[2]), "+v"(tm[3]) : "v"(addrV), "v"(addrM) : "s2", "s3", "vcc", "memory");
}
template <>
DEV void lds_read_lanes<8>(u32x4 (&tv)[8], u32x4 (&tm)[8], uint32_t addrV, uint32_t addrM) {
  asm volatile("s_mov_b64 s[2:3], exec\n\tv_cmpx_ne_u32 -1, %17\n\tds_read_b128 %8, %17\n\tds_read_b128 %9, %17 offset:16\n\tds_read_b128 %10, %17 offset:32\n\tds_read_b128 %11, %17 offset:48\n\t"
               "ds_read_b128 %12, %17 offset:64\n\tds_read_b128 %13, %17 offset:80\n\tds_read_b128 %14, %17 offset:96\n\tds_read_b128 %15, %17 offset:112\n\t"
               "s_mov_b64 exec, s[2:3]\n\tv_cmpx_ne_u32 -1, %16\n\tds_read_b128 %0, %16\n\tds_read_b128 %1, %16 offset:16\n\tds_read_b128 %2, %16 offset:32\n\tds_read_b128 %3, %16 offset:48\n\t"
               "ds_read_b128 %4, %16 offset:64\n\tds_read_b128 %5, %16 offset:80\n\tds_read_b128 %6, %16 offset:96\n\tds_read_b128 %7, %16 offset:112\n\t"
               "s_mov_b64 exec, s[2:3]\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 4"
               : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]), "+v"(tv[4]), "+v"(tv[5]), "+v"(tv[6]), "+v"(tv[7]),
                 "+v"(tm[0]), "+v"(tm[1]), "+v"(tm[2]), "+v"(tm[3]), "+v"(tm[4]), "+v"(tm[5]), "+v"(tm[6]), "+v"(tm[7]) : "v"(addrV), "v"(addrM) : "s2", "s3", "vcc", "memory");
}

template <int NR, int NS, int RING, bool FINAL, bool WIDE, int LQ>
__global__ void __launch_bounds__(NS * 64, NR == 16 ? 3 : (NR == 32 && !FINAL) ? 2 : 1) k_sw_w(SwDev s, int n, int flip, const uint8_t* __restrict__ gm, uint8_t* __restrict__ volF,
                                                  const uint8_t* __restrict__ volH0, const uint8_t* __restrict__ volH1, uint32_t* __restrict__ gx,
                                                  uint32_t* __restrict__ ctr, uint32_t* __restrict__ gminR, uint32_t* __restrict__ dLp) {
  constexpr int NQ = LQ, PX = 64 / LQ;                         // lanes per pixel (4, or 8 for D = 256: 16 disparity pairs per lane either way), pixels per strip
  constexpr bool LATE_PROD = FINAL && LQ == 8;                 // where the last strip requests its producer block's columns (see there)
  constexpr int DPL = 2 * NR, SLOT = 3 * NQ * NR, BLK = NS * PX, MR = PX + DPL;
  constexpr int PW = LQ == 4 ? 4 : 2, NP = SLOT / PW, NG = (NP + 63) / 64;       // dwords per piece of a row of columns between blocks, pieces per row, pieces per lane
  typedef Piece<PW> Px;
  typedef typename Px::T piece_t;
  static_assert((RING & (RING - 1)) == 0, "ring depth is a power of two");
  __shared__ __attribute__((aligned(16))) uint32_t ring[RING][NS][SLOT];                    // boundary columns [row mod RING][strip][V0 | M0 | M1][quarter][NR]
  __shared__ __attribute__((aligned(16))) uint32_t nextblk[SLOT];                           // the next block's columns for the last strip (written and read by that wave only)
  __shared__ uint32_t minR[FINAL ? NS : 1][FINAL ? 2 : 1][FINAL ? NQ : 1][FINAL ? MR : 1];   // right-image winners of one row of one strip, per disparity quarter (rows alternate)
  __shared__ uint32_t minR_trash[FINAL ? NS : 1][FINAL ? NQ : 1][FINAL ? MR : 1];              // where lanes outside the image send theirs
  __shared__ int prog[NS], cons[NS];                           // rows published by strip w / rows of strip w's columns consumed by strip w-1
  __shared__ int s_ticket;
  extern __shared__ uint16_t sS[];                             // FINAL + sub-pixel: S of the block's pixels [BLK][D]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int W = s.W, H = s.H, D = s.D, NB = s.NB;
  const uint32_t P2pk = (uint32_t)s.P2 * 0x10001u;
  const uint32_t tagpk = (((uint32_t)s.epoch & 0xFFu) << 8) | (((uint32_t)s.epoch >> 8) << 24);
  if (tid == 0) s_ticket = (int)atomicAdd(ctr, 1u);
  for (int k = tid; k < RING * NS * SLOT; k += NS * 64) (&ring[0][0][0])[k] = P2pk;            // X = P2: a path that starts here
  if (FINAL) for (int k = tid; k < NS * 2 * NQ * MR; k += NS * 64) (&minR[0][0][0][0])[k] = 0xFFFFFFFFu;
  __syncthreads();
  const int ticket = __builtin_amdgcn_readfirstlane(s_ticket);     // wave-uniform, and known to be: everything derived from it (frame, block, rows) stays scalar
  // producers (larger j) hold the smaller tickets; the block index is the major order: all frames walk through their parallelogram in phase
  // and finish together (frame-major tickets were measured: 2 588 against 3 168 pairs/s — the last frames run alone at the end)
  const int j = NB - 1 - ticket / n, frame = ticket % n;
  const int x0 = s.xmin + BLK * j;                             // sheared origin of this block: x' in [x0, x0 + BLK)
  const int ybs = max(0, -(x0 + BLK - 1)), ybe = min(H - 1, W - 1 - x0);
  if (ybs > ybe) return;
  if (tid < NS) { prog[tid] = ybs; cons[tid] = ybs - 1; }       // rows < prog published (row ybs - 1 = the initial fill); rows < cons read by the left neighbour
  __syncthreads();                                             // the only barriers of the kernel: before the first row
  uint32_t* my_gx = gx + ((size_t)frame * NB + j) * H * (size_t)SLOT;
  // the producer block (j + 1) and the rows it works on
  const bool has_prod = j + 1 < NB;
  const int x0p = x0 + BLK;
  const int ybsp = max(0, -(x0p + BLK - 1)), ybep = min(H - 1, W - 1 - x0p);
  const uint32_t* p_gx = gx + ((size_t)frame * NB + j + 1) * H * (size_t)SLOT;
  const bool last = wave == NS - 1;
  // The columns cross between blocks in 16-BYTE pieces (round 5; they were single dwords): a write-through dword store is one fabric write
  // of its own and costs ~6x a 16-byte store's time per byte, an 8-byte one 2.7x (MI355X_MICROARCH.md, stores of each flavour).  Every dword
  // still carries its tag, so nothing is assumed about how a wider store becomes visible.  Lanes beyond the row's NP pieces: the buffer's range
  // check drops their stores and returns zeros to their loads (they count as valid).  (Eight lanes per pixel: 8-byte pieces — as many
  // registers as the dwords took; with 16-byte ones the final sweep spills a dozen more.)
  piece_t g[NG];                                               // last strip: the producer's row, loaded a row ahead of its use
#pragma unroll
  for (int k = 0; k < NG; k++) g[k] = Px::zero();
  auto gx_rsrc = [&](const uint32_t* row) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(row), 0, SLOT * 4, 0x00020000);
  };
  auto load_prod = [&](int yr, piece_t (&dst)[NG]) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t r = gx_rsrc(p_gx + (size_t)yr * SLOT);     // wave-uniform
    int ln = lane;
    asm volatile("" : "+v"(ln));                                // scalar base + lane offset, formed here: a per-lane 64-bit pointer kept across the loop is two registers the final sweep lacks
#pragma unroll
    for (int k = 0; k < NG; k++) dst[k] = Px::load(r, 4 * PW * (ln + 64 * k));
  };
  auto tags_ok = [&](const piece_t (&v)[NG]) __attribute__((always_inline)) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NG; k++) ok = ok && (Px::all_tagged(v[k], tagpk) || (NP % 64 != 0 && lane + 64 * k >= NP));
    return ok;
  };
  auto to_nextblk = [&](const piece_t (&v)[NG], uint32_t* nb) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NG; k++) {
      const int o = lane + 64 * k;
      if (NP % 64 == 0 || o < NP) *reinterpret_cast<piece_t*>(nb + PW * o) = v[k] & 0x00FF00FFu;
    }
  };
  const int q = lane / PX, p = lane & (PX - 1);
  const int xl = x0 + PX * wave + p;                           // this lane's sheared column
  const uint32_t P1pk = (uint32_t)s.P1 * 0x10001u;
  uint32_t V[NR], G[NR], M[NR];                                // X of the pixel this lane computed last, per path: vertical, own diagonal, other diagonal
#pragma unroll
  for (int r = 0; r < NR; r++) V[r] = G[r] = M[r] = P2pk;
  const size_t img_rows = (size_t)H * s.Wp;
  // A ROW'S INPUT BYTES travel through LDS.  The lanes of a strip read overlapping windows of the same ~150 bytes of the right row (and 19 of
  // the left one): loaded per lane into registers a row ahead — 17 loop-carried registers that the final sweep had no room for — they are now
  // fetched ONCE per wave (8 bytes per lane), two rows ahead, and read from LDS right before the costs.  A lane's window starts at byte
  // p + DPL q: any alignment, and a misaligned ds_read costs 64 cycles of the CU's LDS pipeline against 4.6 for an aligned one
  // (scripts/probes/lds_unaligned_probe.hip).  The fetching lanes therefore store FOUR copies of the row, shifted by 0 .. 3 bytes
  // (v_alignbyte of their two dwords), and a lane reads dword-aligned from copy p & 3.  Two rows of LDS per strip: the commit of row yb + 2
  // overwrites row yb's slot behind row yb's reads (LDS serves a wave's instructions in order).
  constexpr int RBYTES = PX - 1 + DPL * (NQ - 1) + 4 * (NR / 2 - 1) + 8;      // right-row bytes a strip touches
  constexpr int NRD = (RBYTES + 3) / 4, NLD = (PX + 6) / 4, TD = NRD + NLD, NSTG = (TD + 63) / 64, RS = (TD + 3) & ~3;
  __shared__ uint32_t rowbuf[NS][2][4][RS];                     // [strip][row parity][byte shift][dword]
  const int y0 = flip ? H - 1 - ybs : ybs;
  const long long row_step = (flip ? -(long long)s.Wp : (long long)s.Wp) + 1;
  const uint8_t* gsrc;                                          // left row of the strip's next fetch (wave-uniform; the right image lies n images further)
  uint32_t soff[NSTG];                                          // this lane's dword of a fetch, relative to gsrc
  {
    gsrc = gm + (size_t)frame * img_rows + (size_t)y0 * s.Wp + s.padl - 1 + (x0 + PX * wave + ybs);
    const uint32_t to_right = (uint32_t)((size_t)n * img_rows);
#pragma unroll
    for (int k = 0; k < NSTG; k++) { const int i = lane + 64 * k; soff[k] = i < NRD ? to_right + 4 * i : 4 * (min(i, TD - 1) - NRD); }
  }
  int next_row = ybs;                                           // the row the next fetch is for (rows beyond ybe fetch ybe again: unconditional loads)
  auto stage_load = [&](uint64_t (&d)[NSTG]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NSTG; k++) d[k] = load_u64_unaligned(gsrc + soff[k]);
    gsrc += next_row < ybe ? row_step : 0;
    next_row++;
  };
  auto stage_write = [&](int row, const uint64_t (&d)[NSTG]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NSTG; k++) {
      const uint32_t lo = (uint32_t)d[k], hi = (uint32_t)(d[k] >> 32);
      if (TD % 64 != 0 && lane + 64 * k >= TD) continue;        // (the last lanes fetched a copy of dword TD - 1)
      uint32_t* dst = &rowbuf[wave][row & 1][0][lane + 64 * k];
      dst[0] = lo;
      dst[RS] = __builtin_amdgcn_alignbyte(hi, lo, 1);
      dst[2 * RS] = __builtin_amdgcn_alignbyte(hi, lo, 2);
      dst[3 * RS] = __builtin_amdgcn_alignbyte(hi, lo, 3);
    }
  };
  struct RowIn { uint64_t ww[NR / 2]; uint32_t ref; };
  const int rd_at = (p & 3) * RS + (p >> 2) + (DPL / 4) * q;   // this lane's first dword inside a staged row: copy p & 3, dword-aligned
  auto read_row = [&](int row, RowIn& in_) __attribute__((always_inline)) {
    const uint32_t* rb = &rowbuf[wave][row & 1][0][0] + rd_at;
#pragma unroll
    for (int k = 0; k < NR / 2; k++) in_.ww[k] = (uint64_t)rb[k] | ((uint64_t)rb[k + 1] << 32);      // each pair on its own (ds_read2_b32): one v_mqsad operand, no re-pairing
    in_.ref = (&rowbuf[wave][row & 1][0][0])[(p & 3) * RS + NRD + (p >> 2)];
  };
  if (last && has_prod && ybs - 1 >= ybsp && ybs - 1 <= ybep) load_prod(ybs - 1, g);
  uint64_t stg[NSTG];                                           // the fetch in flight: row yb + 2 at the top of row yb
  {
    uint64_t d0[NSTG], d1[NSTG];
    stage_load(d0); stage_load(d1); stage_load(stg);
    stage_write(ybs, d0); stage_write(ybs + 1, d1);
  }
  // The volumes are addressed as ROW BUFFERS: a buffer resource over the row's W * D bytes (scalar registers, rebuilt per row) and one 32-bit
  // offset per lane.  Lanes outside the image have an offset outside the buffer (a negative column wraps to ~2^32): their loads return zeros
  // nobody uses and their stores are dropped — every instruction stays unconditional (the compiler can count what is in flight), no
  // clamping, and no 64-bit address arithmetic per lane (it cost ~15 vector instructions and 6 registers per row).
  // The final sweep's three stored volumes are kept as the 16-byte vectors they are loaded as: split into dwords they become separate
  // loop-carried values, the compiler gives some of them other registers at the top of the loop than the load writes, and the copy it then
  // needs waits for the load right behind it.
  constexpr int NVF = FINAL ? (WIDE ? NR / 4 : NR / 8) : 1, NVH = FINAL ? NR / 8 : 1;
  constexpr int FB = WIDE ? 2 : 1;                              // bytes per cell of the F volume
  u32x4 fF[NVF], fH0[NVH], fH1[NVH];
  auto row_rsrc = [&](const uint8_t* vol, int yb, int cell_bytes) __attribute__((always_inline)) {
    const int y = flip ? H - 1 - yb : yb;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(vol + ((size_t)frame * H + y) * W * D * cell_bytes), 0, W * D * cell_bytes, 0x00020000);
  };
  auto load_volumes = [&](int yb) __attribute__((always_inline)) {
    if constexpr (FINAL) {
      const __amdgpu_buffer_rsrc_t rF = row_rsrc(volF, yb, FB), r0 = row_rsrc(volH0, yb, 1), r1 = row_rsrc(volH1, yb, 1);
      const int off = (xl + yb) * D + 16 * q, offF = (xl + yb) * D * FB + 16 * q;
#pragma unroll
      for (int c = 0; c < NVF; c++) fF[c] = __builtin_amdgcn_raw_buffer_load_b128(rF, offF + 16 * NQ * c, 0, 0);
#pragma unroll
      for (int c = 0; c < NVH; c++) fH0[c] = __builtin_amdgcn_raw_buffer_load_b128(r0, off + 16 * NQ * c, 0, 0);
#pragma unroll
      for (int c = 0; c < NVH; c++) fH1[c] = __builtin_amdgcn_raw_buffer_load_b128(r1, off + 16 * NQ * c, 0, 0);
    }
  };
  load_volumes(ybs);
  // everything requested so far is complete before the loop starts (uses the compiler must wait for): the waits inside the loop are then
  // written for what a row leaves in flight, not for the prologue
#pragma unroll
  for (int k = 0; k < NSTG; k++) asm volatile("" : : "v"(stg[k]));
  if constexpr (FINAL) {
#pragma unroll
    for (int k = 0; k < NVF; k++) asm volatile("" : : "v"(fF[k]));
#pragma unroll
    for (int k = 0; k < NVH; k++) asm volatile("" : : "v"(fH0[k]), "v"(fH1[k]));
  }
  // flush one row of this strip's right-image minima (minR[wave][buf]) to the row's global minima
  auto flush_minima = [&](int yb, int buf) __attribute__((always_inline)) {
    if constexpr (FINAL) {
      const int y = flip ? H - 1 - yb : yb;
      uint32_t* grow = gminR + ((size_t)frame * H + y) * W;
      uint32_t* mrow = &minR[wave][buf][0][0];
      int ln = lane;
      asm volatile("" : "+v"(ln));                              // the index arithmetic below is redone per row: hoisted out of the loop it costs a dozen registers the kernel does not have
      constexpr int KF = (NQ * MR + 63) / 64;
      uint32_t kvs[KF];
#pragma unroll
      for (int k = 0; k < KF; k++) {                            // every read first: one LDS round trip for the row instead of one per 64 entries
        const int idx = ln + 64 * k;
        kvs[k] = ((NQ * MR) % 64 == 0 || idx < NQ * MR) ? mrow[idx] : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int k = 0; k < KF; k++) {
        const int idx = ln + 64 * k;
        const uint32_t kv = kvs[k];
        if (kv != 0xFFFFFFFFu) {
          mrow[idx] = 0xFFFFFFFFu;
          const int qq = idx / MR, ee = idx - qq * MR;
          const int xr = x0 + PX * wave + yb + ee + DPL * qq;
          if (xr >= 0 && xr < W) atomicMin(grow + xr, kv + (uint32_t)(DPL * qq));
        }
      }
    }
  };
  int known_p = ybs, known_c = ybs - 1;                        // cached prog[wave + 1] / cons[wave]
  // Order inside a row: everything that was loaded from memory was requested a whole row earlier, into registers that had just
  // been consumed (no second set of registers, and every wait counts only loads that are a row old):
  //   costs from the row's bytes -> request the next row's bytes;  [last strip: the producer's columns, requested a row ago]
  //   -> neighbour columns -> the three paths -> publish -> [final: S from the stored volumes -> request the next row's volumes -> winners]
  for (int yb = ybs; yb <= ybe; yb++) {
    const int y = flip ? H - 1 - yb : yb;
    const int xk = xl + yb;
    const bool in = xk >= 0 && xk < W;
    uint32_t Cp[NR], acc[NR];
    {
      RowIn cur;
      read_row(yb, cur);                                       // LDS reads of the row's bytes ...
      if constexpr (FINAL) {                                   // ... in flight while the previous row's minima are flushed (their atomics were served long ago)
        __builtin_amdgcn_sched_barrier(0);
        if (yb > ybs && !SW_DBG(2)) flush_minima(yb - 1, (yb - 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      costs64<NR>(cur.ww, cur.ref & 0x00FFFFFFu, P2pk, Cp);
    }
#pragma unroll
    for (int r = 0; r < NR; r++) asm volatile("" : "+v"(Cp[r]) : : "memory");   // the costs are computed HERE (they would otherwise sink to their first
    __builtin_amdgcn_sched_barrier(0);                         // use, past the loads that reuse their input registers — which then get copied)
    if (!SW_DBG(16)) { stage_write(yb + 2, stg); stage_load(stg); }   // row yb + 2 (fetched during row yb - 1) into the ring; request row yb + 3
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!FINAL) {
#pragma unroll
      for (int r = 0; r < NR; r++) acc[r] = 0u;
    }
    // ---- last strip: the producer block's columns of row yb - 1 ----
    if (last) {
      const int yr = yb - 1;
      if (has_prod && yr >= ybsp && yr <= ybep && !SW_DBG(1)) {
        // g was loaded for exactly this row (before the loop or during the previous row).  The usual case — every tag is this launch's —
        // has its own code path, so that its wait counts only what is older than g; the retry loop (the producer has not written the
        // whole row yet) reloads into other registers.
        if (__builtin_amdgcn_ballot_w64(!tags_ok(g)) == 0ull) {
          to_nextblk(g, nextblk);
        } else {
          piece_t v[NG];
          do {
            __builtin_amdgcn_s_sleep(8);
            load_prod(yr, v);
          } while (__builtin_amdgcn_ballot_w64(!tags_ok(v)) != 0ull);
          to_nextblk(v, nextblk);
        }
      } else {
#pragma unroll
        for (int k = 0; k < (SLOT + 63) / 64; k++) { const int o = lane + 64 * k; if (SLOT % 64 == 0 || o < SLOT) nextblk[o] = P2pk; }
      }
      if constexpr (!LATE_PROD) { if (has_prod && yb >= ybsp && yb <= ybep && yb < ybe && !SW_DBG(1)) load_prod(yb, g); }  // the next row's, speculatively: checked when it is needed
    }
    // ---- the right neighbour's columns of row yb - 1 ----
    const uint32_t* e;
    if (!last) {
      while (known_p < yb && !SW_DBG(32)) {
        known_p = __builtin_amdgcn_readfirstlane(lds_load_relaxed(&prog[wave + 1]));
        if (known_p < yb) __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      e = &ring[(yb - 1) & (RING - 1)][wave + 1][0];
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      e = nextblk;
    }
    // V moves one column to the left, M two (DPP row shifts; the last lanes keep their own value for the moment).  What enters from
    // the right neighbour — its column 0 of V into lane 15, its columns 0 and 1 of M into lanes 14 and 15 — is read from LDS by those
    // lanes only, straight into the shifted registers: 2 NR shifts (the first form shifted M twice: 3 NR), 8 instead of 12 LDS reads
    // and no temporaries.  The reads and the wait for them are ONE asm statement with the registers as in/out operands: the other
    // lanes keep their value (which C++ cannot say without 48 extra moves per row), and the compiler never sees the registers while
    // the reads are in flight.
    {
#pragma unroll
      for (int r = 0; r < NR; r++) {
        V[r] = (uint32_t)__builtin_amdgcn_update_dpp((int)V[r], (int)V[r], 0x101, 0xf, 0xf, false);   // row_shl:1
        M[r] = (uint32_t)__builtin_amdgcn_update_dpp((int)M[r], (int)M[r], 0x102, 0xf, 0xf, false);   // row_shl:2
      }
      u32x4 tv[NR / 4], tm[NR / 4];
#pragma unroll
      for (int k = 0; k < NR / 4; k++) { tv[k] = (u32x4){V[4 * k], V[4 * k + 1], V[4 * k + 2], V[4 * k + 3]}; tm[k] = (u32x4){M[4 * k], M[4 * k + 1], M[4 * k + 2], M[4 * k + 3]}; }
      const uint32_t aV = lds_addr(e + (0 * NQ + q) * NR), aM = lds_addr(e + ((p == PX - 2 ? 1 : 2) * NQ + q) * NR);
      lds_read_lanes<NR / 4>(tv, tm, p == PX - 1 ? aV : 0xFFFFFFFFu, p >= PX - 2 ? aM : 0xFFFFFFFFu);
#pragma unroll
      for (int k = 0; k < NR / 4; k++) {
        V[4 * k] = tv[k].x; V[4 * k + 1] = tv[k].y; V[4 * k + 2] = tv[k].z; V[4 * k + 3] = tv[k].w;
        M[4 * k] = tm[k].x; M[4 * k + 1] = tm[k].y; M[4 * k + 2] = tm[k].z; M[4 * k + 3] = tm[k].w;
      }
      if (!last) {                                             // LDS serves a wave's instructions in order: the reads above are done
        if (lane == 0) lds_store_relaxed(&cons[wave + 1], yb);
      }
    }
    if constexpr (FINAL) {
      // S starts as 8 (C + P2) - (the five stored paths) and the three upward paths subtract their Y from it.  The stored volumes — requested
      // after the previous row's paths — are consumed HERE, before the cells: their 24 registers are free while the cells hold their
      // temporaries (the kernel then fits three waves per SIMD), at the price of a shorter lead for those loads.
#pragma unroll
      for (int k = 0; k < NR / 2; k++) {
        uint32_t a, b;
        const uint32_t h0 = fH0[k >> 2][k & 3], h1 = fH1[k >> 2][k & 3];
        if constexpr (WIDE) { a = pk_add(pk_add(fF[(2 * k) >> 2][(2 * k) & 3], unpack_lo(h0)), unpack_lo(h1)); b = pk_add(pk_add(fF[(2 * k + 1) >> 2][(2 * k + 1) & 3], unpack_hi(h0)), unpack_hi(h1)); }
        else { const uint32_t hb = h0 + h1, ff = fF[k >> 2][k & 3];    // bytes <= 2 P2 <= 170: no carry between bytes
               a = pk_add(unpack_lo(ff), unpack_lo(hb)); b = pk_add(unpack_hi(ff), unpack_hi(hb)); }
        acc[2 * k] = pk_sub(pk_shl3(Cp[2 * k]), a);
        acc[2 * k + 1] = pk_sub(pk_shl3(Cp[2 * k + 1]), b);
      }
#pragma unroll
      for (int r = 0; r < NR; r++) asm volatile("" : "+v"(acc[r]));
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      uint32_t upV, dnV, upG, dnG, upM, dnM, mn, Ln[NR];
      path_neighbours<NR, LQ>(V, lane, q, P1pk, upV, dnV);
      path_neighbours<NR, LQ>(G, lane, q, P1pk, upG, dnG);
      path_neighbours<NR, LQ>(M, lane, q, P1pk, upM, dnM);
      path_cells<NR, FINAL>(V, upV, dnV, Cp, acc, Ln, mn, P1pk);
      path_normalise<NR>(V, Ln, pixel_min<LQ>(mn), (uint32_t)s.P2);
      path_cells<NR, FINAL>(G, upG, dnG, Cp, acc, Ln, mn, P1pk);
      path_normalise<NR>(G, Ln, pixel_min<LQ>(mn), (uint32_t)s.P2);
      path_cells<NR, FINAL>(M, upM, dnM, Cp, acc, Ln, mn, P1pk);
      path_normalise<NR>(M, Ln, pixel_min<LQ>(mn), (uint32_t)s.P2);
    }
    if constexpr (FINAL) {
      __builtin_amdgcn_sched_barrier(0);
      if (!SW_DBG(4)) load_volumes(min(yb + 1, ybe));   // the next row's volumes: in flight during the winners, the publishing and the next row's costs
      __builtin_amdgcn_sched_barrier(0);
    }
    if (__builtin_amdgcn_ballot_w64(!in)) {                    // a strip crossing the image border: pixels outside carry Lq = 0 (a path entering the image starts with L = C)
#pragma unroll
      for (int r = 0; r < NR; r++) { V[r] = in ? V[r] : P2pk; G[r] = in ? G[r] : P2pk; M[r] = in ? M[r] : P2pk; }
    }
    if constexpr (LATE_PROD) {
      // (final sweep with eight lanes per pixel: six registers of producer columns do not fit next to the cells' temporaries — held across the
      // cells they are spilled, and a spill right behind the load is a synchronous wait.  Requested behind the cells instead, defined on every
      // path so that no old value stays alive around the loop.)
      if (last && has_prod && yb >= ybsp && yb <= ybep && yb < ybe && !SW_DBG(1)) load_prod(yb, g);
      else {
#pragma unroll
        for (int k = 0; k < NG; k++) g[k] = Px::zero();
      }
    }
    // ---- publish this strip's first two columns of row yb ----
    if (wave > 0 || j > 0) {
      if (wave > 0) {
        while (known_c < yb - RING + 1 && !SW_DBG(32)) {                      // the slot still holds row yb - RING until the left neighbour has read it
          known_c = __builtin_amdgcn_readfirstlane(lds_load_relaxed(&cons[wave]));
          if (known_c < yb - RING + 1) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      }
      uint32_t* o = &ring[yb & (RING - 1)][wave][0];
      if (p == 0) {
#pragma unroll
        for (int r = 0; r < NR; r++) { o[(0 * NQ + q) * NR + r] = V[r]; o[(1 * NQ + q) * NR + r] = M[r]; }
      }
      if (p == 1) {
#pragma unroll
        for (int r = 0; r < NR; r++) o[(2 * NQ + q) * NR + r] = M[r];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      if (wave > 0) {
        if (lane == 0) lds_store_relaxed(&prog[wave], yb + 1);
      } else {                                                 // strip 0: to the next block through memory, tagged
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        const __amdgpu_buffer_rsrc_t rG = gx_rsrc(my_gx + (size_t)yb * SLOT);
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int k = 0; k < NG; k++) {
          const int oo = ln + 64 * k;                           // (lanes beyond the row read a valid piece again; their store is dropped)
          const piece_t v = *reinterpret_cast<const piece_t*>(o + PW * min(oo, NP - 1)) | tagpk;
          Px::store(v, rG, 4 * PW * oo);
        }
      }
    }
    if (!last && yb < ybe) known_p = __builtin_amdgcn_readfirstlane(lds_load_relaxed(&prog[wave + 1]));   // for the next row: usually already far enough
    if constexpr (!FINAL) {
      // pixels outside the image store into the slack behind the volume: an unconditional store keeps the next row's wait for its input
      // bytes from also waiting for these stores (the compiler can then count them)
      if (!SW_DBG(4)) {
        const __amdgpu_buffer_rsrc_t rF = row_rsrc(volF, yb, FB);
        const int off = xk * D * FB + 16 * q;
        if constexpr (WIDE) {
#pragma unroll
          for (int c = 0; c < NR / 4; c++) __builtin_amdgcn_raw_buffer_store_b128((u32x4){acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]}, rF, off + 16 * NQ * c, 0, 0);
        } else {
#pragma unroll
          for (int c = 0; c < NR / 8; c++)
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){pack4(acc[8 * c], acc[8 * c + 1]), pack4(acc[8 * c + 2], acc[8 * c + 3]), pack4(acc[8 * c + 4], acc[8 * c + 5]),
                                                           pack4(acc[8 * c + 6], acc[8 * c + 7])}, rF, off + 16 * NQ * c, 0, 0);
        }
      }
    } else {
      const uint32_t (&S)[NR] = acc;                           // S = 8 (C + P2) - (the three upward Y + the stored five)
      uint32_t key = 0xFFFFFFFFu;
      // lanes outside the image aim their minima at a trash row: the atomics are unconditional and can be issued between the instructions
      // that build the keys (32 of them back to back fill the LDS queue and stall the wave)
      uint32_t* mr = in ? &minR[wave][yb & 1][q][p] : &minR_trash[wave][q][p];
#pragma unroll
      for (int r = 0; r < NR; r++) {
        const uint32_t klo = (S[r] << 16) | (uint32_t)r, khi = (S[r] & 0xFFFF0000u) | (uint32_t)(r + NR);
        if (!SW_DBG(2)) { atomicMin(mr + r, klo); atomicMin(mr + r + NR, khi); }
        key = min(min(key, klo), khi);
      }
      key = in ? key + (uint32_t)(DPL * q) : 0xFFFFFFFFu;
      {
        if constexpr (LQ == 8) key = min(key, (uint32_t)__builtin_amdgcn_update_dpp((int)key, (int)key, 0x128, 0xf, 0xf, false));   // row_ror:8
        const auto a = __builtin_amdgcn_permlane16_swap(key, key, false, false);
        key = min(a[0], a[1]);
        const auto b = __builtin_amdgcn_permlane32_swap(key, key, false, false);
        key = min(b[0], b[1]);
      }
      const int d = (int)(key & 0xFFFFu);
      int d16 = 16 * d;
      if (s.subpixel) {
        uint32_t* my = reinterpret_cast<uint32_t*>(sS + ((size_t)(PX * wave + p) * D + DPL * q));
#pragma unroll
        for (int r = 0; r < NR; r += 2) {                      // S in disparity order: low halves are j = r, high halves j = r + NR
          my[r / 2] = __builtin_amdgcn_perm(S[r + 1], S[r], 0x05040100u);
          my[(r + NR) / 2] = __builtin_amdgcn_perm(S[r + 1], S[r], 0x07060302u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (in && q == 0 && d > 0 && d < D - 1) {
          const uint16_t* ps = sS + (size_t)(PX * wave + p) * D;
          const int sm = ps[d - 1], sc = ps[d], sp = ps[d + 1];
          const int den = max(sm + sp - 2 * sc, 1);
          d16 = 16 * d + (16 * (sm - sp) + den) / (2 * den);
        }
      }
      // unconditional (the other lanes write into the slack behind the array): the next row's first wait can then count it
      {
        const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(dLp + ((size_t)frame * H + y) * W, 0, W * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32((uint32_t)d | ((uint32_t)(uint16_t)d16 << 16), rD, q == 0 ? xk * 4 : -1, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (FINAL) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
    flush_minima(ybe, ybe & 1);
  }
}

// ---- L/R check: the left winner survives if the right image's winner at x - d agrees ----
__global__ void __launch_bounds__(256) k_sw_lr(SwDev s, int n, const uint32_t* __restrict__ dLp, const uint32_t* __restrict__ gminR, int16_t* __restrict__ disp) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, frame = blockIdx.z;
  if (x >= s.W) return;
  const int xk = s.W - 1 - x;
  const size_t row = ((size_t)frame * s.H + y) * s.W;
  const uint32_t e = dLp[row + xk];
  const int d = (int)(e & 0xFFFFu);
  bool ok = true;
  if (s.lr >= 0) {
    const int xr = xk + d;                                     // x - d >= 0  <=>  x_k + d <= W - 1
    ok = xr < s.W && abs(d - (int)(gminR[row + min(xr, s.W - 1)] & 0xFFFFu)) <= s.lr;
  }
  const int scale = s.subpixel ? 16 : 1;
  disp[row + x] = (int16_t)(ok ? (s.subpixel ? (int)(int16_t)(e >> 16) : d) : -scale);
}

}  // namespace jnav_sgm

// ---------------------------------------------------------------- host side ----------------------------------------------------------------
namespace jnav_sgm {

// strips (= computing waves) per workgroup of the row sweeps: 4 (JN_SGM_NS=2 or 8 for A/B; D = 256 always 4)
static int strips_for(int D) {
  const char* e = JN_HOOK_ENV("JN_SGM_NS");
  const int v = e ? atoi(e) : 4;
  return (D != 256 && (v == 2 || v == 8)) ? v : 4;
}

// Lanes per pixel.  D = 256 with four lanes per pixel means 32 disparity pairs per lane and path: 96 registers of path state, ~320 with the
// temporaries — one wave per SIMD, and one wave issues a vector instruction every ~8 cycles at best (scripts/probes/dep_chain_probe.hip): half
// the SIMD's rate.  With EIGHT lanes per pixel (strips of 8 pixels) a lane holds 16 pairs as at D = 128: the same code, three waves per SIMD.
// JN_SGM_LQ=4 keeps the four-lane layout for A/B.
static int lanes_per_pixel(int D) {
  if (D != 256) return 4;
  const char* e = JN_HOOK_ENV("JN_SGM_LQ");
  return e && atoi(e) == 4 ? 4 : 8;
}

void sweep_geometry(int W, int H, int D, int P1, int P2, int cap, int lr, int subpixel, SwDev* s, SweepSizes* z, int max_batch) {
  s->W = W; s->H = H; s->D = D; s->P1 = P1; s->P2 = P2; s->cap = cap; s->lr = lr; s->subpixel = subpixel ? 1 : 0;
  s->epoch = 0;
  const int BLK = strips_for(D) * (64 / lanes_per_pixel(D));
  s->padl = BLK + 32; s->Wp = ((s->padl + W + D + BLK + 64) + 15) / 16 * 16;   // a block's width of padding: lanes outside the image read plain bytes
  s->xmin = -(H - 1);
  s->NB = (W + H - 1 + BLK - 1) / BLK;
  s->wide = 3 * P2 > 255 ? 1 : 0;
  s->dbg = JN_HOOK_ENV("JN_SGM_DBG") ? atoi(JN_HOOK_ENV("JN_SGM_DBG")) : 0;
  const size_t px = (size_t)W * H;
  z->gm = (size_t)2 * max_batch * H * s->Wp + 256;
  z->vol = (size_t)max_batch * px * D + 4096;                   // one byte volume (+ slack: where k_sw_w's lanes outside the image store); the F volume is twice that when wide
  z->gx = (size_t)max_batch * s->NB * H * (3 * NQ * (D / 8)) * sizeof(uint32_t);
  z->flags = 16 * sizeof(uint32_t);                             // the two sweeps' ticket counters
  z->minr = (size_t)max_batch * px * sizeof(uint32_t);
  z->dl = ((size_t)max_batch * px + 64) * sizeof(uint32_t);     // + a slack row: where k_sw_w's lanes without a pixel store
}

// k_sw_w: every launch on the buffer gets the next 16-bit tag; when the tag wraps the buffer is zeroed (tag 0 is never used), so
// that rows an earlier, larger batch left behind can never carry the current tag
template <int NR, int NS, int LQ>
static hipError_t launch_w(SwDev s, int n, bool final, hipStream_t st, SweepBuffers& b) {
  constexpr int RING = (NR <= 16 && LQ == 4) ? 8 : 4, PX = 64 / LQ;     // rows of boundary columns a strip may run ahead of its left neighbour (LDS: 3 x 4 NR x lanes-per-pixel dwords per strip and row)
  const dim3 grid((unsigned)(n * s.NB)), block(NS * 64);
  const size_t dyn = final && s.subpixel ? (size_t)NS * PX * s.D * sizeof(uint16_t) : 0;
  uint32_t* ctr = b.flags + (final ? 1 : 0);                    // the sweep's ticket counter (both zeroed by run_all's one memset)
  if (++b.epoch > 0xFFFFu) {
    hipError_t e = hipMemsetAsync(b.gx, 0, b.gx_bytes, st);     // the WHOLE buffer: frames a smaller batch does not touch keep older tags
    if (e != hipSuccess) return e;
    b.epoch = 1;
  }
  s.epoch = (int)b.epoch;
#define JN_SW_W(FINAL, WIDE)                                                                                                            \
  do {                                                                                                                                  \
    if (dyn > 32 * 1024) {                                                                                                              \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sw_w<NR, NS, RING, FINAL, WIDE, LQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
      if (e != hipSuccess) return e;                                                                                                    \
    }                                                                                                                                   \
    hipLaunchKernelGGL((k_sw_w<NR, NS, RING, FINAL, WIDE, LQ>), grid, block, dyn, st, s, n, FINAL ? 1 : 0, b.gm, b.volF, b.volH0, b.volH1, b.gx, ctr, b.minr, b.dl); \
  } while (0)
  if (final) { if (s.wide) JN_SW_W(true, true); else JN_SW_W(true, false); }
  else { if (s.wide) JN_SW_W(false, true); else JN_SW_W(false, false); }
#undef JN_SW_W
  return hipGetLastError();
}

template <int NR, int NS, int LQ = 4>
static hipError_t run_all(const SwDev& s, int n, const uint8_t* dI1, const uint8_t* dI2, int pitch, long long stride, int16_t* dDisp, hipStream_t st,
                          SweepBuffers& b, hipEvent_t* ev, bool side_overlap, bool lr_kernel) {
  hipError_t e;
  const size_t px = (size_t)s.W * s.H;
  if ((e = hipEventRecord(ev[0], st)) != hipSuccess) return e;
#ifdef JN_SGM_PROFILE
  static const int exp_ = JN_HOOK_ENV("JN_SGM_EXP") ? atoi(JN_HOOK_ENV("JN_SGM_EXP")) : 0;   // attribution only (results WRONG): 1 no prefilter, 2 no L/R kernel, 8 no minima memset
  if (!(exp_ & 1))
#endif
  hipLaunchKernelGGL(k_sw_prefilter, dim3((s.Wp + 255) / 256, (s.H + 4 * kPreRows - 1) / (4 * kPreRows), 2 * n), dim3(256), 0, st, s, dI1, dI2, pitch, stride, n, b.gm);
#ifdef JN_SGM_PROFILE
  if (!(exp_ & 8))
#endif
  if ((e = hipMemsetAsync(b.minr, 0xFF, (size_t)n * px * sizeof(uint32_t), st)) != hipSuccess) return e;
  if ((e = hipEventRecord(ev[1], st)) != hipSuccess) return e;
  constexpr int PXL = 64 / LQ;                                  // image rows per wave of the horizontal sweep = pixels per strip of the row sweeps
  // the horizontal sweep on the side stream, next to the downward sweep (JN_SGM_OVERLAP=0: one after the other on `st`, for A/B)
  static const int overlap_env = JN_HOOK_ENV("JN_SGM_OVERLAP") ? atoi(JN_HOOK_ENV("JN_SGM_OVERLAP")) : -1;
  const bool overlap = overlap_env >= 0 ? overlap_env != 0 : side_overlap;      // default: on for a lone synchronous batch, off when batches are pipelined over slots (the other slots fill the GPU; measured neutral to -3 % there)
  hipStream_t hs = st;
  if (overlap) {
    if (!b.side) {
      if ((e = hipStreamCreateWithFlags(&b.side, hipStreamNonBlocking)) != hipSuccess) return e;
      if ((e = hipEventCreateWithFlags(&b.ev_fork, hipEventDisableTiming)) != hipSuccess) return e;
      if ((e = hipEventCreateWithFlags(&b.ev_join, hipEventDisableTiming)) != hipSuccess) return e;
    }
    if ((e = hipEventRecord(b.ev_fork, st)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(b.side, b.ev_fork, 0)) != hipSuccess) return e;
    hs = b.side;
  }
  hipLaunchKernelGGL((k_sw_h<NR, LQ>), dim3((s.H + 4 * PXL - 1) / (4 * PXL), n, 2), dim3(256), 0, hs, s, n, b.gm, b.volH0, b.volH1);
  if (overlap && (e = hipEventRecord(b.ev_join, b.side)) != hipSuccess) return e;
  auto sweep = [&](bool final) -> hipError_t { return launch_w<NR, NS, LQ>(s, n, final, st, b); };
  if ((e = hipMemsetAsync(b.flags, 0, 2 * sizeof(uint32_t), st)) != hipSuccess) return e;
  if ((e = sweep(false)) != hipSuccess) return e;
  if (overlap && (e = hipStreamWaitEvent(st, b.ev_join, 0)) != hipSuccess) return e;   // the final sweep reads the horizontal volumes
  if ((e = hipEventRecord(ev[2], st)) != hipSuccess) return e;
  if ((e = sweep(true)) != hipSuccess) return e;
#ifdef JN_SGM_PROFILE
  if (!(exp_ & 2))
#endif
  if (lr_kernel) hipLaunchKernelGGL(k_sw_lr, dim3((s.W + 255) / 256, s.H, n), dim3(256), 0, st, s, n, b.dl, b.minr, dDisp);
  if ((e = hipEventRecord(ev[3], st)) != hipSuccess) return e;
  return hipGetLastError();
}

void sweep_release(SweepBuffers& b) {
  if (b.side) { hipStreamSynchronize(b.side); hipStreamDestroy(b.side); b.side = nullptr; }
  if (b.ev_fork) { hipEventDestroy(b.ev_fork); b.ev_fork = nullptr; }
  if (b.ev_join) { hipEventDestroy(b.ev_join); b.ev_join = nullptr; }
}

hipError_t sweep_run(const SwDev& s, int n, const uint8_t* dI1, const uint8_t* dI2, int pitch, long long stride, int16_t* dDisp, hipStream_t st,
                     SweepBuffers& b, hipEvent_t* ev, bool side_overlap, bool lr_kernel) {
  const int lq = lanes_per_pixel(s.D);
  const int ns = (s.padl - 32) / (64 / lq);                    // strips per block, as sweep_geometry() chose them (padl = BLK + 32)
#define JN_RUN(NR, NS) run_all<NR, NS>(s, n, dI1, dI2, pitch, stride, dDisp, st, b, ev, side_overlap, lr_kernel)
  if (s.D == 64) return ns == 2 ? JN_RUN(8, 2) : ns == 8 ? JN_RUN(8, 8) : JN_RUN(8, 4);
  if (s.D == 128) return ns == 2 ? JN_RUN(16, 2) : ns == 8 ? JN_RUN(16, 8) : JN_RUN(16, 4);
  if (lq == 8) return run_all<16, 4, 8>(s, n, dI1, dI2, pitch, stride, dDisp, st, b, ev, side_overlap, lr_kernel);     // D = 256: eight lanes per pixel, 16 pairs per lane
  return JN_RUN(32, 4);
#undef JN_RUN
}

}  // namespace jnav_sgm
