// bm_mfma.hip — the block matcher's sum-of-squared-differences cost (include/jn_bm.h, JN_BM_COST_SSD) as a banded int8 contraction on the
// matrix cores of gfx950.  Product code.  No reference counterpart (like the whole block-matching mode): the definition is in jn_bm.h, the
// scalar restatement it is compared with bit for bit lives outside the product (the checker).
//
// The contraction.  With g in [0, 62] (the prefiltered images) and P_I(p, y) the (2r+1)^2 patch of image I around column p of row y,
//     C(x, y, d) = sum (gL - gR)^2 = |P_L(x)|^2 + |P_R(u)|^2 - 2 P_L(x) . P_R(u),      u = x - d,
// so for one image row the costs of ALL (x, u) pairs are a product of two patch matrices, of which only the band 0 <= x - u < D is wanted:
// a banded GEMM with K = (2r+1)^2 int8 terms.  It is never formed with that K.  The window sum is separable in y: a wave walks down a band
// of rows with the cross terms of its (x, u) pairs as RUNNING sums in the MFMA accumulators,
//     X(x, u, y) = X(x, u, y-1) + sum_i L(x+i, y+r) R(u+i, y+r) - sum_i L(x+i, y-r-1) R(u+i, y-r-1),
// and BOTH row terms go into ONE v_mfma_i32_32x32x32_i8: the K = 32 of the instruction holds the 2r+1 taps of the row that enters (k < 16)
// and the 2r+1 taps of the row that leaves (k >= 16), with the sign of one operand carrying the subtraction.  One MFMA per 32 x 32 tile of
// (u, x) pairs and row: 32 768 multiply-adds, of which (2r+1) * 2 * 1024 are the algorithm's — the price of not holding 2r+1 row products.
//
// Layout of a wave (left-referenced pass; the right-referenced one swaps the images' roles).  B operand: the 32 columns x of the wave's
// tile; A operand: 32 candidate columns u of the other image, NT = D / 32 + 1 tiles of them cover the band.  The accumulator of a tile has
// a lane's column x fixed and 16 of the tile's rows u in its 16 registers, so the minimum over u — the winner-takes-all — is IN-LANE, one
// lane swap folds the two half-waves.  Keys: with QL(x) = 256 |P_L(x)|^2 + x and QR(u) = 256 |P_R(u)|^2 - u (k_bmq_box),
//     256 C + d = QL(x) + QR(u) - 512 X:   one v_lshl_add_u32 per pair on the accumulator (which holds -X) and half a v_min3_i32;
// QL(x) is a per-lane constant added after the minimum, QR(u) comes from LDS as 16-byte broadcast reads.  Of the first and the last tile
// only the pairs with 0 <= d < D count: for every row index exactly one of the two tiles is valid, a v_cndmask picks it.
// Operands are built from wave-private rows in LDS (a ring of 2r+2 rows per image, the A-side image also negated): three aligned dwords,
// two v_alignbyte and a mask per operand.  No workgroup barrier: waves are independent.
// Arithmetic per row and 32 columns (D = 128): 5 MFMAs (5 x 32 cycles), ~150 vector instructions; the pass is bound by vector issue, not by
// the matrix cores — DESIGN.md has the measured numbers next to the v_qsad (SAD) kernel's.
#include <hip/hip_runtime.h>
#include "hooks.h"
#include <stdint.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "bm_mfma.h"
#include "prefilter.h"

namespace jnav_bmq {

#define DEV static __device__ __forceinline__

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ---- prefilter: g = clamp(Sobel_x, -cap, cap) + cap in [0, 2 cap] (an int8 the matrix cores take as it is), replicated borders, rows
// padded by padx columns: no window ever needs a clamp in x ----
__global__ void __launch_bounds__(256) k_bmq_prefilter(QDev s, const uint8_t* __restrict__ I1, const uint8_t* __restrict__ I2, int pitch, long long stride,
                                                       int n, uint8_t* __restrict__ g) {
  const int xp = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y, img = blockIdx.z;      // four padded columns per thread (Wp is a multiple of 16)
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
  const uint32_t c = (uint32_t)s.cap;
  *reinterpret_cast<uint32_t*>(g + ((size_t)img * s.H + y) * s.Wp + xp) =
      ((uint32_t)v[0] + c) | (((uint32_t)v[1] + c) << 8) | (((uint32_t)v[2] + c) << 16) | (((uint32_t)v[3] + c) << 24);
}

// bytes [sh, sh + taps) of the 12 bytes d0 d1 d2, as three dwords with everything beyond `taps` zero
DEV void window(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t sh, int taps, uint32_t& w0, uint32_t& w1, uint32_t& w2) {
  w0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
  w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
  w2 = d2 >> (8 * sh);
  if (taps <= 4) { w0 &= taps == 4 ? 0xFFFFFFFFu : ((1u << (8 * taps)) - 1u); w1 = 0; w2 = 0; }
  else if (taps <= 8) { w1 &= taps == 8 ? 0xFFFFFFFFu : ((1u << (8 * (taps - 4))) - 1u); w2 = 0; }
  else w2 &= (1u << (8 * (taps - 8))) - 1u;
}

// ---- |P(p, y)|^2 of every padded column and the two halves of the keys: Q[img][y][c] = 256 |P|^2 + x (left images, img < n) or
// 256 |P|^2 - x (right images), x = c - padx.  A block takes 256 columns x BOXBAND rows: row sums of squares (v_dot4 on the window's
// dwords) into LDS, then the vertical window as a running sum. ----
constexpr int kBoxBand = 32;
template <int R>
__global__ void __launch_bounds__(256) k_bmq_box(QDev s, int n, const uint8_t* __restrict__ g, int32_t* __restrict__ Q) {
  constexpr int TAPS = 2 * R + 1, ROWS = kBoxBand + 2 * R;
  __shared__ int hs[ROWS][256];
  const int c = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * kBoxBand, img = blockIdx.z;
  const int cc = min(c, s.Wp - 1);
  const int start = min(max(cc - R, 0), s.Wp - 12);            // first byte of the window (the outermost padding columns are all replicas: clamping there changes nothing)
  const int al = start & ~3, sh = start & 3;
  const uint8_t* gi = g + (size_t)img * s.H * s.Wp;
  for (int t = 0; t < ROWS; t++) {
    const int yy = min(max(y0 - R + t, 0), s.H - 1);
    const uint32_t* p = reinterpret_cast<const uint32_t*>(gi + (size_t)yy * s.Wp + al);
    uint32_t w0, w1, w2;
    window(p[0], p[1], p[2], sh, TAPS, w0, w1, w2);
    hs[t][threadIdx.x] = (int)__builtin_amdgcn_udot4(w0, w0, __builtin_amdgcn_udot4(w1, w1, __builtin_amdgcn_udot4(w2, w2, 0u, false), false), false);
  }
  // (each thread reads only what it wrote: no barrier)
  int sum = 0;
#pragma unroll
  for (int t = 0; t < 2 * R; t++) sum += hs[t][threadIdx.x];
  for (int ry = 0; ry < kBoxBand; ry++) {
    sum += hs[ry + 2 * R][threadIdx.x];
    const int y = y0 + ry;
    if (y < s.H && c < s.Wp) {
      const int x = c - s.padx;
      Q[((size_t)img * s.H + y) * s.Wp + c] = 256 * sum + (img < n ? x : -x);
    }
    sum -= hs[ry][threadIdx.x];
  }
}

// -(bytes), bytes in [0, 127]: 0x80 - b never borrows from the neighbouring byte, ^ 0x80 turns it into the two's complement
DEV uint32_t neg_bytes(uint32_t x) { return (0x80808080u - x) ^ 0x80808080u; }

// ---- matching: one side.  SIDE 0: reference columns x of the left image against u = x - d of the right one; SIDE 1: reference columns u
// of the right image against x = u + d of the left one.  keys: [n][H][W] uint32 (cost << 8 | d). ----
template <int NT, int R, int SIDE>
__global__ void __launch_bounds__(256, 2) k_bmq_match(QDev s, int n, int band, const uint8_t* __restrict__ g, const int32_t* __restrict__ Q,
                                                   uint32_t* __restrict__ keys_out) {
  constexpr int TAPS = 2 * R + 1, RING = 2 * R + 2;
  constexpr int WAD = 8 * NT + 8;                               // dwords of an A-side row: 4 bytes ahead of the first tile, 32 NT columns, 4 + 12 behind, rounded up
  constexpr int WBD = 16;                                       // dwords of a B-side row: 4 + 32 + 4 + 12, rounded up
  constexpr int NLD = (WAD + WBD + 63) / 64;                    // image dwords a lane fetches per row
  constexpr int NQL = (32 * NT + 63) / 64;                      // key halves of the A side a lane fetches per row
  constexpr int WAVE_DW = RING * (2 * WAD + WBD) + WAD + WBD + 32 * NT;
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, c = lane & 31;
  const int x0 = (blockIdx.x * 4 + wave) * 32, y0 = blockIdx.y * band, img = blockIdx.z;
  if (x0 >= s.W) return;                                        // whole waves only: there is no workgroup barrier in this kernel
  const int y1 = min(y0 + band, s.H);
  uint32_t* apos = lds + (size_t)wave * WAVE_DW;                // [RING][WAD]   A-side rows as they are
  uint32_t* aneg = apos + RING * WAD;                           // [RING][WAD]   negated
  uint32_t* bb = aneg + RING * WAD;                             // [RING][WBD]   B-side rows
  uint32_t* azero = bb + RING * WBD;                            // [WAD + WBD]   zeros: "no row leaves" on the first row of a band
  int32_t* qrow = reinterpret_cast<int32_t*>(azero + WAD + WBD);   // [32 NT]     the A side's key halves of the current row
  for (int i = lane; i < WAD + WBD; i += 64) azero[i] = 0u;
  const int imgA = SIDE == 0 ? n + img : img, imgB = SIDE == 0 ? img : n + img;
  const int colA = s.padx + (SIDE == 0 ? x0 - 32 * (NT - 1) : x0) - 4, colB = s.padx + x0 - 4;    // multiples of 4
  const uint8_t* gA = g + (size_t)imgA * s.H * s.Wp + colA;
  const uint8_t* gB = g + (size_t)imgB * s.H * s.Wp + colB;
  const int32_t* QA = Q + (size_t)imgA * s.H * s.Wp + colA + 4;  // key half of the A side's first tile column
  const int32_t* QB = Q + (size_t)imgB * s.H * s.Wp + colB + 4 + c;
  // a lane's dwords of a row fetch: the A-side row first, the B-side row behind it
  const uint8_t* src[NLD];
  int dst[NLD];                                                  // dword index inside [apos row | bb row], -1: nothing
#pragma unroll
  for (int k = 0; k < NLD; k++) {
    const int i = lane + 64 * k;
    src[k] = i < WAD ? gA + 4 * i : gB + 4 * min(i - WAD, WBD - 1);
    dst[k] = i < WAD + WBD ? i : -1;
  }
  auto fetch_row = [&](int t, uint32_t (&d)[NLD]) __attribute__((always_inline)) {
    const size_t ro = (size_t)min(max(t, 0), s.H - 1) * s.Wp;
#pragma unroll
    for (int k = 0; k < NLD; k++) d[k] = *reinterpret_cast<const uint32_t*>(src[k] + ro);
  };
  auto commit_row = [&](int slot, const uint32_t (&d)[NLD]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NLD; k++) {
      if (dst[k] < 0) continue;
      if (dst[k] < WAD) { apos[slot * WAD + dst[k]] = d[k]; aneg[slot * WAD + dst[k]] = neg_bytes(d[k]); }
      else bb[slot * WBD + dst[k] - WAD] = d[k];
    }
  };
  // operands: this lane's 2r+1 taps of a row, starting at its column - r; dword-aligned reads + a per-lane byte shift
  const int offB = c + 4 - R, offA = c + 4 - R;                  // byte offsets in the staged rows (tile t of the A side adds 32 t)
  const int dwB = offB >> 2, dwA = offA >> 2;
  const uint32_t shB = offB & 3, shA = offA & 3;
  auto operand = [&](const uint32_t* row, int dw, uint32_t sh) __attribute__((always_inline)) {
    uint32_t w0, w1, w2;
    window(row[dw], row[dw + 1], row[dw + 2], sh, TAPS, w0, w1, w2);
    return (v4i){(int)w0, (int)w1, (int)w2, 0};
  };
  v16i acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int v = 0; v < 16; v++) acc[t][v] = 0;
  // one step of the running sums: half 0 of the contraction takes (rowA0, rowB0), half 1 (rowA1, rowB1)
  auto step = [&](const uint32_t* rowA0, const uint32_t* rowB0, const uint32_t* rowA1, const uint32_t* rowB1) __attribute__((always_inline)) {
    const uint32_t* ra = h ? rowA1 : rowA0;
    const v4i b = operand(h ? rowB1 : rowB0, dwB, shB);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(operand(ra + 8 * t, dwA, shA), b, acc[t], 0, 0, 0);
  };
  // ---- warm-up: rows y0 - r .. y0 + r - 1 enter the sums two at a time (both negated: the accumulators hold -X) ----
  const int tbase = y0 - R;                                      // image row of ring slot 0
  {
    uint32_t d[NLD];
    for (int t = 0; t < 2 * R; t++) { fetch_row(tbase + t, d); commit_row(t, d); }
  }
  for (int t = 0; t < 2 * R; t += 2) step(aneg + t * WAD, bb + t * WBD, aneg + (t + 1) * WAD, bb + (t + 1) * WBD);
  // ---- the band ----
  uint32_t nxt[NLD];                                             // the row entering next, fetched a row ahead
  int32_t qn[NQL], pn;                                           // the key halves of the next row
  auto fetch_keys = [&](int y) __attribute__((always_inline)) {
    const size_t ro = (size_t)min(y, s.H - 1) * s.Wp;
#pragma unroll
    for (int k = 0; k < NQL; k++) qn[k] = QA[ro + min(lane + 64 * k, 32 * NT - 1)];
    pn = QB[ro];
  };
  fetch_row(y0 + R, nxt);
  fetch_keys(y0);
  int slot_new = 2 * R, slot_old = 2 * R + 1;                    // slot of row y + r; of row y - r - 1 (which first exists for y0 + 1: slot 0)
  for (int y = y0; y < y1; y++) {
    commit_row(slot_new, nxt);
#pragma unroll
    for (int k = 0; k < NQL; k++) if ((32 * NT) % 64 == 0 || lane + 64 * k < 32 * NT) qrow[lane + 64 * k] = qn[k];
    const int32_t p_here = pn;
    fetch_row(y + R + 1, nxt);
    fetch_keys(y + 1);
    const bool first = y == y0;
    step(aneg + slot_new * WAD, bb + slot_new * WBD, first ? azero : apos + slot_old * WAD, first ? azero + WAD : bb + slot_old * WBD);
    // ---- winner: min over the band of (-X << 9) + key half; in-lane over the 16 rows of every tile, then across the two half-waves ----
    int32_t m = 0x7FFFFFFF, v0[16];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      int32_t val[16];
#pragma unroll
      for (int gq = 0; gq < 4; gq++) {
        const int4 q4 = *reinterpret_cast<const int4*>(qrow + 32 * t + 8 * gq + 4 * h);
        val[4 * gq + 0] = (int32_t)(((uint32_t)acc[t][4 * gq + 0] << 9) + (uint32_t)q4.x);
        val[4 * gq + 1] = (int32_t)(((uint32_t)acc[t][4 * gq + 1] << 9) + (uint32_t)q4.y);
        val[4 * gq + 2] = (int32_t)(((uint32_t)acc[t][4 * gq + 2] << 9) + (uint32_t)q4.z);
        val[4 * gq + 3] = (int32_t)(((uint32_t)acc[t][4 * gq + 3] << 9) + (uint32_t)q4.w);
      }
      if (t == 0) {
#pragma unroll
        for (int v = 0; v < 16; v++) v0[v] = val[v];
      } else if (t == NT - 1) {
        // row i of the first tile is disparity D + c - i (SIDE 0) / i - c (SIDE 1), of the last tile c - i / D + i - c: exactly one of the two lies in [0, D)
#pragma unroll
        for (int v = 0; v < 16; v++) {
          const int i = 8 * (v >> 2) + 4 * h + (v & 3);
          const bool take_first = SIDE == 0 ? i > c : i >= c;
          val[v] = take_first ? v0[v] : val[v];
        }
      }
      if (t > 0) {
#pragma unroll
        for (int v = 0; v < 16; v += 2) m = min(min(m, val[v]), val[v + 1]);       // v_min3_i32: two candidates per instruction
      }
    }
    {
      const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)m, (unsigned)m, false, false);
      m = min((int32_t)sw[0], (int32_t)sw[1]);
    }
    if (h == 0 && x0 + c < s.W) keys_out[((size_t)img * s.H + y) * s.W + x0 + c] = (uint32_t)(m + p_here);
    slot_new = slot_new + 1 == RING ? 0 : slot_new + 1;
    slot_old = slot_old + 1 == RING ? 0 : slot_old + 1;
  }
}

// ---- L/R check and output; with the sub-pixel option the two costs next to the winner straight from their definition:
// C = |P_L|^2 + |P_R|^2 - 2 P_L . P_R with the squared norms from Q and the dot product by v_dot4 ----
template <int R>
__global__ void __launch_bounds__(256) k_bmq_finish(QDev s, int n, const uint8_t* __restrict__ g, const int32_t* __restrict__ Q, const uint32_t* __restrict__ keysL,
                                                    const uint32_t* __restrict__ keysR, int16_t* __restrict__ disp, uint8_t* __restrict__ u8) {
  constexpr int TAPS = 2 * R + 1;
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, img = blockIdx.z;
  if (x >= s.W) return;
  const size_t row = ((size_t)img * s.H + y) * s.W;
  const uint32_t kl = keysL[row + x];
  const int d = (int)(kl & 0xFFu);
  bool ok = true;
  if (s.lr >= 0) ok = x - d >= 0 && abs(d - (int)(keysR[row + x - d] & 0xFFu)) <= s.lr;
  const int scale = s.subpixel ? 16 : 1;
  int out = -scale;
  if (ok) {
    out = scale * d;
    if (s.subpixel && d > 0 && d < s.D - 1) {
      const uint8_t* gL = g + (size_t)img * s.H * s.Wp;
      const uint8_t* gR = g + (size_t)(n + img) * s.H * s.Wp;
      const int32_t* qL = Q + ((size_t)img * s.H + y) * s.Wp + s.padx;
      const int32_t* qR = Q + ((size_t)(n + img) * s.H + y) * s.Wp + s.padx;
      auto norm = [&](const int32_t* q, int p, bool left) { return (q[p] - (left ? p : -p)) >> 8; };     // |P|^2 back out of the key half
      const int cl0 = s.padx + x - R, um = x - (d - 1), up = x - (d + 1);                              // first window bytes
      uint32_t dm = 0, dp = 0;
      for (int j = -R; j <= R; j++) {
        const size_t ro = (size_t)min(max(y + j, 0), s.H - 1) * s.Wp;
        auto win = [&](const uint8_t* base, int col, uint32_t& w0, uint32_t& w1, uint32_t& w2) {
          const uint32_t* p = reinterpret_cast<const uint32_t*>(base + ro + (col & ~3));
          window(p[0], p[1], p[2], col & 3, TAPS, w0, w1, w2);
        };
        uint32_t a0, a1, a2, m0, m1, m2, p0, p1, p2;
        win(gL, cl0, a0, a1, a2); win(gR, s.padx + um - R, m0, m1, m2); win(gR, s.padx + up - R, p0, p1, p2);
        dm = __builtin_amdgcn_udot4(a0, m0, __builtin_amdgcn_udot4(a1, m1, __builtin_amdgcn_udot4(a2, m2, dm, false), false), false);
        dp = __builtin_amdgcn_udot4(a0, p0, __builtin_amdgcn_udot4(a1, p1, __builtin_amdgcn_udot4(a2, p2, dp, false), false), false);
      }
      const int nl = norm(qL, x, true);
      const int c0 = (int)(kl >> 8), cm1 = nl + norm(qR, um, false) - 2 * (int)dm, cp1 = nl + norm(qR, up, false) - 2 * (int)dp;
      const int den = max(cm1 + cp1 - 2 * c0, 1);
      out = 16 * d + (16 * (cm1 - cp1) + den) / (2 * den);
    }
  }
  disp[row + x] = (int16_t)out;
  if (u8) {                                                    // the node's mono8 map of the same value (point_cloud.cpp:422 semantics, as bm.hip's bm_u8): invalid -> 0,
    int v = 0;                                                 // 1/16 pixel rounded half to even, saturated at 255
    if (out >= 0) {
      v = out;
      if (s.subpixel) { const int q = out >> 4, r = out & 15; v = q + ((r > 8 || (r == 8 && (q & 1))) ? 1 : 0); }
      v = min(v, 255);
    }
    u8[row + x] = (uint8_t)v;
  }
}

// ------------------------------------------------------------------- host side -------------------------------------------------------------------
void geometry(int W, int H, int D, int r, int cap, int lr, int subpixel, QDev* s, Sizes* z, int max_batch) {
  s->W = W; s->H = H; s->D = D; s->r = r; s->cap = cap; s->lr = lr; s->subpixel = subpixel ? 1 : 0;
  s->NT = D / 32 + 1;
  s->padx = D + 96;                                             // left of the image: D + 4 + alignment; right: a reference tile beyond W plus D plus the window
  s->Wp = (W + 2 * s->padx + 15) & ~15;
  z->g = (size_t)2 * max_batch * H * s->Wp + 64;
  z->q = ((size_t)2 * max_batch * H * s->Wp + 64) * sizeof(int32_t);
}

template <int NT, int R, int SIDE>
static hipError_t launch_match(hipStream_t st, const QDev& s, int n, int band, const uint8_t* g, const int32_t* Q, uint32_t* keys) {
  constexpr int RING = 2 * R + 2, WAD = 8 * NT + 8, WBD = 16, WAVE_DW = RING * (2 * WAD + WBD) + WAD + WBD + 32 * NT;
  const size_t ldsb = (size_t)4 * WAVE_DW * sizeof(uint32_t);
  if (ldsb > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bmq_match<NT, R, SIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    if (e != hipSuccess) return e;
  }
  const dim3 grid(((s.W + 31) / 32 + 3) / 4, (s.H + band - 1) / band, n);
  hipLaunchKernelGGL((k_bmq_match<NT, R, SIDE>), grid, dim3(256), ldsb, st, s, n, band, g, Q, keys);
  return hipGetLastError();
}
template <int R, int SIDE>
static hipError_t launch_match_r(hipStream_t st, const QDev& s, int n, int band, const uint8_t* g, const int32_t* Q, uint32_t* keys) {
  switch (s.NT) {
    case 2: return launch_match<2, R, SIDE>(st, s, n, band, g, Q, keys);
    case 3: return launch_match<3, R, SIDE>(st, s, n, band, g, Q, keys);
    case 4: return launch_match<4, R, SIDE>(st, s, n, band, g, Q, keys);
    case 5: return launch_match<5, R, SIDE>(st, s, n, band, g, Q, keys);
    case 6: return launch_match<6, R, SIDE>(st, s, n, band, g, Q, keys);
    case 7: return launch_match<7, R, SIDE>(st, s, n, band, g, Q, keys);
    case 8: return launch_match<8, R, SIDE>(st, s, n, band, g, Q, keys);
    default: return launch_match<9, R, SIDE>(st, s, n, band, g, Q, keys);
  }
}
template <int SIDE>
static hipError_t launch_match_any(hipStream_t st, const QDev& s, int n, int band, const uint8_t* g, const int32_t* Q, uint32_t* keys) {
  switch (s.r) {
    case 2: return launch_match_r<2, SIDE>(st, s, n, band, g, Q, keys);
    case 3: return launch_match_r<3, SIDE>(st, s, n, band, g, Q, keys);
    default: return launch_match_r<4, SIDE>(st, s, n, band, g, Q, keys);
  }
}

hipError_t run(const QDev& s, int n, const uint8_t* dI1, const uint8_t* dI2, int pitch, long long stride, uint8_t* g, int32_t* Q, uint32_t* keysL, uint32_t* keysR,
               int16_t* dDisp, uint8_t* dU8, hipStream_t st, hipEvent_t* ev) {
  hipError_t e;
  hipLaunchKernelGGL(k_bmq_prefilter, dim3((s.Wp / 4 + 255) / 256, s.H, 2 * n), dim3(256), 0, st, s, dI1, dI2, pitch, stride, n, g);
  const dim3 bgrid((s.Wp + 255) / 256, (s.H + kBoxBand - 1) / kBoxBand, 2 * n);
  switch (s.r) {
    case 2: hipLaunchKernelGGL(k_bmq_box<2>, bgrid, dim3(256), 0, st, s, n, g, Q); break;
    case 3: hipLaunchKernelGGL(k_bmq_box<3>, bgrid, dim3(256), 0, st, s, n, g, Q); break;
    default: hipLaunchKernelGGL(k_bmq_box<4>, bgrid, dim3(256), 0, st, s, n, g, Q); break;
  }
  if ((e = hipEventRecord(ev[1], st)) != hipSuccess) return e;
  // rows per wave: long bands amortise the 2r rows of warm-up, short ones fill the GPU when the batch is small (a lone pair)
  int band = 96;
  const long long tiles = (long long)((s.W + 31) / 32) * n;
  while (band > 12 && tiles * ((s.H + band - 1) / band) < 8192) band = (band + 1) / 2;
  if (const char* env = JN_HOOK_ENV("JN_BMQ_BAND")) band = std::min(std::max(atoi(env), 1), 1024);
  if ((e = launch_match_any<0>(st, s, n, band, g, Q, keysL)) != hipSuccess) return e;
  if (s.lr >= 0 && (e = launch_match_any<1>(st, s, n, band, g, Q, keysR)) != hipSuccess) return e;
  if ((e = hipEventRecord(ev[2], st)) != hipSuccess) return e;
  const dim3 fgrid((s.W + 255) / 256, s.H, n);
  switch (s.r) {
    case 2: hipLaunchKernelGGL(k_bmq_finish<2>, fgrid, dim3(256), 0, st, s, n, g, Q, keysL, keysR, dDisp, dU8); break;
    case 3: hipLaunchKernelGGL(k_bmq_finish<3>, fgrid, dim3(256), 0, st, s, n, g, Q, keysL, keysR, dDisp, dU8); break;
    default: hipLaunchKernelGGL(k_bmq_finish<4>, fgrid, dim3(256), 0, st, s, n, g, Q, keysL, keysR, dDisp, dU8); break;
  }
  return hipGetLastError();
}

}  // namespace jnav_bmq
