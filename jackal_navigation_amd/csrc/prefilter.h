// prefilter.h — the Sobel-x prefilter shared by the SGM and block-matching modes (include/jn_sgm.h / jn_bm.h: g = clamp(Sobel_x, -cap, cap) + cap,
// replicated borders), four output columns per thread.  Product code, device only.
//
// The three modes store different functions of g in rows with different padding (and the SGM sweeps store them x-mirrored); what they share
// is this: for the four output bytes of an aligned dword the 3x3 Sobel needs six source columns of three rows — three 8-byte loads instead of
// 24 byte loads, and one dword store.  Columns whose window touches the image border or the row's padding take the scalar form.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jnav_pre {

// Sobel-x (not yet clamped) at column x of the rows r0, r1, r2 (already clamped in y), columns clamped to [0, W-1]
static __device__ __forceinline__ int sobel_x_clamped(const uint8_t* r0, const uint8_t* r1, const uint8_t* r2, int x, int W) {
  const int xm = max(x - 1, 0), xq = min(x + 1, W - 1);
  return ((int)r0[xq] - (int)r0[xm]) + 2 * ((int)r1[xq] - (int)r1[xm]) + ((int)r2[xq] - (int)r2[xm]);
}

// The four responses at image columns x(k) = clampW(x0 + S k), k = 0..3 (S = +1: ascending, -1: descending), clamped to [-cap, cap].
template <int S>
static __device__ __forceinline__ void sobel4(const uint8_t* r0, const uint8_t* r1, const uint8_t* r2, int x0, int W, int cap, int (&out)[4]) {
  const int lo = S > 0 ? x0 : x0 - 3, hi = S > 0 ? x0 + 3 : x0;            // the four columns, if none of them needs a clamp
  const int base = lo - 1;                                                  // first byte of the 8-byte window: columns lo-1 .. lo+6
  if (lo >= 1 && hi <= W - 2 && base + 7 <= W - 1) {
    uint64_t a, b, c;
    __builtin_memcpy(&a, r0 + base, 8); __builtin_memcpy(&b, r1 + base, 8); __builtin_memcpy(&c, r2 + base, 8);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = S > 0 ? k + 1 : 4 - k;                                  // byte of column x0 + S k inside the window
      const int sh_hi = 8 * (i + 1), sh_lo = 8 * (i - 1);
      const int sx = ((int)((a >> sh_hi) & 255u) - (int)((a >> sh_lo) & 255u)) + 2 * ((int)((b >> sh_hi) & 255u) - (int)((b >> sh_lo) & 255u)) +
                     ((int)((c >> sh_hi) & 255u) - (int)((c >> sh_lo) & 255u));
      out[k] = min(max(sx, -cap), cap);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = min(max(sobel_x_clamped(r0, r1, r2, min(max(x0 + S * k, 0), W - 1), W), -cap), cap);
  }
}

}  // namespace jnav_pre
