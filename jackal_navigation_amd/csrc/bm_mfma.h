// bm_mfma.h — host interface of the matrix-core block matcher (bm_mfma.hip: include/jn_bm.h's JN_BM_COST_SSD) used by bm.hip's C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace jnav_bmq {

struct QDev {
  int W, H, D, r, cap, lr, subpixel;
  int Wp, padx;      // prefiltered rows: Wp bytes, image column x at byte padx + x, replicated borders
  int NT;            // tiles of 32 candidate columns that cover the band of D disparities: D / 32 + 1
};
struct Sizes { size_t g, q; };     // bytes: prefiltered rows [2 n][H][Wp] u8; key halves [2 n][H][Wp] int32

void geometry(int W, int H, int D, int r, int cap, int lr, int subpixel, QDev* s, Sizes* z, int max_batch);

// Queues prefilter, squared patch norms, the left- and right-referenced matching passes, the L/R check + output on `st`.
// ev[1] is recorded before the matching passes, ev[2] behind them (ev[0] / ev[3] are the caller's).
hipError_t run(const QDev& s, int n, const uint8_t* dI1, const uint8_t* dI2, int pitch, long long stride, uint8_t* g, int32_t* Q, uint32_t* keysL, uint32_t* keysR,
               int16_t* dDisp, uint8_t* dU8, hipStream_t st, hipEvent_t* ev);

}  // namespace jnav_bmq
