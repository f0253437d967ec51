// sgm_sweep.h — host interface of the sweep kernels (sgm_sweep.hip) used by the SGM mode's C ABI (sgm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace jnav_sgm {

struct SwDev {
  int W, H, D, P1, P2, lr, subpixel, cap;
  int Wp, padl;      // prefiltered rows, x-mirrored and padded: Wp bytes per row, image column x_k = W-1-x at byte padl + x_k
  int NB, xmin;      // sheared blocks of the row sweeps: x' = x_k - (row in sweep order) in [xmin, W-1], NB blocks
  int dbg;           // JN_SGM_DBG profiling switches of k_sw_w (builds with -DJN_SGM_PROFILE only; results are then WRONG): see sgm_sweep.hip
  int wide;          // 3 P2 > 255: the three-path volume is u16, the horizontal volumes are unpacked one by one
  int epoch;         // k_sw_w: 16-bit tag of this launch's boundary columns (set per launch from SweepBuffers::epoch, never 0)
};

struct SweepSizes { size_t gm, vol, gx, flags, minr, dl; };     // bytes; the F volume takes 2 * vol when SwDev::wide

struct SweepBuffers {
  uint8_t* gm;            // prefiltered rows [2 n][H][Wp]
  uint8_t* volF;          // m of the three downward paths [n][H][W][D] (u8, or u16 when wide)
  uint8_t* volH0;         // m of the horizontal path walking x_k upwards
  uint8_t* volH1;         // ... downwards
  uint32_t* gx;           // boundary columns handed from block to block [n][NB][H][3][4][D/8]; zeroed by the owner when allocated
  size_t gx_bytes;        // size of gx (all max_batch frames): what is zeroed when the tag wraps
  uint32_t epoch;         // launches of k_sw_w on gx so far, modulo 2^16 (sweep_run advances it and zeroes gx when it wraps)
  uint32_t* flags;        // the ticket counters of the two row sweeps ([0] downward, [1] upward)
  uint32_t* minr;         // right-image winners [n][H][W] (S << 16 | d)
  uint32_t* dl;           // left winners [n][H][W] (d | d16 << 16), mirrored columns
  // The horizontal sweep and the downward sweep are independent (both read the prefiltered rows, they write different volumes): they run
  // CONCURRENTLY, the horizontal one on this side stream (created by the first sweep_run, destroyed by sweep_release), forked and joined
  // with the two events.  One is bound by its volume writes, the other by issue and synchronisation: together they fill the GPU better.
  hipStream_t side;
  hipEvent_t ev_fork, ev_join;
};
void sweep_release(SweepBuffers& b);     // the side stream and its events (the device buffers belong to the caller)

void sweep_geometry(int W, int H, int D, int P1, int P2, int cap, int lr, int subpixel, SwDev* s, SweepSizes* z, int max_batch);

// Queues prefilter, the two horizontal paths, the downward sweep, the upward sweep + winners, the L/R check on `st`.
// ev[0..3] are recorded before the prefilter, before the paths, before the final sweep and at the end.
// side_overlap: run the horizontal sweep on the buffers' side stream next to the downward sweep (a lone batch); JN_SGM_OVERLAP=0/1 overrides.
// lr_kernel = false leaves the L/R check (and the int16 map) to the caller's next kernel: jn_sgm_submit_scan's tail applies it while it scans
// (kernels.h, launch_scan with SgmWinners); dDisp is then not written here.
hipError_t sweep_run(const SwDev& s, int n, const uint8_t* dI1, const uint8_t* dI2, int pitch, long long stride, int16_t* dDisp, hipStream_t st,
                     SweepBuffers& b, hipEvent_t* ev, bool side_overlap, bool lr_kernel = true);

}  // namespace jnav_sgm
