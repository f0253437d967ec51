/* jn_bm.h — C ABI of the block-matching mode of libjn_stereo.so.
 *
 * NO REFERENCE COUNTERPART (like jn_sgm.h): sourishg/jackal-navigation's only matcher is libelas; BASELINE.json's
 * config 2 ("640x480 D=64 block-matching, batch=1, latency mode") and SURVEY.md 8f rank 4 name a block matcher, so
 * it is defined HERE and restated scalar in oracle/bm_oracle.cpp (the checker): parity is SELF-REFERENTIAL ("parity
 * unpinned").  It slots in where generateDisparityMap (src/obstacle_avoidance/point_cloud.cpp:406-429)
 * calls Elas::process; the output format is jn_sgm.h's, so jn_sgm_disparity_to_u8 and the node's tail consume it.
 *
 * Definition (all integer arithmetic; D = number of disparities, d in [0, D); r = block_radius):
 *   prefilter  g = clamp(Sobel_x, -cap, cap) + cap with replicated borders, exactly jn_sgm.h's
 *   cost       CL(x,y,d) = sum_{j=-r..r} sum_{i=-r..r} | gL(cl(x+i), cr(y+j)) - gR(cl(x+i-d), cr(y+j)) |
 *              CR(x,y,d) = sum_{j=-r..r} sum_{i=-r..r} | gR(cl(x+i), cr(y+j)) - gL(cl(x+i+d), cr(y+j)) |
 *              cl = clamp to [0, W-1], cr = clamp to [0, H-1]        (<= (2r+1)^2 * 2 cap, fits 16 bits)
 *              cost_function = JN_BM_COST_SSD: the same with (a - b)^2 in place of |a - b|  (<= (2r+1)^2 * (2 cap)^2 < 2^19).
 *              This is BASELINE.json config 5's "int8 cost volume (CDNA4 MFMA path)": sum (a-b)^2 = sum a^2 + sum b^2 - 2 sum a b,
 *              and the cross term over all (x, x-d) pairs of a row is a banded product of the two images' patch matrices — an
 *              int8 contraction for the matrix cores (csrc/bm_mfma.hip).  A different cost function, hence a different (equally
 *              self-defined) result than SAD; all of WTA / L-R check / sub-pixel below apply to it unchanged.
 *   WTA        dL(x,y) = smallest d minimising CL(x,y,d);  dR(x,y) = smallest d minimising CR(x,y,d)
 *   L/R check  dL(p) is kept iff x - dL >= 0 and |dL(p) - dR(x - dL, y)| <= lr_max_diff, else invalid (lr_max_diff < 0: all kept)
 *   sub-pixel  (optional) for 0 < d < D-1, with C = CL(x,y,.): den = max(C(d-1) + C(d+1) - 2 C(d), 1),
 *              d16 = 16 d + (16 (C(d-1) - C(d+1)) + den) / (2 den)  (C integer division, truncating); otherwise d16 = 16 d
 *   output     int16 per pixel: d (subpixel = 0) or d16 (subpixel = 1); invalid = -1 (resp. -16)
 */
#ifndef JN_BM_H
#define JN_BM_H

#include <stdint.h>
#include "jn_stereo.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jn_bm_params {
  int32_t num_disparities;   /* D: a multiple of 8 in [8, 256] */
  int32_t block_radius;      /* r: 2, 3 or 4 (5x5, 7x7, 9x9 windows) */
  int32_t prefilter_cap;     /* Sobel clip, 1..31 */
  int32_t lr_max_diff;       /* L/R check tolerance; < 0 disables the check (and the right-referenced pass) */
  int32_t subpixel;          /* 0: integer disparities, 1: 1/16 pixel */
  int32_t cost_function;     /* JN_BM_COST_SAD (default) or JN_BM_COST_SSD (then D must be a multiple of 32 in [32, 256]) */
} jn_bm_params;   /* ABI: six int32 since jn_version() "jn_stereo 0.3" (0.2 had five, without cost_function); the struct carries no size
                   * field, so a caller checks jn_version() — or simply fills the struct through jn_bm_params_default() — before jn_bm_create */
#define JN_BM_COST_SAD 0
#define JN_BM_COST_SSD 1

/* D = 64, r = 4, cap = 31, lr_max_diff = 1, subpixel = 0, cost_function = JN_BM_COST_SAD */
void jn_bm_params_default(jn_bm_params* p);

typedef struct jn_bm jn_bm;   /* opaque: padded prefiltered rows and one 8-byte winner record per pixel and side for max_batch pairs */

jn_status jn_bm_create(const jn_bm_params* p, int32_t width, int32_t height, int32_t max_batch, int32_t device, jn_bm** out);
void jn_bm_destroy(jn_bm* h);

/* n rectified pairs (device pointers, image b at dI + b*image_stride, rows `pitch` bytes apart) -> dDisp [n][height][width]
 * int16 (device).  Synchronous. */
jn_status jn_bm_process_batch(jn_bm* h, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch, int64_t image_stride,
                              int16_t* dDisp);

/* The same plus the node's tail on the same stream, one synchronisation in all (latency mode): the mono8 depth map
 * (point_cloud.cpp:422 semantics, as jn_sgm_disparity_to_u8) and the LUT scan of it (jn_obstacle_scan's outputs:
 * dBins [n][sp->bins], dMeta [n][4]); dLut from jn_build_valid_disp_lut.  All device pointers. */
jn_status jn_bm_process_scan(jn_bm* h, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch, int64_t image_stride,
                             int16_t* dDisp, const jn_scan_params* sp, const uint8_t* dLut, uint8_t* dDispU8, double* dBins, double* dMeta);

/* Milliseconds of the last batch: prefilter, the left- and right-referenced matching launches, check + output (+ the scan
 * tail after jn_bm_process_scan). */
typedef struct jn_bm_times { float prefilter, match, finish, total; } jn_bm_times;
/* Several batches in flight (as jn_sgm_submit_scan / jn_sgm_wait): slot in [0, 6); each slot has its own stream, events and scratch,
 * allocated when the slot is first used (slot 0 is the synchronous calls').  jn_bm_submit_scan queues the matcher and — when sp is not
 * NULL — the mono8 map and the LUT scan on the slot's stream and returns; jn_bm_wait(slot) must precede reading the slot's outputs or
 * submitting to it again.  With sp == NULL the u8 / bins / meta pointers may be NULL (disparities only).  A process that keeps more than
 * eight streams busy should export GPU_MAX_HW_QUEUES=16. */
jn_status jn_bm_submit_scan(jn_bm* h, int32_t slot, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch, int64_t image_stride,
                            int16_t* dDisp, const jn_scan_params* sp, const uint8_t* dLut, uint8_t* dDispU8, double* dBins, double* dMeta);
jn_status jn_bm_wait(jn_bm* h, int32_t slot);
jn_status jn_bm_last_times(jn_bm* h, jn_bm_times* out);

#ifdef __cplusplus
}
#endif
#endif /* JN_BM_H */
