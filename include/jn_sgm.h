/* jn_sgm.h — C ABI of the semi-global-matching mode of libjn_stereo.so.
 *
 * NO REFERENCE COUNTERPART.  sourishg/jackal-navigation has exactly one stereo matcher, libelas
 * (SURVEY.md 0.1); BASELINE.json's configs 2-5 and the north star name block matching / 8-path SGM /
 * sub-pixel refinement, which the reference does not contain.  This mode is therefore defined HERE
 * (and restated scalar in oracle/sgm_oracle.cpp, the checker): parity is SELF-REFERENTIAL — "parity
 * unpinned" in the sense of SURVEY.md 8c.  It slots in where generateDisparityMap
 * (src/obstacle_avoidance/point_cloud.cpp:406-429) calls Elas::process: rectified u8 pair in, a
 * disparity map out that the node's tail (jn_disparity_scan and friends in jn_stereo.h) consumes.
 *
 * Definition (all integer arithmetic; D = number of disparities, d in [0, D)):
 *   prefilter  g(x,y) = clamp(Sx(x,y), -cap, cap) + cap, Sx = 3x3 Sobel in x on the u8 image with
 *              replicated borders:  Sx = (I(x+1,y-1) - I(x-1,y-1)) + 2 (I(x+1,y) - I(x-1,y)) + (I(x+1,y+1) - I(x-1,y+1))
 *   cost       C(x,y,d) = sum_{i=-1..1} | gL(cl(x+i), y) - gR(cl(x+i-d), y) |, cl = clamp to [0, W-1]
 *              (SAD over a 1x3 window of the Sobel-prefiltered images, <= 3*2*cap)
 *   paths      8 directions r in {(1,0),(-1,0),(0,1),(0,-1),(1,1),(-1,-1),(-1,1),(1,-1)}; along each line
 *              L_r(p,d) = C(p,d) + min( L_r(p-r,d), L_r(p-r,d-1)+P1, L_r(p-r,d+1)+P1, min_k L_r(p-r,k)+P2 ) - min_k L_r(p-r,k)
 *              (d-1 < 0 and d+1 >= D do not exist; the first pixel of a line has L_r = C)
 *   sum        S(p,d) = sum_r L_r(p,d)        (each L_r <= 3*2*cap + P2 <= 255 is required, so S fits 16 bits)
 *   WTA        dL(p) = smallest d minimising S(p,d);  dR(x,y) = smallest d minimising S(x+d,y,d) over x+d < W
 *   L/R check  dL(p) is kept iff x - dL >= 0 and |dL(p) - dR(x - dL, y)| <= lr_max_diff, else invalid
 *   sub-pixel  (optional) for 0 < d < D-1: den = max(S(d-1) + S(d+1) - 2 S(d), 1),
 *              d16 = 16 d + (16 (S(d-1) - S(d+1)) + den) / (2 den)  (C integer division, truncating); otherwise d16 = 16 d
 *   output     int16 per pixel: d (subpixel = 0) or d16 (subpixel = 1); invalid = -1 (resp. -16)
 */
#ifndef JN_SGM_H
#define JN_SGM_H

#include <stdint.h>
#include "jn_stereo.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jn_sgm_params {
  int32_t num_disparities;   /* D: 64, 128 or 256 */
  int32_t P1, P2;            /* smoothness penalties; 3*2*prefilter_cap + P2 <= 255 */
  int32_t prefilter_cap;     /* Sobel clip, 1..31 */
  int32_t lr_max_diff;       /* L/R check tolerance; < 0 disables the check */
  int32_t subpixel;          /* 0: integer disparities, 1: 1/16 pixel */
} jn_sgm_params;

/* D = 128, P1 = 10, P2 = 60, cap = 31, lr_max_diff = 1, subpixel = 0 */
void jn_sgm_params_default(jn_sgm_params* p);

typedef struct jn_sgm jn_sgm;   /* opaque: buffers for up to max_batch pairs (three byte volumes of W*H*D per pair + boundary columns) */

jn_status jn_sgm_create(const jn_sgm_params* p, int32_t width, int32_t height, int32_t max_batch, int32_t device, jn_sgm** out);
void jn_sgm_destroy(jn_sgm* h);

/* n rectified pairs (device pointers, image b at dI + b*image_stride, rows `pitch` bytes apart) -> dDisp [n][height][width]
 * int16 (device).  Synchronous. */
jn_status jn_sgm_process_batch(jn_sgm* h, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch, int64_t image_stride,
                               int16_t* dDisp);

/* Pipelined form: up to eight batches in flight, one per slot (slot 0 shares jn_sgm_process_batch's buffers; slots 1-7 allocate their
 * own — three byte volumes of max_batch*W*H*D each — when first used).  jn_sgm_submit_scan queues the whole mode on the slot's stream and
 * returns; with sp != NULL also the node's tail on the same stream, as ONE kernel that applies the L/R check, writes dDisp, the mono8 map
 * (jn_sgm_disparity_to_u8's values) into dDispU8 and the LUT scan
 * of it (jn_obstacle_scan's outputs dBins [n][sp->bins], dMeta [n][4]; dLut from jn_build_valid_disp_lut).  jn_sgm_wait returns when the
 * slot's batch is complete (the caller's buffers must stay untouched until then).  Batches on different slots overlap on the GPU — the
 * row sweeps are pipelines whose ends leave part of the GPU idle, which the next batch's sweeps fill.  All device pointers. */
jn_status jn_sgm_submit_scan(jn_sgm* h, int32_t slot, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch, int64_t image_stride,
                             int16_t* dDisp, const jn_scan_params* sp, const uint8_t* dLut, uint8_t* dDispU8, double* dBins, double* dMeta);
jn_status jn_sgm_wait(jn_sgm* h, int32_t slot);

/* Milliseconds of the last batch: prefilter; paths = the two horizontal sweeps + the downward sweep; wta = the upward sweep with the
 * winners + the L/R check. */
typedef struct jn_sgm_times { float prefilter, paths, wta, total; } jn_sgm_times;
jn_status jn_sgm_last_times(jn_sgm* h, jn_sgm_times* out);

/* Test hook: device pointers of the intermediate buffers of the last batch (valid until the next call on `h`).
 * which: 0 = sum of (L - C) over the three downward paths [n][H][W][D] (bytes; 16-bit when info[0] != 0), 1 / 2 = the same for
 * the horizontal path towards -x / +x (bytes), 3 = right-image winners [n][H][W] u32 (S << 16 | d), 4 = left winners
 * [n][H][W] u32 (d | d16 << 16), 5 = prefiltered rows.  All volumes are indexed by the MIRRORED column W-1-x.
 * info = {wide, row pitch of the prefiltered rows, left padding, blocks per frame, implementation (0 = round-2 kernels: no buffers)}. */
const void* jn_sgm_debug_ptr(jn_sgm* h, int32_t which, int32_t info[5]);

/* int16 SGM disparities -> the u8 depth map the node publishes (point_cloud.cpp:422 semantics: invalid -> 0, values
 * saturate at 255; 1/16-pixel input is rounded half-to-even like convertTo does for floats). */
jn_status jn_sgm_disparity_to_u8(int32_t device, const int16_t* dDisp, int32_t subpixel, uint8_t* dOut, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* JN_SGM_H */
