// oracle/oracle.h — C ABI of the CPU restatement (TEST INFRASTRUCTURE ONLY).
//
// Everything under oracle/ is a checker: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it.  The product (jackal_navigation_amd/) never links or calls it.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

// Mirror of Elas::parameters (reference elas.h:60-82), same order, bools as int32.
typedef struct orc_params {
  int32_t disp_min, disp_max;
  float   support_threshold;
  int32_t support_texture, candidate_stepsize, incon_window_size, incon_threshold, incon_min_support;
  int32_t add_corners, grid_size;
  float   beta, gamma, sigma, sradius;
  int32_t match_texture, lr_threshold;
  float   speckle_sim_threshold;
  int32_t speckle_size, ipol_gap_width;
  int32_t filter_median, filter_adaptive_mean, postprocess_only_left, subsampling;
} orc_params;

// elas.h:92-145 presets. setting 0 = ROBOTICS, 1 = MIDDLEBURY.
void orc_params_default(orc_params* p, int setting);

// SURVEY.md Appendix A synthetic stereo pair (xorshift32, seed; scene disparity range sceneD).
void orc_synth_pair(int32_t W, int32_t H, int32_t sceneD, uint32_t seed, uint8_t* L, uint8_t* R);

// FNV-1a-64 over 32-bit words (the hash SURVEY.md §8c quotes for final D1 maps).
uint64_t orc_fnv1a64_u32(const uint32_t* words, int64_t n);

// ---- ELAS stages (each cites the reference lines it restates, see elas_oracle.cpp) ----------
// Sobel: I [H][bpl] -> du,dv [H][bpl]; rows 0 and H-1 are left untouched.
void orc_sobel3x3(const uint8_t* I, int32_t bpl, int32_t H, uint8_t* du, uint8_t* dv);
// Descriptor: [H][W][16]; bytes outside u in [3,W-4], v in [3,H-4] are set to 0 — the reference leaves them
// uninitialised and reads columns 2 and W-3 (see elas_oracle.cpp).  orc_set_uninit_fill picks another byte (tests only).
void orc_set_uninit_fill(int32_t byte);
void orc_descriptor(const uint8_t* I, int32_t W, int32_t H, int32_t pitch, uint8_t* desc);
// One support candidate (returns d or -1).
int32_t orc_match_candidate(const orc_params* p, const uint8_t* desc1, const uint8_t* desc2, int32_t W,
                            int32_t H, int32_t u, int32_t v, int right);
// Candidate grid before filtering: D_can [ch][cw] (row 0 / col 0 stay 0).  Returns cw*ch.
int32_t orc_candidates(const orc_params* p, const uint8_t* desc1, const uint8_t* desc2, int32_t W, int32_t H,
                       int16_t* D_can, int32_t* cw, int32_t* ch);
void orc_remove_inconsistent(const orc_params* p, int16_t* D_can, int32_t cw, int32_t ch);
void orc_remove_redundant(int16_t* D_can, int32_t cw, int32_t ch, int32_t max_dist, int32_t thresh, int vertical);
// Full support stage: returns count, writes (u,v,d) triples.
int32_t orc_support(const orc_params* p, const uint8_t* desc1, const uint8_t* desc2, int32_t W, int32_t H,
                    int32_t* uvd, int32_t cap);
// Delaunay triangulation of float points exactly as Triangle "zQB" produces it (corner order too).
int32_t orc_triangulate(const float* xy, int32_t n, int32_t* corners, int32_t cap);
// Triangles + planes for one side: corners [n][3], planes [n][6] (t1a,t1b,t1c,t2a,t2b,t2c).
int32_t orc_triangles(const int32_t* uvd, int32_t nsup, int right, int32_t* corners, float* planes, int32_t cap);
// Grid prior: grid [gh][gw][disp_max+2].
void orc_grid(const orc_params* p, const int32_t* uvd, int32_t nsup, int32_t W, int32_t H, int right,
              int32_t* grid, int32_t* dims3);
// Dense matching for one side.
void orc_dense(const orc_params* p, const int32_t* uvd, int32_t nsup, const int32_t* corners,
               const float* planes, int32_t ntri, const int32_t* grid, const int32_t* grid_dims,
               const uint8_t* desc1, const uint8_t* desc2, int32_t W, int32_t H, int right, float* D);
void orc_lr_check(const orc_params* p, float* D1, float* D2, int32_t W, int32_t H);
void orc_speckle(const orc_params* p, float* D, int32_t W, int32_t H);
void orc_gap(const orc_params* p, float* D, int32_t W, int32_t H);
void orc_adaptive_mean(float* D, int32_t W, int32_t H);
void orc_median(float* D, int32_t W, int32_t H);
// Whole pipeline == Elas::process (elas.cpp:32-151).  Returns 0, or 1 if <3 support points
// (D1/D2 untouched, as the reference).
int32_t orc_elas_process(const orc_params* p, const uint8_t* I1, const uint8_t* I2, float* D1, float* D2,
                         int32_t W, int32_t H, int32_t pitch);

// ---- node side (point_cloud.cpp) ------------------------------------------------------------
typedef struct orc_scan_params {
  double Q[16];       // 4x4 row-major reprojection matrix (point_cloud.cpp:543-544)
  double XR[9];       // camera->robot rotation (calibration yml XR)
  double XT[3];       // camera->robot translation
  int32_t crop_offset_x, crop_offset_y;   // point_cloud.cpp:51-52
  double gp_height_thresh, gp_angle_thresh, gp_dist_thresh;   // :66-68
  double fov_deg;     // :217
  int32_t bins;       // :218
  double pi_approx;   // 3.1415 literal used at :254 and :277
} orc_scan_params;

void orc_scan_params_default(orc_scan_params* sp, int32_t W, int32_t H);
// convertTo(CV_8U): round-half-even + saturate (point_cloud.cpp:422).
void orc_disparity_to_u8(const float* D, uint8_t* out, int64_t n);
// cacheDisparityValues (point_cloud.cpp:104-147): lut [H][W][2].
void orc_build_valid_disp_lut(const orc_scan_params* sp, int32_t W, int32_t H, uint8_t* lut);
// publishObstacleScan(Mat&) (point_cloud.cpp:213-296): un-compacted bins (1e9 = empty) + 4 scalars
// {angle_min, angle_max, range_min, range_max}.  Returns number of pixels that contributed.
int64_t orc_obstacle_scan(const orc_scan_params* sp, const uint8_t* disp, const uint8_t* lut, int32_t W,
                          int32_t H, double* bins, double* meta4);
// ranges compaction of :278-282; returns count.
int32_t orc_compact_ranges(const double* bins, int32_t nb, float* ranges);
// publishPointCloud -g path (point_cloud.cpp:321-352): xyz float32 triples in i-outer/j-inner order.
int64_t orc_point_cloud(const orc_scan_params* sp, const uint8_t* disp, int32_t W, int32_t H, float* xyz);
// publishObstacleScan(vector<Point3d>) (point_cloud.cpp:149-211): ground-plane filter + binning of
// robot-frame points given as doubles [n][3].
int64_t orc_obstacle_scan_points(const orc_scan_params* sp, const double* xyz, int64_t n, double* bins, double* meta4);

int64_t orc_obstacle_scan_cloud(const orc_scan_params* sp, const uint8_t* disp, int32_t W, int32_t H, double* bins, double* meta4);
// rectification front end (definitions; OpenCV-side parity is unpinned)
void orc_init_undistort_rectify_map(const double* K, const double* D, const double* R, const double* P, int32_t W, int32_t H,
                                    float* mapx, float* mapy);
void orc_remap_bilinear(const uint8_t* src, int32_t sw, int32_t sh, int32_t spitch, const float* mapx, const float* mapy,
                        uint8_t* dst, int32_t W, int32_t H, int32_t dpitch);

#ifdef __cplusplus
}
#endif
