// kernels.h — launchers of the gfx950 kernels (defined in kernels.hip).  Product code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "jn_types.h"
#include "../../include/jn_stereo.h"

namespace jnav {

// Sets the dynamic-LDS limits of every kernel that needs more than the default, for the current device, once.
hipError_t configure_device_kernels();

// All launchers are asynchronous on `st`.  `n` = frames in the batch; per-frame arrays are laid out
// frame-major with the strides given.

// GPU stage A -------------------------------------------------------------------------------
// Sobel + 16-byte descriptors (filter.cpp:372-416, descriptor.cpp:84-111), fused: desc [2n][H][W] uint4;
// image order L0..L(n-1), R0..R(n-1).
void launch_descriptor(hipStream_t st, const DevParams& dp, const uint8_t* I1, const uint8_t* I2, int32_t in_pitch,
                       int64_t in_stride, int n, uint4* desc);
// The plane data flow (default): the two Sobel responses as byte planes [2n][du | dv][H][plane_pitch(W)], 2 bytes per pixel instead of the
// 16 of a materialised descriptor; the matching kernels assemble the descriptors they stage in LDS from them (kernels.hip).
int plane_pitch(int W);
size_t plane_bytes(int W, int H, int images);
void launch_sobel_planes(hipStream_t st, const DevParams& dp, const uint8_t* I1, const uint8_t* I2, int32_t in_pitch, int64_t in_stride, int n, uint8_t* planes);
// What the matching kernels read descriptors from: the planes (planes = true, Wp = plane_pitch(W)) or the materialised image.
struct DescSrc { const void* ptr; int Wp; bool planes; };
// Support matching with back-check (elas.cpp:269-413): D_can [n][ch][cw] int16.  dry = true launches nothing and returns whether an
// LDS-staged kernel takes these parameters (the only form the plane flow has); otherwise the return value says the same of what ran.
bool launch_support(hipStream_t st, const DevParams& dp, int n, const DescSrc& desc, int16_t* d_can, bool dry = false);

// In-place support-point filters (elas.cpp:153-235 as called at :416-422) on d_can.  Returns false (nothing
// launched) when the lattice does not fit the LDS; the host stage then runs them.
// (uc, vc, d) int16 triples of the support points of each frame in the reference's order + their counts; `list` and
// `count` may be pinned host memory
void launch_support_list(hipStream_t st, const DevParams& dp, int n, const int16_t* d_can, int16_t* list, int32_t* count, int cap);
// Alternating-cut arrangement of the support points of every frame side (what Delaunay::arrange + split compute on the host):
// arr [n][2][arr_cap] vertex numbers, arr_ok [n][2] (0: leave the side to the host: too many points or coinciding vertices).
size_t arrange_lds_bytes(int arr_cap);
// arr_cap: vertices per side this launch orders in LDS (sizes it); sides with more use their slice of gbuf (capacity g_cap per
// side, arrange_scratch_bytes(n, g_cap) bytes in all; may be null) or are left to the host.  arr_stride: layout of arr.
size_t arrange_scratch_bytes(int n, int g_cap);
// What the support points' coordinates range over, for k_arrange's rank form (the orders from bitmaps and prefix counts instead of sorts):
// lattice rows vc in [0, ny), lattice columns uc in [0, nxl), right-image abscissae uc * step - d in [xmin, xmin + nxr).  ny = 0: not known, sorts.
struct ArrBounds { int ny, nxl, xmin, nxr; };
void launch_arrange(hipStream_t st, int n, const int16_t* list, const int32_t* count, int list_cap, int step, int arr_cap, int arr_stride, uint16_t* arr,
                    int32_t* arr_ok, void* gbuf, int g_cap, ArrBounds bnd = ArrBounds{0, 0, 0, 0});
// The hull recursion of the Delaunay triangulation on the GPU (delaunay_gpu.hip): one workgroup per frame side, behind k_arrange.  Writes
// FrameInfo (device), the support points and the triangles' corner indices into the batch payload (frame i at payload_stride * i, laid out
// as HostWorker::place() does); sides it cannot take (too many vertices for cap_pts, coinciding vertices) set need_host[frame].
int delaunay_gpu_capacity(size_t lds_bytes);          // vertices per side that fit
size_t delaunay_gpu_lds_bytes(int points);
hipError_t configure_delaunay_kernel();
int delaunay_gpu_max_points();                         // most vertices a side may have at all (with the global scratch)
size_t delaunay_gpu_scratch_bytes(int frames, int gcap);   // the scratch that lets sides of up to gcap vertices through (subtrees in LDS, the top levels in global memory)
hipError_t launch_delaunay(hipStream_t st, int n, const int16_t* list, const int32_t* count, int list_cap, int step, const uint16_t* arr, const int32_t* arr_ok,
                     int arr_stride, int cap_pts, uint8_t* payload, long long payload_stride, FrameInfo* info, int32_t* need_host,
                     long long* dbg_clock = nullptr,    // dbg_clock (optional, 64 entries): 100 MHz time stamps per tree level of frame 0's sides
                     uint8_t* gscratch = nullptr, int gcap = 0, int expect_pts = 0,
                     bool wide = false);   // wide: coordinates may leave (-2048, 2048) (images of 2048 columns or rows and more): integer predicates
// true when the classify + resolve form of the support filters applies (lattice and codes fit the LDS)
bool support_filters_fast(const DevParams& dp, int win, int min_support);
// list / count / list_cap / listed (optional): where the classification + resolution route takes the lattice, k_filter_resolve also writes
// the support list (launch_support_list's output) and sets *listed
bool launch_support_filters(hipStream_t st, const DevParams& dp, int n, int win, int tol, int min_support, int16_t* d_can,
                            void* scratch, int16_t* list = nullptr, int32_t* count = nullptr, int list_cap = 0, bool* listed = nullptr);

// GPU stage B -------------------------------------------------------------------------------
// Grid prior (elas.cpp:579-659) from the support points: mark/gridbits [n][2][gh*gw][8] uint32.
void launch_grid(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const uint8_t* payload,
                 int64_t payload_stride, int max_sup, uint32_t* mark, uint32_t* gridbits, bool clear = true);
void launch_grid_clear(hipStream_t st, const DevParams& dp, int n, uint32_t* mark);
// the whole grid (clear, mark, dilate) from the GPU's own support list: for parameter sets without corner points, queued behind stage A
void launch_grid_from_list(hipStream_t st, const DevParams& dp, int n, const int16_t* list, const int32_t* count, int cap, uint32_t* mark, uint32_t* gridbits);      // the clear alone (queued ahead by a latency-mode handle)
// Plane fits + edge lines per triangle (elas.cpp:507-577, :847-872): recs [n][2][tri_cap].
void launch_tri_setup(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const uint8_t* payload,
                      int64_t payload_stride, int max_tri, int tri_cap, TriRec* recs);
// Triangle candidates per 32x8 tile with their row masks: bin_count [n][2][tiles], bin_list [n][2][tiles][kBinCap].
// payload != nullptr: the triangles' records are formed on the way (launch_tri_setup's work, same kernel) and written to recs.
void launch_bin(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, TriRec* recs, int tri_cap,
                int max_tri, int32_t* bin_count, BinEntry* bin_list, bool clear = true, const uint8_t* payload = nullptr, int64_t payload_stride = 0);
void launch_bin_clear(hipStream_t st, const DevParams& dp, int n, int32_t* bin_count);
// Dense MAP matching with in-kernel triangle lookup (elas.cpp:683-907): raw [n][2][H][W] float.
// Planes: k_owner (which triangle owns a pixel, its plane's disparity: one 16-bit word per pixel, left in `raw`) + k_dense_row; materialised
// descriptors: k_dense.  dry = true launches nothing and returns whether the plane form takes these parameters.
bool launch_dense(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const TriRec* recs, int tri_cap,
                  const int32_t* bin_count, const BinEntry* bin_list, const uint32_t* gridbits, const DescSrc& desc, int16_t* raw, bool dry = false,
                  hipEvent_t ev_owner = nullptr);   // ev_owner: recorded between k_owner and k_dense_row (timing)
// Left/right consistency (elas.cpp:909-979): raw -> D1, D2 (user buffers).
void launch_lr(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const int16_t* raw, float* D1, float* D2);
// Speckle removal (elas.cpp:981-1099) in place on D [n][H][W]; label/size scratch [n][H][W] int32 each.
void launch_speckle(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, float* D, int32_t* label, int32_t* size,
                    void* scratch);
// launch_lr(raw -> D, D2) + launch_speckle(D) with the L/R check and the speckle pass' row labelling as ONE kernel (the row stays in LDS between them)
void launch_lr_speckle(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const int16_t* raw, float* D, float* D2, int32_t* label, int32_t* size,
                       void* scratch);
// Gap interpolation (elas.cpp:1101-1284): rows D->tmp, columns tmp->D.
void launch_gap(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, float* D, float* tmp);
// Adaptive mean (elas.cpp:1287-1492): horizontal D->tmp, vertical tmp->D.
void launch_adaptive_mean(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, float* D, float* tmp);
// Gap interpolation + adaptive mean as one pass in -> out (different buffers), when gap_mean_fusable(dp).
bool gap_mean_fusable(const DevParams& dp, int n);
void launch_gap_mean_fused(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const float* in, float* out, bool mean);
void launch_copy_ok(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const float* src, float* dst);   // frames with info.ok only
// subsampling = 1: L/R check of the half-size maps picked out of the full-size matcher output (dp = full size); 4-pixel adaptive mean (dph = half size)
void launch_lr_sub(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, const int16_t* raw, float* D1, float* D2);
void launch_adaptive_mean_sub(hipStream_t st, const DevParams& dph, int n, const FrameInfo* info, float* D, float* tmp);
void launch_median(hipStream_t st, const DevParams& dp, int n, const FrameInfo* info, float* D, float* tmp);

// Node side (point_cloud.cpp) -------------------------------------------------------------------
void launch_to_u8(hipStream_t st, const float* D, uint8_t* out, int64_t count);
void launch_valid_lut(hipStream_t st, const jn_scan_params& sp, int W, int H, uint8_t* lut);
// scratch: [n][4] uint64.  If dD != nullptr the u8 map is produced from it first (fused), else dDisp is read.
// lut == nullptr selects the -g flavour (points with d >= 2 minus the ground model, point_cloud.cpp:149-211).
// sgm != nullptr (with lut): the disparities are the SGM mode's winners — left winners dl [n][H][W] (d | d16 << 16, x-mirrored columns) and the
// right image's minr [n][H][W] (S << 16 | d, mirrored): the scan kernel applies the L/R check, writes the int16 map `disp` and the mono8 map
// dDisp on the way (include/jn_sgm.h's definitions of both) and scans what it wrote.
struct SgmWinners { const uint32_t* dl = nullptr; const uint32_t* minr = nullptr; int16_t* disp = nullptr; int lr = -1, subpixel = 0; };
void launch_scan(hipStream_t st, const jn_scan_params& sp, int n, const float* dD, uint8_t* dDisp, const uint8_t* lut,
                 int W, int H, double* bins, double* meta, unsigned long long* scratch, double* flat = nullptr, const SgmWinners* sgm = nullptr);
// Cross-rig merge: pack (bins, meta with maxima negated) into `flat` [n*bins + n*4] or unpack it back.
void launch_scan_pack(hipStream_t st, int n, int bins, double* dBins, double* dMeta, double* flat, bool pack);
// Rectification front end (point_cloud.cpp:440, :481, :553-554).
void launch_undistort_map(hipStream_t st, const double iR[9], const double K[9], const double D[5], int W, int H, float* mapx, float* mapy);
void launch_remap(hipStream_t st, int n, const uint8_t* src, int sw, int sh, int spitch, int64_t sstride, const float* mapx,
                  const float* mapy, uint8_t* dst, int W, int H, int dpitch, int64_t dstride);
// Point cloud (-g): counts per column, exclusive scan, scatter.  col_count: [W+1] int64 scratch.
void launch_point_cloud(hipStream_t st, const jn_scan_params& sp, const uint8_t* disp, int W, int H, float* xyz,
                        long long* col_count);

}  // namespace jnav
