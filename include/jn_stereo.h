/* jn_stereo.h — C ABI of libjn_stereo.so: the MI355X (gfx950) replacement for the per-frame hot
 * path of jackal_nav's `point_cloud` node.
 *
 * Every entry point names the reference interface it replaces (paths relative to the reference
 * repository sourishg/jackal-navigation).  Plain pointers and sizes only; no C++/torch types.
 * "device pointer" = memory of the GPU the handle was created on (hipMalloc or a torch tensor's
 * data_ptr()); everything else is host memory.
 *
 * Seam B1  Elas::parameters / Elas::process          src/elas/elas.h:59-162, elas.cpp:32-151
 * Seam B2  convertTo(CV_8U), cacheDisparityValues,    src/obstacle_avoidance/point_cloud.cpp:422,
 *          publishObstacleScan(Mat&), publishPointCloud   :104-147, :213-296, :298-404
 *
 * There is no CPU fallback: every compute entry point returns JN_ERR_NO_DEVICE when no gfx950
 * device is usable.
 */
#ifndef JN_STEREO_H
#define JN_STEREO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum jn_status {
  JN_OK = 0,
  JN_ERR_FEW_SUPPORT = 1,   /* elas.cpp:66-71 "Need at least 3 support points": outputs left untouched */
  JN_ERR_UNSUPPORTED = 2,   /* parameter combination outside the HIP path (see jn_elas_create) */
  JN_ERR_INVALID = 3,       /* bad argument */
  JN_ERR_NO_DEVICE = 4,     /* no usable HIP device / HIP runtime failure */
  JN_ERR_INTERNAL = 5,
  JN_ERR_COMM = 6           /* RCCL missing or a collective failed (jn_comm_*) */
} jn_status;

/* ---- Seam B1 ------------------------------------------------------------------------------ */

/* Field-for-field mirror of Elas::parameters (elas.h:60-82); bools widened to int32. */
typedef struct jn_elas_params {
  int32_t disp_min;
  int32_t disp_max;
  float   support_threshold;
  int32_t support_texture;
  int32_t candidate_stepsize;
  int32_t incon_window_size;
  int32_t incon_threshold;
  int32_t incon_min_support;
  int32_t add_corners;
  int32_t grid_size;
  float   beta;
  float   gamma;
  float   sigma;
  float   sradius;
  int32_t match_texture;
  int32_t lr_threshold;
  float   speckle_sim_threshold;
  int32_t speckle_size;
  int32_t ipol_gap_width;
  int32_t filter_median;
  int32_t filter_adaptive_mean;
  int32_t postprocess_only_left;
  int32_t subsampling;
} jn_elas_params;

#define JN_SETTING_ROBOTICS   0
#define JN_SETTING_MIDDLEBURY 1

/* Elas::parameters::parameters(setting) — elas.h:85-145. */
void jn_elas_params_default(jn_elas_params* p, int32_t setting);

typedef struct jn_elas jn_elas;   /* opaque; replaces an `Elas` object (elas.h:148) */

/* Replaces `Elas elas(param)` (point_cloud.cpp:416-418), but long-lived: all device and pinned
 * buffers for up to `max_batch` WxH pairs per pipeline slot are allocated once.
 *   device        HIP device ordinal
 *   host_threads  worker threads for the host stage (the hull recursion of the Delaunay triangulations; the support
 *                 filters only where no kernel takes the lattice); 0 = one per CPU the process may use (affinity
 *                 mask, cut down to the container's cgroup CPU quota); 16 feed one MI355X at 22 k 720p pairs/s.
 *                 A batch handle (max_batch > 1) of a process pinned to 16 cores or fewer (or with a CPU quota below 14, or created
 *                 with 0 < host_threads < 14: the caller's share of cores that several ranks divide) has NO host
 *                 stage: the hull recursion runs on the GPU as well (19.6 k pairs/s with 0.3 busy cores; JN_GPU_DELAUNAY=0/1 decides
 *                 otherwise) and the pool idles; a batch the GPU cannot triangulate (coinciding right-image vertices, more than ~3 700
 *                 support points a side) goes through the host stage after all.
 *   slots         pipeline depth for jn_elas_submit (>=1); each slot has its own stream/buffers
 * Unsupported (JN_ERR_UNSUPPORTED): subsampling with an odd width or height, disp_max > 255 or < 10, disp_min > disp_max,
 * candidate_stepsize < 1, grid_size < 1, plane radius > 7.  Both presets of elas.h:92-145 are supported; disp_min is honoured as the
 * reference does (elas.cpp:323-333: the first disparity the support matching tries; negative values act as 0; nothing else reads it).
 * subsampling = 1 (elas.h:82, :160-162): D1 / D2 are (W/2) x (H/2) maps — per frame (W/2)*(H/2) floats, frames back to back — holding the
 * disparities of the even pixels of the even rows, post-processed at that size as the reference does (elas.cpp:914-941, :987-992,
 * :1107-1112, :1323-1391); the scan tail (jn_elas_submit_scan) is refused for such a handle (the node never subsamples: its Q matrix
 * and LUT are the full image's).  Odd sizes are refused because the reference's (u/2, v/2, width/2) addressing runs over its rows there.
 * UNINITIALISED BYTES, both presets: the reference never writes descriptor columns 0..2 and W-3..W-1 (descriptor.cpp:29,
 * :84-88) but reads column W-3 in the right-image support match (elas.cpp:326, :340-349) and columns 2 and W-3 in
 * findMatch (elas.cpp:744-746, :752-754, :763-765, :770-772), so its D1/D2 depend on what malloc returned — with the node's
 * ROBOTICS parameters too (tens to hundreds of pixels per frame), and far more with add_corners (MIDDLEBURY).  This
 * library defines those bytes as 0, which is what the reference sees in freshly mapped memory; all parity statements are
 * against the reference run with zero-filled allocations.
 * On any failure everything allocated so far is released and *out stays NULL. */
jn_status jn_elas_create(const jn_elas_params* p, int32_t width, int32_t height, int32_t max_batch,
                         int32_t device, int32_t host_threads, int32_t slots, jn_elas** out);
void jn_elas_destroy(jn_elas* h);

/* Drop-in for Elas::process (elas.h:154-162, elas.cpp:32).  Host pointers; synchronous.
 * dims = {width, height, bytes per line}; D1/D2 are caller-allocated width*height floats.
 * Valid disparities are >= 0, invalid ones are -10.  On JN_ERR_FEW_SUPPORT D1/D2 are left
 * untouched exactly as the reference does (it prints an error and returns). */
jn_status jn_elas_process(jn_elas* h, const uint8_t* I1, const uint8_t* I2, float* D1, float* D2,
                          const int32_t dims[3]);

/* Batched form, device pointers: n rectified pairs, image b at dI + b*image_stride (bytes, rows
 * `pitch` bytes apart); outputs dD + b*width*height floats.  status[b] (host, may be NULL)
 * receives JN_OK or JN_ERR_FEW_SUPPORT per pair.  Synchronous on slot 0. */
jn_status jn_elas_process_batch(jn_elas* h, int32_t n, const uint8_t* dI1, const uint8_t* dI2,
                                int32_t pitch, int64_t image_stride, float* dD1, float* dD2,
                                int32_t* status);

/* Pipelined form: enqueue a batch on `slot` and return; the slot's worker runs
 * GPU stage A -> host stage -> GPU stage B.  jn_elas_wait blocks until that slot is idle and
 * returns the batch's status (per-pair codes in the `status` array given to submit). */
jn_status jn_elas_submit(jn_elas* h, int32_t slot, int32_t n, const uint8_t* dI1, const uint8_t* dI2,
                         int32_t pitch, int64_t image_stride, float* dD1, float* dD2, int32_t* status);
jn_status jn_elas_wait(jn_elas* h, int32_t slot);

/* jn_elas_submit for HOST buffers (pageable or pinned): n pairs at I + b*image_stride, rows `pitch` bytes apart; D1 / D2
 * [n][height][width] floats.  The slot's worker copies the images in, runs the batch and copies the maps out before
 * jn_elas_wait returns, so that with several slots the copies of one batch overlap the kernels of the others — the
 * streaming form of Elas::process for a caller that has the next frames at hand (the synchronous one-pair drop-in is
 * jn_elas_process below).  A pair with too few support points leaves its D1 / D2 untouched (elas.cpp:66-71) and has
 * status JN_ERR_FEW_SUPPORT; the buffers must stay valid until the wait. */
jn_status jn_elas_submit_host(jn_elas* h, int32_t slot, int32_t n, const uint8_t* I1, const uint8_t* I2,
                              int32_t pitch, int64_t image_stride, float* D1, float* D2, int32_t* status);

/* Per-stage timings of the last batch on a slot, milliseconds (reference stage names,
 * elas.cpp:54-144 PROFILE labels + JackalTimeLog fields msg/JackalTimeLog.msg:1-4).  The gpu_*, d2h and h2d entries come
 * from timing events between the stages; each costs a few microseconds of idle GPU, so a handle created with max_batch 1
 * (latency mode) leaves them out and reports 0 there — host_stage and total are always measured.  JN_STAGE_EVENTS=1 / 0 at
 * create time overrides either way. */
typedef struct jn_stage_times {
  float gpu_descriptor, gpu_support, d2h, host_stage, h2d, gpu_matching, gpu_lr, gpu_speckle,
        gpu_gap, gpu_adaptive_mean, total;
} jn_stage_times;
jn_status jn_elas_last_times(jn_elas* h, int32_t slot, jn_stage_times* out);

/* ---- Seam B2 ------------------------------------------------------------------------------ */

/* The file-scope state of point_cloud.cpp that the scan functions read (:28-69, :217-218). */
typedef struct jn_scan_params {
  double  Q[16];            /* 4x4 row-major, stereoRectify output (point_cloud.cpp:543-544) */
  double  XR[9];            /* camera->robot rotation (calibration yml, point_cloud.cpp:537) */
  double  XT[3];            /* camera->robot translation (:538) */
  int32_t crop_offset_x;    /* :51 */
  int32_t crop_offset_y;    /* :52 */
  double  gp_height_thresh; /* GP_HEIGHT_THRESH :66 */
  double  gp_angle_thresh;  /* GP_ANGLE_THRESH  :67 */
  double  gp_dist_thresh;   /* GP_DIST_THRESH   :68 */
  double  fov_deg;          /* :217 */
  int32_t bins;             /* :218, <= 1024 */
  double  pi_approx;        /* the literal 3.1415 of :254 */
} jn_scan_params;

/* Defaults: reference constants, XR/XT of calibration/amrl_jackal_webcam_stereo.yml:39-52 and the
 * analytic zero-disparity Q for K1/T of that file scaled from 640x360 to width x height. */
void jn_scan_params_default(jn_scan_params* sp, int32_t width, int32_t height);

#define JN_SCAN_EMPTY 1e9   /* `INF` of point_cloud.cpp:54: value of a bin no pixel fell into */

/* leftdpf.convertTo(show, CV_8U, 1.) (point_cloud.cpp:422): round-half-even, saturate; -10 -> 0.
 * Device pointers, n elements. */
jn_status jn_disparity_to_u8(int32_t device, const float* dD, uint8_t* dOut, int64_t n);

/* cacheDisparityValues() (point_cloud.cpp:104-147): dLut is [height][width][2] uint8 (CV_8UC2). */
jn_status jn_build_valid_disp_lut(int32_t device, const jn_scan_params* sp, int32_t width, int32_t height,
                                  uint8_t* dLut);

/* publishObstacleScan(Mat&) (point_cloud.cpp:213-296) for n u8 disparity maps (device,
 * contiguous).  Outputs, device pointers: dBins [n][bins] doubles, un-compacted, JN_SCAN_EMPTY =
 * no return; dMeta [n][4] = angle_min, angle_max, range_min, range_max (initial values 400, -400,
 * 1e9, -500 as at :219-220 when no pixel qualifies).
 * Divergence from the reference (which has undefined behaviour there): pixels whose homogeneous
 * w is 0 are skipped and bins outside [0,bins) are not written. */
jn_status jn_obstacle_scan(int32_t device, const jn_scan_params* sp, int32_t n, const uint8_t* dDisp,
                           const uint8_t* dLut, int32_t width, int32_t height, double* dBins, double* dMeta);

/* The -g flavour: publishPointCloud (point_cloud.cpp:321-352) followed by
 * publishObstacleScan(vector<Point3d>) (:149-211) without materialising the cloud: every pixel with
 * d >= 2 is reprojected (double), points on the ground model (:166-172) are dropped, the rest are
 * binned.  Same outputs and divergence note as jn_obstacle_scan. */
jn_status jn_obstacle_scan_cloud(int32_t device, const jn_scan_params* sp, int32_t n, const uint8_t* dDisp,
                                 int32_t width, int32_t height, double* dBins, double* dMeta);

/* Fused tail of the node: float disparity -> u8 map (published on /webcam/left/depth_map) -> scan. */
jn_status jn_disparity_scan(int32_t device, const jn_scan_params* sp, int32_t n, const float* dD,
                            const uint8_t* dLut, int32_t width, int32_t height, uint8_t* dDispU8,
                            double* dBins, double* dMeta);

/* jn_elas_submit followed by the node's tail on the same stream: convertTo(CV_8U) of D1 (point_cloud.cpp:422) and
 * publishObstacleScan(Mat&) (:213-296) with the cached LUT — i.e. what jn_disparity_scan does, without a second
 * call or a host round trip in between.  After jn_elas_wait: dD1/dD2 as for jn_elas_submit, dDispU8 [n][H][W],
 * dBins [n][bins], dMeta [n][4].  Frames that fail (status != 0) leave dD1 untouched, and the tail then scans
 * whatever dD1 held (the reference publishes its zero-initialised map in that case, point_cloud.cpp:414-428). */
jn_status jn_elas_submit_scan(jn_elas* h, int32_t slot, int32_t n, const uint8_t* dI1, const uint8_t* dI2, int32_t pitch,
                              int64_t image_stride, float* dD1, float* dD2, const jn_scan_params* sp, const uint8_t* dLut,
                              uint8_t* dDispU8, double* dBins, double* dMeta, int32_t* status);

/* The `ranges` compaction of point_cloud.cpp:278-282 (host): bins < 1e9-1, pushed from the last
 * bin to the first, as float32.  Returns the count. */
int32_t jn_compact_ranges(const double* bins, int32_t nbins, float* ranges);

/* publishPointCloud -g (point_cloud.cpp:321-352): every pixel with d >= 2 as a robot-frame
 * float32 xyz triple, in the reference's i-outer / j-inner order.  dXyz must hold
 * width*height*3 floats; *count (host) receives the number of points. */
jn_status jn_point_cloud(int32_t device, const jn_scan_params* sp, const uint8_t* dDisp, int32_t width,
                         int32_t height, float* dXyz, int64_t* count);

/* ---- rectification front end (OpenCV-side arithmetic: parity UNPINNED) ---- */

/* The matrices `main` reads from the calibration YAML (point_cloud.cpp:530-536). */
typedef struct jn_stereo_calib {
  double K1[9], D1[5], K2[9], D2[5];   /* camera matrices, Brown distortion (k1,k2,p1,p2,k3) */
  double R[9], T[3];                   /* pose of camera 2 w.r.t. camera 1 */
  int32_t calib_width, calib_height;   /* calib_im_size, point_cloud.cpp:38 */
} jn_stereo_calib;
typedef struct jn_rectification { double R1[9], R2[9], P1[12], P2[12], Q[16]; } jn_rectification;

/* calibration/amrl_jackal_webcam_stereo.yml:1-37 */
void jn_stereo_calib_default(jn_stereo_calib* c);

/* cv::stereoRectify(K1,D1,K2,D2,calib_size,R,T, R1,R2,P1,P2,Q, CV_CALIB_ZERO_DISPARITY, alpha=0,
 * newImageSize) as called at point_cloud.cpp:543-544.  Host, double. */
jn_status jn_stereo_rectify(const jn_stereo_calib* c, int32_t new_width, int32_t new_height, jn_rectification* out);

/* cv::initUndistortRectifyMap(K, D, R, P, size, CV_32F, mapx, mapy) (point_cloud.cpp:553-554).
 * dMapX/dMapY: device float [height][width]. */
jn_status jn_init_undistort_rectify_map(int32_t device, const double K[9], const double D[5], const double R[9],
                                        const double P[12], int32_t width, int32_t height, float* dMapX, float* dMapY);

/* cv::remap(src, dst, mapx, mapy, INTER_LINEAR) with BORDER_CONSTANT 0 (point_cloud.cpp:440, :481)
 * for n grey source frames (device).  Source coordinates are quantised to 1/32 pixel
 * (round-half-even, as OpenCV does); the four taps are blended with exact 10-bit weights
 * ((32-fx)(32-fy) ...)/1024, rounded to nearest.
 * This IS cv::remap's 8-bit INTER_LINEAR arithmetic (imgwarp.cpp, OpenCV 2.4 / 3 / 4): its 32x32 table holds the four
 * weights scaled by INTER_REMAP_COEF_SCALE = 32768 and rounded to short — but (32-fx)(32-fy)/1024 * 32768 = 32 (32-fx)(32-fy)
 * is an integer, so nothing is rounded and every quadruple sums to 32768 — except the phase (0, 0), whose weight 32768 a short
 * cannot hold: OpenCV stores 32767 and its fix-up adds the missing unit to another tap, which never changes an 8-bit pixel
 * ((32767 p + p' + 16384) >> 15 == p).  Its final FixedPtCast, (sum + 16384) >> 15 with sum = 32 acc, equals (acc + 512) >> 10.
 * tests/test_node_oracle.py rebuilds the table by OpenCV's published recipe and checks all 1024 phases and both identities.  What stays unpinned (no
 * OpenCV in the build image) is only that the recipe is restated, not linked. */
jn_status jn_remap_bilinear(int32_t device, int32_t n, const uint8_t* dSrc, int32_t src_width, int32_t src_height,
                            int32_t src_pitch, int64_t src_stride, const float* dMapX, const float* dMapY,
                            uint8_t* dDst, int32_t width, int32_t height, int32_t dst_pitch, int64_t dst_stride);

/* cv::imdecode(Mat(msg->data), CV_LOAD_IMAGE_GRAYSCALE) (point_cloud.cpp:436, :478) for the node's JPEG camera frames:
 * what OpenCV gets from libjpeg with a grey output colour space, i.e. the luminance plane reconstructed with the "slow
 * integer" inverse DCT (third-party arithmetic outside the reference tree, restated from the JPEG standard and the IJG
 * definition of that IDCT; pinned by Pillow / libjpeg-turbo fixtures, tests/golden/make_jpeg_golden.py).  Entropy decoding
 * runs on the calling thread, dequantisation + IDCT on the GPU; the image lands in device memory (dOut, rows out_pitch
 * bytes apart, at least out_rows rows) ready for jn_remap_bilinear.  Baseline / extended-sequential Huffman JPEG with 8-bit
 * samples, 1 or 3 components in one scan, restart intervals; anything else (progressive, arithmetic, 12-bit) returns
 * JN_ERR_UNSUPPORTED, damaged data JN_ERR_INVALID.  jn_jpeg_info reads the frame size without decoding (host only). */
jn_status jn_jpeg_info(const uint8_t* jpeg, int64_t nbytes, int32_t* width, int32_t* height);
jn_status jn_jpeg_decode_gray(int32_t device, const uint8_t* jpeg, int64_t nbytes, uint8_t* dOut, int32_t out_pitch, int32_t out_rows,
                              int32_t* width, int32_t* height);
/* Both eyes of one stereo frame (the reference decodes them in its two image callbacks, point_cloud.cpp:436 and :478): the two
 * entropy decodes run on two threads (a long-lived helper per calling thread takes the right eye), the two inverse DCTs are
 * queued together and waited for once.  Same results and errors as two jn_jpeg_decode_gray calls; both frames must have the
 * same size, returned in *width / *height. */
jn_status jn_jpeg_decode_gray_pair(int32_t device, const uint8_t* jpegL, int64_t nbytesL, const uint8_t* jpegR, int64_t nbytesR, uint8_t* dOutL,
                                   uint8_t* dOutR, int32_t out_pitch, int32_t out_rows, int32_t* width, int32_t* height);

/* Host-stage hook (CPU only, like jn_host_triangulate): the entropy-decoded luminance coefficients of a JPEG frame, natural
 * (de-zigzagged) order, not dequantised, blocks_h x blocks_w blocks of 64 (padded to whole MCUs), and the luminance
 * quantisation table.  Returns the number of coefficients, or -(jn_status) on error; coef may be NULL to query sizes. */
int64_t jn_host_jpeg_coefficients(const uint8_t* jpeg, int64_t nbytes, int16_t* coef, int64_t coef_capacity, uint16_t quant[64],
                                  int32_t* width, int32_t* height, int32_t* blocks_w, int32_t* blocks_h);

/* ---- cross-rig merge (SURVEY.md 8b `jn_scan_allreduce`, 8e) ---------------------------------
 * The path's one exchange step: per-rig obstacle scans -> robot-level scan = element-wise MIN over
 * the bins (point_cloud.cpp:264-266 applied across rigs) and min / max / min / max of the four
 * LaserScan extrema (:255-260), followed on every rank by the compaction of :278-282
 * (jn_compact_ranges).  One rank (= one process) per GPU; the collective is RCCL's
 * ncclAllReduce(ncclMin, ncclDouble) over xGMI on ONE packed buffer per batch (maxima travel
 * negated), i.e. a single latency-bound message of n*(bins+4)*8 bytes.  librccl.so.1 is bound at
 * run time (dlopen; the copy a host process already loaded, e.g. PyTorch's, is re-used), so the
 * library itself has no link-time dependency on RCCL. */
#define JN_COMM_ID_BYTES 128
typedef struct jn_comm jn_comm;
/* ncclGetUniqueId: rank 0 creates the id and hands it to the other ranks by any side channel. */
jn_status jn_comm_unique_id(uint8_t id[JN_COMM_ID_BYTES]);
/* ncclCommInitRank on `device`; collective over all `world` ranks. */
jn_status jn_comm_create(const uint8_t id[JN_COMM_ID_BYTES], int32_t rank, int32_t world, int32_t device, jn_comm** out);
/* What RCCL itself reports for the communicator (ncclCommUserRank / ncclCommCount / ncclCommCuDevice). */
jn_status jn_comm_info(jn_comm* c, int32_t* rank, int32_t* world, int32_t* device);
/* In place on device memory of the communicator's GPU: dBins [n][bins] doubles and dMeta [n][4] as
 * jn_obstacle_scan writes them; afterwards every rank holds the merged scans.  The inputs must be
 * complete (e.g. after jn_elas_wait); returns when the merged values are in place. */
jn_status jn_scan_allreduce(jn_comm* c, int32_t n, int32_t bins, double* dBins, double* dMeta);
/* The merge as the TAIL OF EVERY SCAN BATCH: with a communicator attached, jn_elas_submit_scan's batches finish with
 * ncclAllReduce(MIN) -> unpack on the buffer the scan kernel packed, queued by the slot's WORKER thread once the scan is complete;
 * the worker (not the submitting thread) waits for its turn and for the merged bins, so jn_elas_wait returns with the
 * ROBOT-level bins in dBins / dMeta.  RCCL needs every rank to issue a communicator's collectives in one order: batches queue
 * their merges in submission order, so every rank must submit the same sequence of scan batches (same n, same bins).
 * Failure behaviour: a batch that fails on THIS rank before its merge still takes its turn and contributes the identity of the
 * reduction (+inf), so the peers are not left inside the collective — they get the other rigs' scan, this rank's jn_elas_wait
 * reports the batch's error.  A merge that does not complete within JN_COMM_TIMEOUT_MS (default 30000; 0 = wait for ever) —
 * a peer died or never issued its collective — is aborted (ncclCommAbort), the handle's communicator is marked dead and the
 * batch, like every later scan batch of the handle, returns JN_ERR_COMM instead of hanging.
 * Do not mix: while a communicator is attached to a handle, jn_scan_allreduce must not be called on it from other threads
 * unless every rank interleaves the two in the same order (the collectives of one communicator are totally ordered).
 * Call with no batch in flight; c = NULL detaches.  The communicator must live on the handle's device and outlive its use
 * here.  jn_elas_merge_time: milliseconds on the worker's clock from the slot's last scan being complete to its merged bins
 * being in place (waiting for its turn in the submission order included). */
jn_status jn_elas_set_comm(jn_elas* h, jn_comm* c);
/* Testing aid: the submission numbers of the last scan batches of `h` in the order their merges were QUEUED (at most `cap`,
 * oldest first; returns how many were written).  In a correct run this is 0, 1, 2, ... whatever order the batches' host
 * stages finished in. */
int32_t jn_elas_merge_order(jn_elas* h, uint64_t* out, int32_t cap);
/* Testing aid: triangle lists of the 32x8 tiles of `slot`'s last batch (call with no batch in flight on it; frames that failed count with
 * whatever their tiles held): out[0] = the longest list, out[1] = tiles whose list is longer than the 16 entries the ownership pass
 * resolves by ranked cover words (they take its entry-by-entry form), out[2] = tiles whose list overflowed its 64 entries (scan over all
 * of the side's triangles).  Tests use it to show that a scene really took those routes. */
jn_status jn_elas_bin_stats(jn_elas* h, int32_t slot, int32_t out[3]);
/* Testing aid: which routes the handle takes.  out[0] = 1 when its batches triangulate on the GPU (delaunay_gpu.hip: batch handles of
 * processes with fewer than 14 cores of their own, or JN_GPU_DELAUNAY=1; no host stage then), out[1] = batches of `slot` that went through the
 * host stage after all because the GPU handed a side back (coinciding vertices, too many vertices for the LDS), out[2] = 1 when descriptors
 * are assembled from the Sobel planes (the default) rather than materialised. */
jn_status jn_elas_route_stats(jn_elas* h, int32_t slot, int32_t out[3]);
jn_status jn_elas_merge_time(jn_elas* h, int32_t slot, float* ms);
void jn_comm_destroy(jn_comm* c);

/* ---- utilities ---------------------------------------------------------------------------- */

/* Synthetic rectified pair of the benchmark (SURVEY.md Appendix A generator), host buffers. */
void jn_synth_pair(int32_t width, int32_t height, int32_t scene_disp, uint32_t seed, uint8_t* L, uint8_t* R);

/* Device helpers so that non-torch callers (tests, C hosts) need no HIP headers. */
jn_status jn_device_count(int32_t* count);
jn_status jn_device_malloc(int32_t device, int64_t bytes, void** out);
jn_status jn_device_free(int32_t device, void* p);
jn_status jn_memcpy_h2d(int32_t device, void* dst, const void* src, int64_t bytes);
jn_status jn_memcpy_d2h(int32_t device, void* dst, const void* src, int64_t bytes);
jn_status jn_device_synchronize(int32_t device);

/* Timing of one kernel class with HIP events on the library's own stream, for bench.py's
 * roofline block: average milliseconds per launch during the last batch of `slot` (both sides), and
 * launches counted.  kernel: "k_dense_row" (or "k_dense": the dense matcher of the handle's data
 * flow), "k_owner" (the ownership pass in front of k_dense_row; 0 for a handle on materialised descriptors). */
jn_status jn_elas_kernel_time(jn_elas* h, int32_t slot, const char* kernel, float* avg_ms, int32_t* launches);

const char* jn_version(void);

/* FNV-1a-64 over 32-bit words (host memory): the hash SURVEY.md 8c quotes for D1/D2 float maps. */
uint64_t jn_fnv1a64_u32(const uint32_t* words, int64_t n);

/* ---- host-stage hooks (CPU only; no device needed) ------------------------------------------
 * The serial middle of ELAS runs on host threads between the two GPU stages.  These two entry
 * points expose it so it can be verified, and reused by other hosts, without a GPU. */

/* Triangle's "zQB" divide-and-conquer result (src/elas/triangle.cpp:8499 as called from
 * elas.cpp:487-488) for n integer points: writes (org,dest,apex) per triangle into tri
 * (capacity 6*n ints); returns the triangle count or -1. */
int32_t jn_host_triangulate(const int32_t* x, const int32_t* y, int32_t n, int32_t* tri);
/* The same with the triangulation cut into `parts` (1, 2 or 4) independent pieces that run on their own threads and
 * are merged afterwards — what jn_elas does when its pool has idle threads (a lone pair, a few large frames).  The
 * output, order included, equals jn_host_triangulate's. */
int32_t jn_host_triangulate_parts(const int32_t* x, const int32_t* y, int32_t n, int32_t* tri, int32_t parts);

/* Support filters + support list + Delaunay x2 for ONE frame (elas.cpp:416-431, :445-505).
 * d_can [ch][cw] is filtered in place.  payload receives what the GPU stage consumes:
 *   [nsup x (u,v,d) int32 at sup_offset][ntri[s] x 3 int32 corner indices at corner_offset[s]]
 * (plane fits and the grid prior, elas.cpp:507-659, are computed on the GPU from these).
 * Returns bytes used or -1 if payload_cap is too small. */
typedef struct jn_host_frame_info {
  int32_t ok, nsup, ntri[2];
  int64_t sup_offset;
  int64_t corner_offset[2];
  int64_t reserved;
} jn_host_frame_info;
int64_t jn_host_stage(const jn_elas_params* p, int32_t width, int32_t height, int16_t* d_can, uint8_t* payload,
                      int64_t payload_cap, jn_host_frame_info* info);

/* The alternating-cut arrangement the triangulation starts from (triangle.cpp:5514-5606, :6197-6206): on the host (what
 * jn_host_triangulate uses; returns 1, or 0 when vertices coincide and Triangle's own sort has to be replayed) and on the GPU
 * from a support list of (uc, vc, d) triples with coordinates x = uc*step (left) / uc*step - d (right), y = vc*step
 * (kernels.hip k_arrange; ok[side] = 0 when the side is left to the host).  Exposed so that the two can be compared. */
int32_t jn_host_arrangement(const int32_t* x, const int32_t* y, int32_t n, uint16_t* out);
jn_status jn_device_arrangement(int32_t device, const int16_t* triples, int32_t n, int32_t step, uint16_t* left, uint16_t* right,
                                int32_t ok[2]);
/* The whole triangulation on the GPU (k_arrange + k_delaunay, delaunay_gpu.hip: what a batch handle runs instead of the host stage) for one
 * frame's support list: the triangles of the left side (vertices (uc*step, vc*step)) and of the right side ((uc*step - d, vc*step)), (org,
 * dest, apex) per triangle in Triangle's output order, capacity 6*n ints each; ntri[side] = how many; *need_host = bit mask of the sides
 * the GPU handed back (too many vertices for the LDS, coinciding vertices).  Exposed so that it can be compared with jn_host_triangulate. */
jn_status jn_device_triangulate(int32_t device, const int16_t* triples, int32_t n, int32_t step, int32_t* tri_left, int32_t* tri_right, int32_t ntri[2],
                                int32_t* need_host);

/* The GPU's support filters on their own (elas.cpp:416-422 on n candidate lattices [n][ch][cw], host memory, filtered in
 * place), exposed so that the kernels can be verified on arbitrary lattices.  form: 0 = whatever jn_elas would use,
 * 1 = skewed wavefront, 2 = classification + resolution.  Returns JN_ERR_UNSUPPORTED when no kernel takes the lattice
 * (jn_elas then filters on the host). */
jn_status jn_device_support_filters(int32_t device, const jn_elas_params* p, int32_t width, int32_t height, int32_t n,
                                    int16_t* d_can, int32_t form);

/* ---------------------------------------------------------------------------------------------
 * Scan consumer: the decision navigate.cpp takes from one LaserScan.  Host code;
 * present so that scans from this library can be checked to drive the same stop / turn decisions.
 * The reference keeps this state in file-scope globals (navigate.cpp:22-24, :44-45); here the caller
 * owns it. */
#define JN_NAV_MAX_HISTORY 64
typedef struct jn_nav_params {
  double clear_front;        /* navigate.cpp:37  0.24 + 0.8 */
  double clear_side;         /* :38  0.3 */
  double stop_dist;          /* :125 0.5: any return closer than this is an obstacle */
  int32_t laser_pt_thresh;   /* :42  8 */
  int32_t history;           /* :129 20 votes kept */
  int32_t history_votes;     /* :146 more than 2 positives in the history => obstacle */
  int32_t reserved;
} jn_nav_params;
typedef struct jn_nav_state {
  int32_t votes[JN_NAV_MAX_HISTORY];
  int32_t head, filled, positives;
  int32_t last_dir;          /* :45 */
} jn_nav_state;
typedef struct jn_nav_decision {
  int32_t points_inside;     /* `count` of :108-111 */
  int32_t points;            /* laserPoints.size() */
  int32_t obstacle;          /* checkObstacle's return value */
  int32_t direction;         /* 0 none, 1 left, 2 right (chooseDirection; 0 when no obstacle, :252) */
  double closest;            /* closestObst */
  double confidence;         /* `conf` of :149 */
} jn_nav_decision;
void jn_nav_params_default(jn_nav_params* p);
void jn_nav_state_reset(jn_nav_state* s);
/* laserScanCallback (navigate.cpp:344-363): ranges[i] at angle i*(max-min)/n + min -> xy[2*i], xy[2*i+1]. */
int32_t jn_scan_to_points(const float* ranges, int32_t n, float angle_min, float angle_max, double* xy);
/* checkObstacle (:101-153) followed by obstacleAvoidMode's use of chooseDirection (:232-235, :252). */
jn_status jn_nav_vote(const jn_nav_params* p, jn_nav_state* s, const double* xy, int32_t n, jn_nav_decision* out);

#ifdef __cplusplus
}
#endif
#endif /* JN_STEREO_H */
