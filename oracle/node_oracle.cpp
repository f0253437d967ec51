// oracle/node_oracle.cpp — TEST INFRASTRUCTURE (CPU restatement), not product code.
//
// Restates the per-frame arithmetic of the reference node that sits after Elas::process
// (reference: src/obstacle_avoidance/point_cloud.cpp).  That file needs ROS + OpenCV and cannot
// be compiled in this image, so these functions are pinned by definition, not by execution:
//   * Mat::convertTo(CV_8U)        -> round-half-to-even + saturate (OpenCV saturate_cast/cvRound)
//   * Mat 4x4*4x1 / 3x3*3x1 double -> plain dot products in row order
//   * everything else              -> the literal C++ of the cited lines.
// PARITY UNPINNED for the OpenCV boundary (no reference test or runnable build exists for it);
// BASELINE.json asks for float reprojection within 1e-4, which this definition satisfies.
//
// One intentional divergence: the reference indexes scan[k] without a bounds
// check and divides by pos.w without a zero check; here pixels with pos.w == 0 are skipped and
// out-of-range bins are not written (their angle/range still enter the min/max, which the
// reference computes before the undefined write).
#include "oracle.h"
#include <cmath>
#include <cstring>
#include <algorithm>

namespace {
const double INF_RANGE = 1e9;   // point_cloud.cpp:54

struct P3 { double x, y, z; };

// point_cloud.cpp:237-253: pos = Q*[i+ox, j+oy, d, 1]; cam = pos.xyz/pos.w; robot = XR*cam + XT
inline bool reproject(const orc_scan_params* sp, int i, int j, int d, P3& out) {
  const double V[4] = {(double)(i + sp->crop_offset_x), (double)(j + sp->crop_offset_y), (double)d, 1.0};
  double pos[4];
  for (int r = 0; r < 4; r++) {
    double s = 0;
    for (int k = 0; k < 4; k++) s += sp->Q[4 * r + k] * V[k];
    pos[r] = s;
  }
  if (pos[3] == 0.0) return false;
  const double c[3] = {pos[0] / pos[3], pos[1] / pos[3], pos[2] / pos[3]};
  double o[3];
  for (int r = 0; r < 3; r++) {
    double s = 0;
    for (int k = 0; k < 3; k++) s += sp->XR[3 * r + k] * c[k];
    o[r] = s + sp->XT[r];
  }
  out.x = o[0]; out.y = o[1]; out.z = o[2];
  return true;
}

// point_cloud.cpp:128-137 / :166-172: true if the point is ground (to be ignored)
inline bool is_ground(const orc_scan_params* sp, double X, double Z) {
  if (X < sp->gp_dist_thresh) return Z < sp->gp_height_thresh;
  return Z < sp->gp_height_thresh + std::tan(sp->gp_angle_thresh) * (X - sp->gp_dist_thresh);
}

struct ScanAcc {
  double* bins; int nb; double amin, amax, rmin, rmax;
  void init(double* b, int n) { bins = b; nb = n; for (int i = 0; i < n; i++) b[i] = INF_RANGE; amin = 400; amax = -400; rmin = INF_RANGE; rmax = -500; }
  // point_cloud.cpp:253-267
  void add(const orc_scan_params* sp, double X, double Y) {
    double th = std::atan2(Y, X);
    double deg = th * 180. / sp->pi_approx;
    amin = std::min(amin, th); amax = std::max(amax, th);
    double r = std::sqrt(Y * Y + X * X);
    rmax = std::max(rmax, r); rmin = std::min(rmin, r);
    double kf = std::floor((double)nb * (sp->fov_deg / 2. - deg) / sp->fov_deg);
    if (!(kf >= 0 && kf < nb)) return;
    int k = (int)kf;
    if (r < bins[k]) bins[k] = r;
  }
  void meta(double* m) { m[0] = amin; m[1] = amax; m[2] = rmin; m[3] = rmax; }
};
}  // namespace

// Q in the analytic zero-disparity form stereoRectify produces, f/cx/cy scaled
// from K1 of calibration/amrl_jackal_webcam_stereo.yml (640x360) to WxH, Tx from T[0]; XR/XT
// from the same file (:39-52).
extern "C" void orc_scan_params_default(orc_scan_params* sp, int32_t W, int32_t H) {
  const double sx = (double)W / 640.0, sy = (double)H / 360.0;
  const double f = 4.6417933392659904e+02 * sx, cx = 3.2479711799310849e+02 * sx, cy = 1.8685472713963392e+02 * sy;
  const double Tx = -9.4052586442980660e-02;
  const double Q[16] = {1, 0, 0, -cx, 0, 1, 0, -cy, 0, 0, 0, f, 0, 0, -1.0 / Tx, 0};
  memcpy(sp->Q, Q, sizeof(Q));
  const double XR[9] = {-0.0007962732853436516, -0.2675000227968607, 0.9635575706420958,
                        -0.9999984502796089, -0.001321509725770019, -0.00119326128710218,
                        0.001592547981999815, -0.9635569909380592, -0.2674985457970802};
  memcpy(sp->XR, XR, sizeof(XR));
  sp->XT[0] = 0; sp->XT[1] = 0; sp->XT[2] = 0.28;
  sp->crop_offset_x = 0; sp->crop_offset_y = 0;
  sp->gp_height_thresh = 0.05; sp->gp_angle_thresh = 4. * 3.1415 / 180.; sp->gp_dist_thresh = 1.0;   // :66-68
  sp->fov_deg = 90.; sp->bins = 90; sp->pi_approx = 3.1415;                                            // :217-218,254
}

// point_cloud.cpp:422
extern "C" void orc_disparity_to_u8(const float* D, uint8_t* out, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    float r = std::nearbyintf(D[i]);              // round-half-even in the default FP environment
    out[i] = (uint8_t)(r < 0.f ? 0 : (r > 255.f ? 255 : (int)r));
  }
}

// point_cloud.cpp:104-147.  lut[...][0] = smallest d in [3,255] whose robot-frame point is above
// the ground model, stored in a uchar so "none" (256) wraps to 0; lut[...][1] = 255.
extern "C" void orc_build_valid_disp_lut(const orc_scan_params* sp, int32_t W, int32_t H, uint8_t* lut) {
  for (int i = 0; i < W; i++)
    for (int j = 0; j < H; j++) {
      int d;
      for (d = 3; d <= 255; d++) {
        P3 p;
        if (!reproject(sp, i, j, d, p)) continue;
        if (p.z < 0.) continue;
        if (is_ground(sp, p.x, p.z)) continue;
        break;
      }
      lut[((size_t)j * W + i) * 2] = (uint8_t)d;
      lut[((size_t)j * W + i) * 2 + 1] = 255;
    }
}

// point_cloud.cpp:213-296
extern "C" int64_t orc_obstacle_scan(const orc_scan_params* sp, const uint8_t* disp, const uint8_t* lut, int32_t W,
                                     int32_t H, double* bins, double* meta4) {
  ScanAcc acc; acc.init(bins, sp->bins);
  int64_t used = 0;
  for (int i = 0; i < W; i++)
    for (int j = 0; j < H; j++) {
      int d = disp[(size_t)j * W + i];
      const uint8_t* l = lut + ((size_t)j * W + i) * 2;
      if (d < l[0] || d > l[1]) continue;
      P3 p;
      if (!reproject(sp, i, j, d, p)) continue;
      acc.add(sp, p.x, p.y); used++;
    }
  acc.meta(meta4);
  return used;
}

// point_cloud.cpp:278-282
extern "C" int32_t orc_compact_ranges(const double* bins, int32_t nb, float* ranges) {
  int32_t n = 0;
  for (int i = nb - 1; i >= 0; i--)
    if (bins[i] < INF_RANGE - 1) ranges[n++] = (float)bins[i];
  return n;
}

// point_cloud.cpp:321-352 (-g): every pixel with d >= 2, i-outer / j-inner, as float32 Point32.
extern "C" int64_t orc_point_cloud(const orc_scan_params* sp, const uint8_t* disp, int32_t W, int32_t H, float* xyz) {
  int64_t n = 0;
  for (int i = 0; i < W; i++)
    for (int j = 0; j < H; j++) {
      int d = disp[(size_t)j * W + i];
      if (d < 2) continue;
      P3 p;
      if (!reproject(sp, i, j, d, p)) continue;
      xyz[3 * n] = (float)p.x; xyz[3 * n + 1] = (float)p.y; xyz[3 * n + 2] = (float)p.z; n++;
    }
  return n;
}

// point_cloud.cpp:149-211
extern "C" int64_t orc_obstacle_scan_points(const orc_scan_params* sp, const double* xyz, int64_t n, double* bins, double* meta4) {
  ScanAcc acc; acc.init(bins, sp->bins);
  int64_t used = 0;
  for (int64_t i = 0; i < n; i++) {
    if (is_ground(sp, xyz[3 * i], xyz[3 * i + 2])) continue;
    acc.add(sp, xyz[3 * i], xyz[3 * i + 1]); used++;
  }
  acc.meta(meta4);
  return used;
}

// -g mode end to end in double (point_cloud.cpp:321-352 then :149-211): the vector<Point3d> the
// reference hands to the scan holds the double robot-frame points, not the float32 message copies.
extern "C" int64_t orc_obstacle_scan_cloud(const orc_scan_params* sp, const uint8_t* disp, int32_t W, int32_t H, double* bins, double* meta4) {
  ScanAcc acc; acc.init(bins, sp->bins);
  int64_t used = 0;
  for (int i = 0; i < W; i++)
    for (int j = 0; j < H; j++) {
      int d = disp[(size_t)j * W + i];
      if (d < 2) continue;
      P3 p;
      if (!reproject(sp, i, j, d, p)) continue;
      if (is_ground(sp, p.x, p.z)) continue;
      acc.add(sp, p.x, p.y); used++;
    }
  acc.meta(meta4);
  return used;
}

// ---- rectification front end (point_cloud.cpp:440, :481, :553-554) — definitions, PARITY UNPINNED ----
// initUndistortRectifyMap: OpenCV's per-pixel formula in double, float maps; iR = inverse(P[:, :3]*R).
extern "C" void orc_init_undistort_rectify_map(const double* K, const double* D, const double* R, const double* P, int32_t W,
                                               int32_t H, float* mapx, float* mapy) {
  double M[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M[3 * i + j] = P[4 * i] * R[j] + P[4 * i + 1] * R[3 + j] + P[4 * i + 2] * R[6 + j];
  const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
  const double id = 1.0 / (M[0] * c00 + M[1] * c01 + M[2] * c02);
  const double iR[9] = {c00 * id, (M[2] * M[7] - M[1] * M[8]) * id, (M[1] * M[5] - M[2] * M[4]) * id,
                        c01 * id, (M[0] * M[8] - M[2] * M[6]) * id, (M[2] * M[3] - M[0] * M[5]) * id,
                        c02 * id, (M[1] * M[6] - M[0] * M[7]) * id, (M[0] * M[4] - M[1] * M[3]) * id};
  const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4], fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
  for (int i = 0; i < H; i++)
    for (int j = 0; j < W; j++) {
      const double _x = j * iR[0] + i * iR[1] + iR[2], _y = j * iR[3] + i * iR[4] + iR[5], _w = j * iR[6] + i * iR[7] + iR[8];
      const double w = 1. / _w, x = _x * w, y = _y * w;
      const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
      const double kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2;
      const double u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0;
      const double v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0;
      mapx[(size_t)i * W + j] = (float)u; mapy[(size_t)i * W + j] = (float)v;
    }
}

// remap INTER_LINEAR / BORDER_CONSTANT 0: 1/32-pixel coordinates (round-half-even), exact 10-bit weights.
extern "C" void orc_remap_bilinear(const uint8_t* src, int32_t sw, int32_t sh, int32_t spitch, const float* mapx, const float* mapy,
                                   uint8_t* dst, int32_t W, int32_t H, int32_t dpitch) {
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const int sx = (int)std::nearbyintf(mapx[(size_t)y * W + x] * 32.0f), sy = (int)std::nearbyintf(mapy[(size_t)y * W + x] * 32.0f);
      const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
      auto tap = [&](int xx, int yy) -> int { return (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? src[(size_t)yy * spitch + xx] : 0; };
      const int acc = (32 - fx) * (32 - fy) * tap(ix, iy) + fx * (32 - fy) * tap(ix + 1, iy) + (32 - fx) * fy * tap(ix, iy + 1) +
                      fx * fy * tap(ix + 1, iy + 1);
      dst[(size_t)y * dpitch + x] = (uint8_t)((acc + 512) >> 10);
    }
}

// ------------------------------------------------------------------------------------------------
// Scan consumer: how navigate.cpp turns the published LaserScan into a stop/turn
// decision.  Parity unpinned like the rest of the node side (ROS absent); restated line by line.
//
// navigate.cpp:344-363 laserScanCallback: ranges[i] sits at angle i*(max-min)/n + min (message floats widened
// to double), point = range * (cos, sin).
extern "C" void orc_scan_to_points(const float* ranges, int32_t n, float angle_min, float angle_max, double* xy) {
  const double minAngle = angle_min, maxAngle = angle_max;
  // The reference is built without optimisation (CMakeLists.txt has no -O flag), so it calls libm's cos and sin
  // separately; an optimising GCC would merge them into one sincos() whose results can differ in the last bit.
  double (*volatile call_cos)(double) = cos;
  double (*volatile call_sin)(double) = sin;
  for (int i = 0; i < n; i++) {
    const double angle = (double)i * (maxAngle - minAngle) / (double)(unsigned)n + minAngle;
    xy[2 * i] = ranges[i] * call_cos(angle);
    xy[2 * i + 1] = ranges[i] * call_sin(angle);
  }
}
// navigate.cpp:101-153 checkObstacle: spatial vote (> laser_pt_thresh points inside the clearance box, or
// anything closer than 0.5 m), then the last-20 history (more than 2 positives => obstacle).
// history: the deque `commands` (oldest first), *len its size.  out: {count, is_obstacle}, closest, confidence.
extern "C" int32_t orc_check_obstacle(const double* xy, int32_t n, int32_t* history, int32_t* len, double clear_front,
                                      double clear_side, int32_t laser_pt_thresh, int32_t* count_out, double* closest_out,
                                      double* conf_out) {
  int count = 0, isObstacle = 0;
  double closestObst = 1000000000;                       // const int INF = 1e9 (:47)
  for (int i = 0; i < n; i++) {
    const double x = xy[2 * i], y = xy[2 * i + 1];
    const double dist = sqrt(x * x + y * y);
    closestObst = std::min(closestObst, dist);
    if (x > 0. && x < clear_front && y > -clear_side && y < clear_side) count++;
  }
  if (count > laser_pt_thresh) isObstacle = 1;
  if (closestObst < 0.5) isObstacle = 1;
  if (*len < 20) history[(*len)++] = isObstacle;
  else {
    for (int i = 1; i < 20; i++) history[i - 1] = history[i];      // pop_front, push_back
    history[19] = isObstacle;
  }
  int one = 0, zero = 0;
  for (int i = 0; i < *len; i++) { if (history[i] == 1) one++; else zero++; }
  if (one > 2) isObstacle = 1;
  *count_out = count; *closest_out = closestObst; *conf_out = (double)one / (double)(one + zero);
  return isObstacle;
}
// navigate.cpp:155-197 chooseDirection: side with fewer points in front wins, with hysteresis on last_dir.
extern "C" int32_t orc_choose_direction(const double* xy, int32_t n, double clear_front, int32_t last_dir) {
  int left_count = 0, right_count = 0;
  for (int i = 0; i < n; i++)
    if (xy[2 * i] > 0. && xy[2 * i] < clear_front) { if (xy[2 * i + 1] < 0) right_count++; else left_count++; }
  if (left_count + right_count < 2) return 0;
  const double conf_left = 2. * (double)right_count / (double)(left_count + right_count);
  const double conf_right = 2. * (double)left_count / (double)(left_count + right_count);
  int dir = 0;
  if (conf_left > conf_right) {
    if (last_dir != 1) dir = (conf_left - conf_right > 0.5) ? 1 : last_dir;
    else dir = 1;
  } else {
    if (last_dir != 2) dir = (conf_right - conf_left > 0.5) ? 2 : last_dir;
    else dir = 2;
  }
  return dir;
}
