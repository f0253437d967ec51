// rectify.cpp — init-time rectification geometry of the node (product code, host, double).
//
// The reference calls OpenCV (point_cloud.cpp:543-544, :553-554):
//     stereoRectify(K1, D1, K2, D2, calib_im_size, R, T, R1, R2, P1, P2, Q, CV_CALIB_ZERO_DISPARITY, 0, rawimsize, ...)
//     initUndistortRectifyMap(K, D, R1|R2, P1|P2, rawimsize, CV_32F, mapx, mapy)
// OpenCV is not part of the reference tree and not installed here, so this is a restatement of the
// published algorithm (Bouguet's rectification as implemented by cvStereoRectify in OpenCV 2.4,
// 5-coefficient Brown distortion).  OpenCV's own bits stay UNPINNED; what pins this
// file is an independent second implementation with other formulas throughout, kept with the test infrastructure
// (tests/test_rectify.py names it: quaternion half rotation, vector-to-vector alignment, Newton or five-sweep undistortion), which it agrees
// with to 1e-12 (rotations) / 2e-6 (P, Q) on nine rigs (tests/test_rectify.py), plus geometric properties.
#include "../../include/jn_stereo.h"
#include <cfloat>
#include <cmath>
#include <cstring>
#include <algorithm>

namespace {

struct M3 { double a[3][3]; };
struct V3 { double v[3]; };

M3 mul(const M3& x, const M3& y) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += x.a[i][k] * y.a[k][j]; r.a[i][j] = s; } return r; }
M3 transpose(const M3& x) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.a[i][j] = x.a[j][i]; return r; }
V3 mul(const M3& x, const V3& y) { V3 r; for (int i = 0; i < 3; i++) r.v[i] = x.a[i][0] * y.v[0] + x.a[i][1] * y.v[1] + x.a[i][2] * y.v[2]; return r; }
double norm(const V3& x) { return std::sqrt(x.v[0] * x.v[0] + x.v[1] * x.v[1] + x.v[2] * x.v[2]); }

// rotation vector -> matrix (Rodrigues formula)
M3 rodrigues(const V3& om) {
  const double th = norm(om);
  M3 R = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
  if (th < DBL_EPSILON) return R;
  const double c = std::cos(th), s = std::sin(th), c1 = 1 - c, itheta = 1.0 / th;
  const double rx = om.v[0] * itheta, ry = om.v[1] * itheta, rz = om.v[2] * itheta;
  const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
  const double rxm[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < 9; k++) R.a[k / 3][k % 3] = c * I[k] + c1 * rrt[k] + s * rxm[k];
  return R;
}
// rotation matrix -> vector
V3 rodrigues_inv(const M3& R) {
  V3 r = {{R.a[2][1] - R.a[1][2], R.a[0][2] - R.a[2][0], R.a[1][0] - R.a[0][1]}};
  const double s = std::sqrt((r.v[0] * r.v[0] + r.v[1] * r.v[1] + r.v[2] * r.v[2]) * 0.25);
  double c = (R.a[0][0] + R.a[1][1] + R.a[2][2] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  const double theta = std::acos(c);
  if (s < 1e-5) {
    if (c > 0) return V3{{0, 0, 0}};
    double t;
    t = (R.a[0][0] + 1) * 0.5; r.v[0] = std::sqrt(std::max(t, 0.));
    t = (R.a[1][1] + 1) * 0.5; r.v[1] = std::sqrt(std::max(t, 0.)) * (R.a[0][1] < 0 ? -1. : 1.);
    t = (R.a[2][2] + 1) * 0.5; r.v[2] = std::sqrt(std::max(t, 0.)) * (R.a[0][2] < 0 ? -1. : 1.);
    if (std::fabs(r.v[0]) < std::fabs(r.v[1]) && std::fabs(r.v[0]) < std::fabs(r.v[2]) && (R.a[1][2] > 0) != (r.v[1] * r.v[2] > 0)) r.v[2] = -r.v[2];
    const double k = theta / norm(r);
    for (double& x : r.v) x *= k;
    return r;
  }
  const double vth = 1 / (2 * s) * theta;
  for (double& x : r.v) x *= vth;
  return r;
}

// iterative inverse of the Brown model + optional rectification R and new projection P (cvUndistortPoints)
void undistort_point(const double K[9], const double D[5], const M3* RR, double u, double v, double& xo, double& yo) {
  const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  double x = (u - cx) / fx, y = (v - cy) / fy;
  const double x0 = x, y0 = y;
  for (int it = 0; it < 5; it++) {
    const double r2 = x * x + y * y;
    const double icdist = 1. / (1 + ((D[4] * r2 + D[1]) * r2 + D[0]) * r2);
    const double dx = 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x);
    const double dy = D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y;
    x = (x0 - dx) * icdist; y = (y0 - dy) * icdist;
  }
  if (RR) {
    const double xx = RR->a[0][0] * x + RR->a[0][1] * y + RR->a[0][2], yy = RR->a[1][0] * x + RR->a[1][1] * y + RR->a[1][2];
    const double ww = 1. / (RR->a[2][0] * x + RR->a[2][1] * y + RR->a[2][2]);
    x = xx * ww; y = yy * ww;
  }
  xo = x; yo = y;
}

struct RectF { float x, y, w, h; };
// inscribed / circumscribed rectangles of the undistorted 9x9 sample grid (icvGetRectangles)
void rectangles(const double K[9], const double D[5], const M3& R, const double P[12], int W, int H, RectF& inner, RectF& outer) {
  const int N = 9;
  M3 Pm = {{{P[0], P[1], P[2]}, {P[4], P[5], P[6]}, {P[8], P[9], P[10]}}};
  const M3 RR = mul(Pm, R);
  float iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX, oX0 = FLT_MAX, oX1 = -FLT_MAX, oY0 = FLT_MAX, oY1 = -FLT_MAX;
  for (int y = 0; y < N; y++)
    for (int x = 0; x < N; x++) {
      double px, py;
      undistort_point(K, D, &RR, (double)((float)x * W / (N - 1)), (double)((float)y * H / (N - 1)), px, py);
      const float fx = (float)px, fy = (float)py;
      oX0 = std::min(oX0, fx); oX1 = std::max(oX1, fx); oY0 = std::min(oY0, fy); oY1 = std::max(oY1, fy);
      if (x == 0) iX0 = std::max(iX0, fx);
      if (x == N - 1) iX1 = std::min(iX1, fx);
      if (y == 0) iY0 = std::max(iY0, fy);
      if (y == N - 1) iY1 = std::min(iY1, fy);
    }
  inner = RectF{iX0, iY0, iX1 - iX0, iY1 - iY0};
  outer = RectF{oX0, oY0, oX1 - oX0, oY1 - oY0};
}

}  // namespace

extern "C" {

// calibration/amrl_jackal_webcam_stereo.yml:1-37, calibrated at 640x360 (point_cloud.cpp:38)
void jn_stereo_calib_default(jn_stereo_calib* c) {
  const double K1[9] = {4.6417933392659904e+02, 0., 3.2479711799310849e+02, 0., 4.6611716361740059e+02, 1.8685472713963392e+02, 0., 0., 1.};
  const double K2[9] = {4.6394860327263103e+02, 0., 3.3106360678338558e+02, 0., 4.6375869272018139e+02, 1.7571346440013161e+02, 0., 0., 1.};
  const double D1[5] = {1.4885193925432560e-01, -3.8454604770748702e-01, -1.8950854861753609e-03, 7.8121300147955593e-03, 2.7294034259258465e-01};
  const double D2[5] = {1.2249914632632175e-01, -2.1440080513600884e-01, -2.8013224434709164e-03, 4.6375383671683921e-03, 4.3812259920217027e-02};
  const double R[9] = {9.9942653697036332e-01, -2.9629020698892981e-02, -1.6392630412826015e-02, 2.9331318104542686e-02, 9.9940562799161392e-01,
                       -1.8112551364675371e-02, 1.6919544251458529e-02, 1.7621347028886541e-02, 9.9970156404359523e-01};
  const double T[3] = {-9.4052586442980660e-02, -1.2149101400467301e-03, -7.2235718228952177e-04};
  memcpy(c->K1, K1, sizeof K1); memcpy(c->K2, K2, sizeof K2); memcpy(c->D1, D1, sizeof D1); memcpy(c->D2, D2, sizeof D2);
  memcpy(c->R, R, sizeof R); memcpy(c->T, T, sizeof T);
  c->calib_width = 640; c->calib_height = 360;
}

// stereoRectify(..., CV_CALIB_ZERO_DISPARITY, alpha = 0, newImageSize) — point_cloud.cpp:543-544
jn_status jn_stereo_rectify(const jn_stereo_calib* c, int32_t new_width, int32_t new_height, jn_rectification* out) {
  if (!c || !out || c->calib_width < 1 || c->calib_height < 1) return JN_ERR_INVALID;
  const int nx = c->calib_width, ny = c->calib_height;
  M3 R; memcpy(R.a, c->R, sizeof R.a);
  const V3 T = {{c->T[0], c->T[1], c->T[2]}};
  V3 om = rodrigues_inv(R);
  for (double& x : om.v) x *= -0.5;                         // each camera turns half way
  const M3 r_r = rodrigues(om);
  V3 t = mul(r_r, T);
  const int idx = std::fabs(t.v[0]) > std::fabs(t.v[1]) ? 0 : 1;     // horizontal or vertical rig
  const double cc = t.v[idx], nt = norm(t);
  V3 uu = {{0, 0, 0}}; uu.v[idx] = cc > 0 ? 1 : -1;
  V3 ww = {{t.v[1] * uu.v[2] - t.v[2] * uu.v[1], t.v[2] * uu.v[0] - t.v[0] * uu.v[2], t.v[0] * uu.v[1] - t.v[1] * uu.v[0]}};
  const double nw = norm(ww);
  if (nw > 0.0) { const double k = std::acos(std::fabs(cc) / nt) / nw; for (double& x : ww.v) x *= k; }
  const M3 wR = rodrigues(ww);                              // turn the baseline onto the image x (or y) axis
  const M3 R1 = mul(wR, transpose(r_r)), R2 = mul(wR, r_r);
  t = mul(R2, T);

  const double* Ks[2] = {c->K1, c->K2};
  const double* Ds[2] = {c->D1, c->D2};
  const M3* Rs[2] = {&R1, &R2};
  double fc_new = DBL_MAX;
  for (int k = 0; k < 2; k++) {
    double fc = Ks[k][4 * (idx ^ 1)];
    if (Ds[k][0] < 0) fc *= 1 + Ds[k][0] * (nx * nx + ny * ny) / (4 * fc * fc);
    fc_new = std::min(fc_new, fc);
  }
  double ccx[2], ccy[2];
  for (int k = 0; k < 2; k++) {
    double ax = 0, ay = 0;
    for (int i = 0; i < 4; i++) {                           // image corners through undistortion, rotation and the new focal length
      const float px = (float)((i % 2) * nx), py = (float)((i < 2 ? 0 : 1) * ny);
      double x, y;
      undistort_point(Ks[k], Ds[k], nullptr, px, py, x, y);
      const float xf = (float)x, yf = (float)y;             // points are stored as float32 between the two calls
      const V3 p = mul(*Rs[k], V3{{xf, yf, 1.0}});
      ax += (float)(fc_new * p.v[0] / p.v[2]); ay += (float)(fc_new * p.v[1] / p.v[2]);
    }
    ccx[k] = nx / 2 - ax / 4; ccy[k] = ny / 2 - ay / 4;
  }
  ccx[0] = ccx[1] = (ccx[0] + ccx[1]) * 0.5;                // CV_CALIB_ZERO_DISPARITY
  ccy[0] = ccy[1] = (ccy[0] + ccy[1]) * 0.5;

  double P1[12] = {fc_new, 0, ccx[0], 0, 0, fc_new, ccy[0], 0, 0, 0, 1, 0};
  double P2[12] = {fc_new, 0, ccx[1], 0, 0, fc_new, ccy[1], 0, 0, 0, 1, 0};
  P2[4 * idx + 3] = t.v[idx] * fc_new;                      // baseline * focal length

  RectF in1, out1, in2, out2;
  rectangles(c->K1, c->D1, R1, P1, nx, ny, in1, out1);
  rectangles(c->K2, c->D2, R2, P2, nx, ny, in2, out2);
  const int nw_ = new_width * new_height != 0 ? new_width : nx, nh_ = new_width * new_height != 0 ? new_height : ny;
  const double cx1_0 = ccx[0], cy1_0 = ccy[0], cx2_0 = ccx[1], cy2_0 = ccy[1];
  const double cx1 = nw_ * cx1_0 / nx, cy1 = nh_ * cy1_0 / ny, cx2 = nw_ * cx2_0 / nx, cy2 = nh_ * cy2_0 / ny;
  // alpha = 0: scale so that only valid pixels remain
  double s0 = std::max(std::max(std::max(cx1 / (cx1_0 - in1.x), cy1 / (cy1_0 - in1.y)), (nw_ - cx1) / (in1.x + in1.w - cx1_0)),
                       (nh_ - cy1) / (in1.y + in1.h - cy1_0));
  s0 = std::max(std::max(std::max(std::max(cx2 / (cx2_0 - in2.x), cy2 / (cy2_0 - in2.y)), (nw_ - cx2) / (in2.x + in2.w - cx2_0)),
                         (nh_ - cy2) / (in2.y + in2.h - cy2_0)), s0);
  const double s = s0;
  fc_new *= s;
  P1[0] = P1[5] = fc_new; P1[2] = cx1; P1[6] = cy1;
  P2[0] = P2[5] = fc_new; P2[2] = cx2; P2[6] = cy2;
  P2[4 * idx + 3] *= s;

  memcpy(out->R1, R1.a, sizeof out->R1); memcpy(out->R2, R2.a, sizeof out->R2);
  memcpy(out->P1, P1, sizeof P1); memcpy(out->P2, P2, sizeof P2);
  const double q[16] = {1, 0, 0, -cx1, 0, 1, 0, -cy1, 0, 0, 0, fc_new, 0, 0, -1. / t.v[idx], (idx == 0 ? cx1 - cx2 : cy1 - cy2) / t.v[idx]};
  memcpy(out->Q, q, sizeof q);
  return JN_OK;
}

}  // extern "C"
