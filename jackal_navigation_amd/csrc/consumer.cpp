// consumer.cpp — how the published obstacle scan is consumed (product code, host side).
//
// navigate.cpp turns /webcam/left/obstacle_scan into a stop / turn decision
// (laserScanCallback :344-363, checkObstacle :101-153, chooseDirection :155-197, the decision part of
// obstacleAvoidMode :229-256).  The arithmetic is a few hundred flops per scan, so it stays on the host;
// it exists so that a scan produced by the HIP path can be shown to drive the same decisions as the
// reference's.  State lives in a caller-owned jn_nav_state (the reference keeps file-scope globals).
#include <cmath>
#include <cstring>
#include "../../include/jn_stereo.h"

extern "C" {

void jn_nav_params_default(jn_nav_params* p) {
  if (!p) return;
  p->clear_front = 0.24 + 0.8;      // navigate.cpp:37
  p->clear_side = 0.3;              // :38
  p->stop_dist = 0.5;               // :125
  p->laser_pt_thresh = 8;           // :42
  p->history = 20;                  // :129
  p->history_votes = 2;             // :146
}

void jn_nav_state_reset(jn_nav_state* s) {
  if (s) std::memset(s, 0, sizeof(*s));
}

int32_t jn_scan_to_points(const float* ranges, int32_t n, float angle_min, float angle_max, double* xy) {
  if (!ranges || !xy || n < 0) return -1;
  const double lo = angle_min, span = (double)angle_max - (double)angle_min, count = (double)n;
  for (int32_t i = 0; i < n; i++) {
    const double a = (double)i * span / count + lo;                    // :357
    const double r = ranges[i];
    xy[2 * i] = r * std::cos(a);
    xy[2 * i + 1] = r * std::sin(a);
  }
  return n;
}

jn_status jn_nav_vote(const jn_nav_params* p, jn_nav_state* s, const double* xy, int32_t n, jn_nav_decision* out) {
  if (!p || !s || !out || (n > 0 && !xy) || n < 0 || p->history < 1 || p->history > JN_NAV_MAX_HISTORY) return JN_ERR_INVALID;
  int32_t inside = 0, left = 0, right = 0;
  double closest = 1e9;                                                // INF, :47
  for (int32_t i = 0; i < n; i++) {
    const double x = xy[2 * i], y = xy[2 * i + 1];
    const double dist = std::sqrt(x * x + y * y);
    if (dist < closest) closest = dist;
    if (x > 0. && x < p->clear_front) {
      if (y > -p->clear_side && y < p->clear_side) inside++;           // :108-111
      if (y < 0) right++; else left++;                                 // :158-166
    }
  }
  const int32_t now = (inside > p->laser_pt_thresh || closest < p->stop_dist) ? 1 : 0;   // :114-126
  // ring of the last `history` votes with a running count of positives (the reference recounts a deque)
  if (s->filled < p->history) s->filled++;
  else s->positives -= s->votes[s->head];
  s->votes[s->head] = now;
  s->positives += now;
  s->head = (s->head + 1) % p->history;
  const int32_t obstacle = (now || s->positives > p->history_votes) ? 1 : 0;             // :146-147
  out->points_inside = inside; out->points = n; out->closest = closest;
  out->confidence = (double)s->positives / (double)s->filled;                             // :149
  out->obstacle = obstacle;
  int32_t dir = 0;
  if (obstacle) {                                                                          // :233-235
    const int32_t ahead = left + right;
    if (ahead >= 2) {                                                                      // :168-169
      const double turn_left = 2. * (double)right / (double)ahead, turn_right = 2. * (double)left / (double)ahead;
      if (turn_left > turn_right) dir = (s->last_dir == 1 || turn_left - turn_right > 0.5) ? 1 : s->last_dir;
      else dir = (s->last_dir == 2 || turn_right - turn_left > 0.5) ? 2 : s->last_dir;
    }
  }
  s->last_dir = dir;                                                                       // :235, :252
  out->direction = dir;
  return JN_OK;
}

}  // extern "C"
