// delaunay.cpp — see delaunay.h.  Host stage of the stereo path (product code).
//
// Decision sequence reproduced (reference line numbers in src/elas/triangle.cpp):
//   sort by (x,y) with LCG-pivot quicksort      :4045-4049, :5446-5500 (replayed only when vertices coincide)
//   duplicates: first in sorted order survives   :6179-6194
//   alternating-cut re-partition                 :5514-5606, :6197-6206 (same result, computed kd-style)
//   2-/3-vertex bases, recursive hull zipping     :5638-5947, :5953-6103
//   output = non-ghost triangles, creation order, (org,dest,apex) of edge 0   :6105-6148, :7832-7843
// Predicates are evaluated exactly in int64 (coordinates are pixel integers), which equals the
// sign the reference's adaptive float predicates return (:2706-2745, :3334-3379).
#include "delaunay.h"
#include "hooks.h"
#include <algorithm>
#include <cstdlib>

namespace jnav {

inline int Delaunay::orient(int a, int b, int c) const {
  const int64_t acx = x_[a] - x_[c], acy = y_[a] - y_[c], bcx = x_[b] - x_[c], bcy = y_[b] - y_[c];
  const int64_t det = acx * bcy - acy * bcx;
  return det > 0 ? 1 : (det < 0 ? -1 : 0);
}

inline int Delaunay::in_circle(int a, int b, int c, int d) const {
  const int64_t ax = x_[a] - x_[d], ay = y_[a] - y_[d];
  const int64_t bx = x_[b] - x_[d], by = y_[b] - y_[d];
  const int64_t cx = x_[c] - x_[d], cy = y_[c] - y_[d];
  const int64_t det = (ax * ax + ay * ay) * (bx * cy - by * cx) + (bx * bx + by * by) * (cx * ay - cy * ax) +
                      (cx * cx + cy * cy) * (ax * by - ay * bx);
  return det > 0 ? 1 : (det < 0 ? -1 : 0);
}

inline bool Delaunay::precedes(int a, int b, int axis) const {
  const int32_t pa = axis ? y_[a] : x_[a], pb = axis ? y_[b] : x_[b];
  if (pa != pb) return pa < pb;
  return (axis ? x_[a] : y_[a]) < (axis ? x_[b] : y_[b]);
}

unsigned Delaunay::draw(unsigned choices) {
  lcg_ = (lcg_ * 1366u + 150889u) % 714025u;
  return (unsigned)(lcg_ / (714025u / choices + 1u));
}

Delaunay::H Delaunay::fresh(Ctx& c) {
  const int t = c.next++;
  link_[4 * t] = link_[4 * t + 1] = link_[4 * t + 2] = -1;
  vert_[4 * t] = vert_[4 * t + 1] = vert_[4 * t + 2] = -1;
  return (H)t << 2;
}

void Delaunay::partition(int32_t* a, int n, int axis, int& l, int& r) {
  const int pv = a[draw((unsigned)n)];
  l = -1; r = n;
  while (l < r) {
    do ++l; while (l <= r && precedes(a[l], pv, axis));
    do --r; while (l <= r && precedes(pv, a[r], axis));
    if (l < r) { const int32_t t = a[l]; a[l] = a[r]; a[r] = t; }
  }
}

void Delaunay::quicksort(int32_t* a, int n) {
  if (n == 2) {
    if (precedes(a[1], a[0], 0)) { const int32_t t = a[0]; a[0] = a[1]; a[1] = t; }
    return;
  }
  int l, r;
  partition(a, n, 0, l, r);
  if (l > 1) quicksort(a, l);
  if (r < n - 2) quicksort(a + r + 1, n - r - 1);
}

// (x,y) order without replaying the quicksort.  With all keys distinct the sorted order is unique, so any
// sort gives what the reference's does: nothing at all when the input is already ascending (the left
// image's support list is, elas.cpp:425-431 walks u outer / v inner), else two stable counting passes (y,
// then x).  Returns false when two vertices coincide; the caller then replays the reference's sort.
bool Delaunay::sort_distinct(int32_t* a, int n) {
  bool ascending = true;
  int32_t xmin = x_[0], xmax = xmin, ymin = y_[0], ymax = ymin;
  for (int i = 1; i < n; i++) {
    const int32_t xi = x_[i], yi = y_[i];
    ascending &= x_[i - 1] < xi || (x_[i - 1] == xi && y_[i - 1] < yi);
    xmin = xi < xmin ? xi : xmin; xmax = xi > xmax ? xi : xmax;
    ymin = yi < ymin ? yi : ymin; ymax = yi > ymax ? yi : ymax;
  }
  if (ascending) {
    for (int i = 0; i < n; i++) a[i] = i;
    return true;
  }
  const int xr = xmax - xmin + 1, yr = ymax - ymin + 1;
  if ((int64_t)xr + yr > 8 * (int64_t)n + 4096) return false;        // sparse keys: buckets would cost more than the sort
  const size_t need = (size_t)(xr > yr ? xr : yr) + 1;
  if (bucket_.size() < need) bucket_.resize(need);
  int32_t* b = bucket_.data();
  int32_t* t = tmp_.data();
  std::fill(b, b + yr + 1, 0);
  for (int i = 0; i < n; i++) b[y_[i] - ymin + 1]++;
  for (int i = 0; i < yr; i++) b[i + 1] += b[i];
  for (int i = 0; i < n; i++) t[b[y_[i] - ymin]++] = i;
  std::fill(b, b + xr + 1, 0);
  for (int i = 0; i < n; i++) b[x_[i] - xmin + 1]++;
  for (int i = 0; i < xr; i++) b[i + 1] += b[i];
  for (int i = 0; i < n; i++) a[b[x_[t[i]] - xmin]++] = t[i];
  for (int i = 1; i < n; i++)
    if (x_[a[i - 1]] == x_[a[i]] && y_[a[i - 1]] == y_[a[i]]) return false;
  return true;
}

// Alternating-cut arrangement (the effect of triangle.cpp:5514-5606, :6197-6206).  Triangle reaches it
// with randomised quick-select; because duplicates are gone the keys are distinct, every median
// split is a unique set partition and the leaves (<= 3 vertices) are x-sorted, so the final array is
// unique.  We build the same array deterministically, kd-tree style: keep the vertices once in
// (x,y) order (array a) and once in (y,x) order; a cut along one order is a prefix, the other
// order is stably partitioned to follow.  The x-ordered array, partitioned in place, IS the result.
// One cut of [lo, hi): the first half of the defining order goes left, the other order follows stably.
void Delaunay::cut(int lo, int hi, int axis) {
  const int n = hi - lo;
  const int half = n >> 1;
  int32_t* def = (axis ? by_y_.data() : order_.data()) + lo;     // order that defines the cut
  int32_t* oth = (axis ? order_.data() : by_y_.data()) + lo;     // order that must follow it
  for (int i = 0; i < half; i++) left_[def[i]] = 1;
  for (int i = half; i < n; i++) left_[def[i]] = 0;
  int32_t* spill = tmp_.data() + lo;                             // ranges of concurrent parts are disjoint
  int nl = 0, nr = 0;
  for (int i = 0; i < n; i++) {
    const int32_t v = oth[i];
    if (left_[v]) oth[nl++] = v; else spill[nr++] = v;
  }
  for (int i = 0; i < nr; i++) oth[nl + i] = spill[i];
}
void Delaunay::split(int lo, int hi, int axis) {
  const int n = hi - lo;
  if (n <= 3) return;
  cut(lo, hi, axis);
  const int half = n >> 1;
  split(lo, lo + half, 1 - axis);
  split(lo + half, hi, 1 - axis);
}

void Delaunay::arrange(int32_t* a, int n) {
  // a == order_.data(), sorted by (x,y).  Stable counting sort by y gives the (y,x) order.
  int32_t ymin = y_[a[0]], ymax = ymin;
  for (int i = 1; i < n; i++) { const int32_t y = y_[a[i]]; ymin = y < ymin ? y : ymin; ymax = y > ymax ? y : ymax; }
  const int range = ymax - ymin + 1;
  if (bucket_.size() < (size_t)range + 1) bucket_.resize(range + 1);
  std::fill(bucket_.begin(), bucket_.begin() + range + 1, 0);
  for (int i = 0; i < n; i++) bucket_[y_[a[i]] - ymin + 1]++;
  for (int i = 0; i < range; i++) bucket_[i + 1] += bucket_[i];
  for (int i = 0; i < n; i++) by_y_[bucket_[y_[a[i]] - ymin]++] = a[i];
}

// Merge two triangulated halves by walking up the seam between their hulls.
void Delaunay::zip(H& farleft, H& innerleft, H& innerright, H& farright, int axis, Ctx& c) {
  int il_dest = v_dest(innerleft), il_apex = v_apex(innerleft);
  int ir_org = v_org(innerright), ir_apex = v_apex(innerright);

  if (axis == 1) {   // horizontal cut: hull handles must point at the extreme-y vertices
    int fl_pt = v_org(farleft), fl_apex = v_apex(farleft);
    int fr_pt = v_dest(farright);
    while (y_[fl_apex] < y_[fl_pt]) {
      farleft = across(ccw_edge(farleft));
      fl_pt = fl_apex; fl_apex = v_apex(farleft);
    }
    H probe = across(innerleft); int pv = v_apex(probe);
    while (y_[pv] > y_[il_dest]) {
      innerleft = ccw_edge(probe);
      il_apex = il_dest; il_dest = pv;
      probe = across(innerleft); pv = v_apex(probe);
    }
    while (y_[ir_apex] < y_[ir_org]) {
      innerright = across(ccw_edge(innerright));
      ir_org = ir_apex; ir_apex = v_apex(innerright);
    }
    probe = across(farright); pv = v_apex(probe);
    while (y_[pv] > y_[fr_pt]) {
      farright = ccw_edge(probe);
      fr_pt = pv;
      probe = across(farright); pv = v_apex(probe);
    }
  }

  for (bool again = true; again;) {   // slide down to the lower common tangent
    again = false;
    if (orient(il_dest, il_apex, ir_org) > 0) {
      innerleft = across(cw_edge(innerleft));
      il_dest = il_apex; il_apex = v_apex(innerleft); again = true;
    }
    if (orient(ir_apex, ir_org, il_dest) > 0) {
      innerright = across(ccw_edge(innerright));
      ir_org = ir_apex; ir_apex = v_apex(innerright); again = true;
    }
  }

  H lcand = across(innerleft), rcand = across(innerright);
  H base = fresh(c);
  glue(base, innerleft);  base = ccw_edge(base);
  glue(base, innerright); base = ccw_edge(base);
  v_org(base) = ir_org; v_dest(base) = il_dest;
  if (il_dest == v_org(farleft)) farleft = ccw_edge(base);
  if (ir_org == v_dest(farright)) farright = cw_edge(base);

  int lo_l = il_dest, lo_r = ir_org;
  int up_l = v_apex(lcand), up_r = v_apex(rcand);

  for (;;) {
    const bool l_done = orient(up_l, lo_l, lo_r) <= 0;
    const bool r_done = orient(up_r, lo_l, lo_r) <= 0;
    if (l_done && r_done) {
      H cap = fresh(c);
      v_org(cap) = lo_l; v_dest(cap) = lo_r;
      glue(cap, base);  cap = ccw_edge(cap);
      glue(cap, rcand); cap = ccw_edge(cap);
      glue(cap, lcand);
      if (axis == 1) {   // back to extreme-x handles
        int fl_pt = v_org(farleft);
        int fr_pt = v_dest(farright), fr_apex = v_apex(farright);
        H probe = across(farleft); int pv = v_apex(probe);
        while (x_[pv] < x_[fl_pt]) {
          farleft = cw_edge(probe);
          fl_pt = pv;
          probe = across(farleft); pv = v_apex(probe);
        }
        while (x_[fr_apex] > x_[fr_pt]) {
          farright = across(cw_edge(farright));
          fr_pt = fr_apex; fr_apex = v_apex(farright);
        }
      }
      return;
    }
    if (!l_done) {   // flip away left-hull edges that the new cross edge invalidates
      H e = across(cw_edge(lcand));
      int w = v_apex(e);
      if (w >= 0) {
        bool bad = in_circle(lo_l, lo_r, up_l, w) > 0;
        while (bad) {
          e = ccw_edge(e); const H top = across(e);
          e = ccw_edge(e); const H side = across(e);
          glue(e, top);
          glue(lcand, side);
          lcand = ccw_edge(lcand); const H outer = across(lcand);
          e = cw_edge(e);
          glue(e, outer);
          v_org(lcand) = lo_l; v_dest(lcand) = -1; v_apex(lcand) = w;
          v_org(e) = -1; v_dest(e) = up_l; v_apex(e) = w;
          up_l = w;
          e = side; w = v_apex(e);
          bad = w >= 0 && in_circle(lo_l, lo_r, up_l, w) > 0;
        }
      }
    }
    if (!r_done) {   // same on the right hull, mirrored
      H e = across(ccw_edge(rcand));
      int w = v_apex(e);
      if (w >= 0) {
        bool bad = in_circle(lo_l, lo_r, up_r, w) > 0;
        while (bad) {
          e = cw_edge(e); const H top = across(e);
          e = cw_edge(e); const H side = across(e);
          glue(e, top);
          glue(rcand, side);
          rcand = cw_edge(rcand); const H outer = across(rcand);
          e = ccw_edge(e);
          glue(e, outer);
          v_org(rcand) = -1; v_dest(rcand) = lo_r; v_apex(rcand) = w;
          v_org(e) = up_r; v_dest(e) = -1; v_apex(e) = w;
          up_r = w;
          e = side; w = v_apex(e);
          bad = w >= 0 && in_circle(lo_l, lo_r, up_r, w) > 0;
        }
      }
    }
    if (l_done || (!r_done && in_circle(up_l, lo_l, lo_r, up_r) > 0)) {
      glue(base, rcand);
      base = cw_edge(rcand);
      v_dest(base) = lo_l;
      lo_r = up_r;
      rcand = across(base);
      up_r = v_apex(rcand);
    } else {
      glue(base, lcand);
      base = ccw_edge(lcand);
      v_org(base) = lo_r;
      lo_l = up_l;
      lcand = across(base);
      up_l = v_apex(lcand);
    }
  }
}

void Delaunay::conquer(int32_t* a, int n, int axis, H& farleft, H& farright, Ctx& c) {
  if (n == 2) {   // a lone edge: two ghosts glued on all three sides
    farleft = fresh(c);  v_org(farleft) = a[0];  v_dest(farleft) = a[1];
    farright = fresh(c); v_org(farright) = a[1]; v_dest(farright) = a[0];
    glue(farleft, farright);
    farleft = cw_edge(farleft); farright = ccw_edge(farright); glue(farleft, farright);
    farleft = cw_edge(farleft); farright = ccw_edge(farright); glue(farleft, farright);
    farleft = cw_edge(farright);
    return;
  }
  if (n == 3) {
    H mid = fresh(c), g1 = fresh(c), g2 = fresh(c), g3 = fresh(c);
    const int turn = orient(a[0], a[1], a[2]);
    if (turn == 0) {   // collinear triple: two edges, four ghosts
      v_org(mid) = a[0]; v_dest(mid) = a[1];
      v_org(g1) = a[1];  v_dest(g1) = a[0];
      v_org(g2) = a[2];  v_dest(g2) = a[1];
      v_org(g3) = a[1];  v_dest(g3) = a[2];
      glue(mid, g1); glue(g2, g3);
      mid = ccw_edge(mid); g1 = cw_edge(g1); g2 = ccw_edge(g2); g3 = cw_edge(g3);
      glue(mid, g3); glue(g1, g2);
      mid = ccw_edge(mid); g1 = cw_edge(g1); g2 = ccw_edge(g2); g3 = cw_edge(g3);
      glue(mid, g1); glue(g2, g3);
      farleft = g1; farright = g2;
    } else {           // one real triangle ringed by three ghosts
      const int second = turn > 0 ? a[1] : a[2], third = turn > 0 ? a[2] : a[1];
      v_org(mid) = a[0];   v_dest(g1) = a[0];   v_org(g3) = a[0];
      v_dest(mid) = second; v_org(g1) = second; v_dest(g2) = second;
      v_apex(mid) = third;  v_org(g2) = third;  v_dest(g3) = third;
      glue(mid, g1); mid = ccw_edge(mid);
      glue(mid, g2); mid = ccw_edge(mid);
      glue(mid, g3);
      g1 = cw_edge(g1); g2 = ccw_edge(g2); glue(g1, g2);
      g1 = cw_edge(g1); g3 = cw_edge(g3);  glue(g1, g3);
      g2 = ccw_edge(g2); g3 = cw_edge(g3); glue(g2, g3);
      farleft = g1;
      farright = turn > 0 ? g2 : ccw_edge(farleft);
    }
    return;
  }
  const int half = n >> 1;
  H il, ir;
  conquer(a, half, 1 - axis, farleft, il, c);
  conquer(a + half, n - half, 1 - axis, ir, farright, c);
  zip(farleft, il, ir, farright, axis, c);
}

int Delaunay::prepare(const int32_t* x, const int32_t* y, int n, int want_parts) {
  nparts_ = 0; k_ = 0;
  if (n < 3) return 0;
  x_ = x; y_ = y; lcg_ = 1;
  const size_t cap = (size_t)8 * n + 64;   // real + ghost triangles ever created (< 4n) + the slack of the parts' slot ranges
  if (link_.size() < 4 * cap) { link_.resize(4 * cap); vert_.resize(4 * cap); }
  if (order_.size() < (size_t)n) { order_.resize(n); by_y_.resize(n); tmp_.resize(n); left_.resize(n); }
  int32_t* a = order_.data();
  int k = n;
  if (!sort_distinct(a, n)) {
    // duplicate vertices: which of them survives depends on the reference's quicksort, pivot draws included
    for (int i = 0; i < n; i++) a[i] = i;
    quicksort(a, n);
    k = 0;
    for (int j = 1; j < n; j++)
      if (x[a[k]] != x[a[j]] || y[a[k]] != y[a[j]]) a[++k] = a[j];
    ++k;
  }
  if (k < 2) return 0;
  k_ = k;
  arrange(a, k);
  // the top of the recursion conquer(a, k, 0): cut on axis 0, children are conquered on axis 1, grandchildren on axis 0
  // cutting pays once a part is worth more than the two extra pool rounds it costs (measured on the GPU box: 819 points
  // of a 640x480 pair 0.09 -> 0.135 ms when cut in four, 3232 points of a 1280x720 pair 0.63 -> 0.27 ms); the test hook
  // JN_DELAUNAY_MIN_POINTS lowers the bar so that small inputs exercise the cuts too
  static const int min_pts = JN_HOOK_ENV("JN_DELAUNAY_MIN_POINTS") ? atoi(JN_HOOK_ENV("JN_DELAUNAY_MIN_POINTS")) : 1024;
  int parts = (want_parts >= 4 && k >= 2 * min_pts) ? 4 : ((want_parts >= 2 && k >= min_pts) ? 2 : 1);
  if (parts == 1) part_[0] = Part{0, k, 0, 0, 0, 0, 0};
  else {
    cut(0, k, 0);
    const int half = k >> 1;
    if (parts == 2) { part_[0] = Part{0, half, 1, 0, 0, 0, 0}; part_[1] = Part{half, k, 1, 0, 0, 0, 0}; }
    else {
      cut(0, half, 1); cut(half, k, 1);
      const int q0 = half >> 1, q2 = (k - half) >> 1;
      part_[0] = Part{0, q0, 0, 0, 0, 0, 0}; part_[1] = Part{q0, half, 0, 0, 0, 0, 0};
      part_[2] = Part{half, half + q2, 0, 0, 0, 0, 0}; part_[3] = Part{half + q2, k, 0, 0, 0, 0, 0};
    }
  }
  // slot ranges in the order the sequential recursion creates triangles: p0 p1 [merge 01] p2 p3 [merge 23] [merge top]
  int slot = 0;
  auto reserve_part = [&](int i) { part_[i].slot0 = slot; slot += 4 * (part_[i].hi - part_[i].lo) + 8; };
  if (parts == 1) reserve_part(0);
  else if (parts == 2) { reserve_part(0); reserve_part(1); zslot_[0] = slot; slot += 2; }
  else { reserve_part(0); reserve_part(1); zslot_[0] = slot; slot += 2; reserve_part(2); reserve_part(3); zslot_[1] = slot; slot += 2; zslot_[2] = slot; slot += 2; }
  nparts_ = parts;
  return parts;
}

void Delaunay::subtree(int i) {
  Part& p = part_[i];
  split(p.lo, p.hi, p.axis);
  Ctx c{p.slot0};
  conquer(order_.data() + p.lo, p.hi - p.lo, p.axis, p.farleft, p.farright, c);
  p.used = c.next - p.slot0;
}

int Delaunay::finish(int32_t* tri) {
  if (nparts_ == 0) return -1;
  int out = 0;
  auto emit = [&](int slot0, int count) {
    for (int t = slot0; t < slot0 + count; t++) {
      const int32_t* c = &vert_[4 * t];
      if ((c[0] | c[1] | c[2]) < 0) continue;       // ghost
      tri[3 * out] = c[1]; tri[3 * out + 1] = c[2]; tri[3 * out + 2] = c[0];
      out++;
    }
  };
  if (nparts_ == 1) { emit(part_[0].slot0, part_[0].used); return out; }
  if (nparts_ == 2) {
    Ctx c{zslot_[0]};
    zip(part_[0].farleft, part_[0].farright, part_[1].farleft, part_[1].farright, 0, c);
    emit(part_[0].slot0, part_[0].used); emit(part_[1].slot0, part_[1].used); emit(zslot_[0], 2);
    return out;
  }
  Ctx c0{zslot_[0]}, c1{zslot_[1]}, c2{zslot_[2]};
  zip(part_[0].farleft, part_[0].farright, part_[1].farleft, part_[1].farright, 1, c0);
  zip(part_[2].farleft, part_[2].farright, part_[3].farleft, part_[3].farright, 1, c1);
  zip(part_[0].farleft, part_[1].farright, part_[2].farleft, part_[3].farright, 0, c2);
  emit(part_[0].slot0, part_[0].used); emit(part_[1].slot0, part_[1].used); emit(zslot_[0], 2);
  emit(part_[2].slot0, part_[2].used); emit(part_[3].slot0, part_[3].used); emit(zslot_[1], 2); emit(zslot_[2], 2);
  return out;
}

int Delaunay::run_arranged(const int32_t* x, const int32_t* y, int n, const uint16_t* arrangement, int32_t* tri) {
  nparts_ = 0; k_ = 0;
  if (n < 3) return -1;
  x_ = x; y_ = y; lcg_ = 1;
  const size_t cap = (size_t)8 * n + 64;
  if (link_.size() < 4 * cap) { link_.resize(4 * cap); vert_.resize(4 * cap); }
  if (order_.size() < (size_t)n) { order_.resize(n); by_y_.resize(n); tmp_.resize(n); left_.resize(n); }
  for (int i = 0; i < n; i++) order_[i] = arrangement[i];
  k_ = n;
  part_[0] = Part{0, n, 0, 0, 0, 0, 0};
  nparts_ = 1;
  Ctx c{0};
  conquer(order_.data(), n, 0, part_[0].farleft, part_[0].farright, c);
  part_[0].used = c.next;
  return finish(tri);
}

bool Delaunay::arrangement(const int32_t* x, const int32_t* y, int n, uint16_t* out) {
  if (n < 3 || n > 65535) return false;
  x_ = x; y_ = y; lcg_ = 1;
  if (order_.size() < (size_t)n) { order_.resize(n); by_y_.resize(n); tmp_.resize(n); left_.resize(n); }
  int32_t* a = order_.data();
  if (!sort_distinct(a, n)) {                       // sparse keys or coinciding vertices: the reference's sort, then its duplicate removal
    for (int i = 0; i < n; i++) a[i] = i;
    quicksort(a, n);
    int k = 0;
    for (int j = 1; j < n; j++)
      if (x[a[k]] != x[a[j]] || y[a[k]] != y[a[j]]) a[++k] = a[j];
    if (k + 1 != n) return false;
  }
  arrange(a, n);
  split(0, n, 0);
  for (int i = 0; i < n; i++) out[i] = (uint16_t)order_[i];
  return true;
}

int Delaunay::run(const int32_t* x, const int32_t* y, int n, int32_t* tri) {
  if (prepare(x, y, n, 1) == 0) return -1;
  subtree(0);
  return finish(tri);
}

}  // namespace jnav
