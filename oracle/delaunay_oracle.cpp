// oracle/delaunay_oracle.cpp — TEST INFRASTRUCTURE (CPU restatement), not product code.
//
// Restates the one code path of Shewchuk's Triangle that libelas exercises through
// triangulate("zQB") (reference: src/elas/elas.cpp:445-505 -> src/elas/triangle.cpp:8499):
// divide-and-conquer Delaunay with alternating cuts, bounded by "ghost" triangles.
//
// ELAS feeds lattice points (multiples of candidate_stepsize), so cocircular quadruples are the
// norm and the triangulation is NOT unique: the answer is whatever Triangle's tie-breaks give.
// This file therefore follows the reference's *decisions* one for one:
//   * randomised quicksort with the LCG pivot (triangle.cpp:4045-4049, 5446-5500) — it decides
//     which of two duplicate right-image vertices survives (triangle.cpp:6179-6194);
//   * median partition + alternating axes (triangle.cpp:5514-5606);
//   * base cases of 2/3 vertices and the hull merge (triangle.cpp:5638-5947, 5953-6103);
//   * ghost removal == "drop every triangle that still has a missing corner"
//     (triangle.cpp:6105-6148), triangles emitted in creation order as (org,dest,apex) of the
//     orientation-0 handle (triangle.cpp:7832-7843).
// The data structure is ours (flat index arrays, no pointer tagging).  Orientation/in-circle
// predicates (triangle.cpp:2706-2745, 3334-3379) are adaptive-exact in the reference; on integer
// coordinates their sign equals the sign of the integer determinant, which is what we compute.
//
// Pinned against oracle/_ref (the compiled reference) by tests/test_oracle_vs_reference.py.
#include "oracle.h"
#include <cstdint>
#include <cmath>
#include <vector>

namespace {

typedef __int128 wide_t;

struct Edge { int t, o; };                       // oriented triangle: index + edge number 0..2
static const int NXT[3] = {1, 2, 0};
static const int PRV[3] = {2, 0, 1};

struct DCMesh {
  const int64_t* X; const int64_t* Y;             // vertex coordinates
  std::vector<int> adj;                            // 3 per triangle: 4*t+o of the neighbour, -1 = none
  std::vector<int> corner;                         // 3 per triangle: vertex id, -1 = missing (ghost)
  uint64_t seed = 1;                               // triangle.cpp:4030
  bool fault = false;                              // would have been a NULL dereference upstream

  // --- handle algebra (triangle.cpp:842-992 semantics) ---
  Edge make() { int t = (int)(corner.size() / 3); for (int i = 0; i < 3; i++) { adj.push_back(-1); corner.push_back(-1); } return Edge{t, 0}; }
  Edge sym(Edge e) { int c = adj[3 * e.t + e.o]; if (c < 0) { fault = true; return e; } return Edge{c >> 2, c & 3}; }
  static Edge nxt(Edge e) { return Edge{e.t, NXT[e.o]}; }
  static Edge prv(Edge e) { return Edge{e.t, PRV[e.o]}; }
  int org(Edge e) const { return corner[3 * e.t + NXT[e.o]]; }
  int dst(Edge e) const { return corner[3 * e.t + PRV[e.o]]; }
  int apx(Edge e) const { return corner[3 * e.t + e.o]; }
  void set_org(Edge e, int v) { corner[3 * e.t + NXT[e.o]] = v; }
  void set_dst(Edge e, int v) { corner[3 * e.t + PRV[e.o]] = v; }
  void set_apx(Edge e, int v) { corner[3 * e.t + e.o] = v; }
  void bond(Edge a, Edge b) { adj[3 * a.t + a.o] = 4 * b.t + b.o; adj[3 * b.t + b.o] = 4 * a.t + a.o; }

  int64_t x(int v) { if (v < 0) { fault = true; return 0; } return X[v]; }
  int64_t y(int v) { if (v < 0) { fault = true; return 0; } return Y[v]; }

  // sign of orient2d(a,b,c): >0 counter-clockwise (triangle.cpp:2706)
  int ccw(int a, int b, int c) {
    wide_t l = (wide_t)(x(a) - x(c)) * (y(b) - y(c));
    wide_t r = (wide_t)(y(a) - y(c)) * (x(b) - x(c));
    return (l > r) - (l < r);
  }
  // sign of incircle(a,b,c,d): >0 if d strictly inside circle abc (a,b,c ccw) (triangle.cpp:3334)
  int incirc(int a, int b, int c, int d) {
    wide_t adx = x(a) - x(d), ady = y(a) - y(d);
    wide_t bdx = x(b) - x(d), bdy = y(b) - y(d);
    wide_t cdx = x(c) - x(d), cdy = y(c) - y(d);
    wide_t al = adx * adx + ady * ady, bl = bdx * bdx + bdy * bdy, cl = cdx * cdx + cdy * cdy;
    wide_t det = al * (bdx * cdy - bdy * cdx) + bl * (cdx * ady - cdy * adx) + cl * (adx * bdy - ady * bdx);
    return (det > 0) - (det < 0);
  }

  // triangle.cpp:4045-4049
  uint64_t pick(unsigned choices) { seed = (seed * 1366u + 150889u) % 714025u; return seed / (714025u / choices + 1); }

  // lexicographic "a before b" on (axis, other axis)
  bool before(int a, int b, int axis) {
    int64_t a1 = axis ? Y[a] : X[a], b1 = axis ? Y[b] : X[b];
    if (a1 != b1) return a1 < b1;
    int64_t a2 = axis ? X[a] : Y[a], b2 = axis ? X[b] : Y[b];
    return a2 < b2;
  }

  // Hoare partition around a random pivot; returns (left,right) as the reference leaves them.
  void split(int* a, int n, int axis, int& left, int& right) {
    int p = a[(int)pick((unsigned)n)];
    left = -1; right = n;
    while (left < right) {
      do { left++; } while (left <= right && before(a[left], p, axis));
      do { right--; } while (left <= right && before(p, a[right], axis));
      if (left < right) { int t = a[left]; a[left] = a[right]; a[right] = t; }
    }
  }
  // triangle.cpp:5446-5500
  void sort_xy(int* a, int n) {
    if (n == 2) { if (before(a[1], a[0], 0)) { int t = a[0]; a[0] = a[1]; a[1] = t; } return; }
    int l, r; split(a, n, 0, l, r);
    if (l > 1) sort_xy(a, l);
    if (r < n - 2) sort_xy(a + r + 1, n - r - 1);
  }
  // triangle.cpp:5514-5572
  void median(int* a, int n, int m, int axis) {
    if (n == 2) { if (before(a[1], a[0], axis)) { int t = a[0]; a[0] = a[1]; a[1] = t; } return; }
    int l, r; split(a, n, axis, l, r);
    if (l > m) median(a, l, m, axis);
    if (r < m - 1) median(a + r + 1, n - r - 1, m - r - 1, axis);
  }
  // triangle.cpp:5586-5606
  void alternate(int* a, int n, int axis) {
    int half = n >> 1;
    if (n <= 3) axis = 0;
    median(a, n, half, axis);
    if (n - half >= 2) {
      if (half >= 2) alternate(a, half, 1 - axis);
      alternate(a + half, n - half, 1 - axis);
    }
  }

  void merge(Edge& farleft, Edge& innerleft, Edge& innerright, Edge& farright, int axis);
  void build(int* a, int n, int axis, Edge& farleft, Edge& farright);
};

// triangle.cpp:5638-5947.  Zips two hulls together, bottom to top.
void DCMesh::merge(Edge& farleft, Edge& innerleft, Edge& innerright, Edge& farright, int axis) {
  int ild = dst(innerleft), ila = apx(innerleft);
  int iro = org(innerright), ira = apx(innerright);

  if (axis == 1) {
    // horizontal cut: re-aim the four hull handles at bottom-most / top-most vertices (:5681-5719)
    int flp = org(farleft), fla = apx(farleft);
    int frp = dst(farright), fra = apx(farright);
    while (y(fla) < y(flp)) {
      farleft = sym(nxt(farleft));
      flp = fla; fla = apx(farleft);
    }
    Edge chk = sym(innerleft); int cv = apx(chk);
    while (y(cv) > y(ild)) {
      innerleft = nxt(chk);
      ila = ild; ild = cv;
      chk = sym(innerleft); cv = apx(chk);
    }
    while (y(ira) < y(iro)) {
      innerright = sym(nxt(innerright));
      iro = ira; ira = apx(innerright);
    }
    chk = sym(farright); cv = apx(chk);
    while (y(cv) > y(frp)) {
      farright = nxt(chk);
      fra = frp; frp = cv;
      chk = sym(farright); cv = apx(chk);
    }
    (void)fra;
    if (fault) return;
  }

  // lower common tangent (:5721-5743)
  bool moved;
  do {
    moved = false;
    if (ccw(ild, ila, iro) > 0) {
      innerleft = sym(prv(innerleft));
      ild = ila; ila = apx(innerleft); moved = true;
    }
    if (ccw(ira, iro, ild) > 0) {
      innerright = sym(nxt(innerright));
      iro = ira; ira = apx(innerright); moved = true;
    }
    if (fault) return;
  } while (moved);

  Edge lcand = sym(innerleft), rcand = sym(innerright);
  // bottom ghost of the merged hull (:5748-5759)
  Edge base = make();
  bond(base, innerleft);  base = nxt(base);
  bond(base, innerright); base = nxt(base);
  set_org(base, iro); set_dst(base, ild);
  if (ild == org(farleft))  farleft = nxt(base);
  if (iro == dst(farright)) farright = prv(base);

  int ll = ild, lr = iro;
  int ul = apx(lcand), ur = apx(rcand);

  for (;;) {
    bool ldone = ccw(ul, ll, lr) <= 0;
    bool rdone = ccw(ur, ll, lr) <= 0;
    if (fault) return;
    if (ldone && rdone) {
      // top ghost (:5790-5803)
      Edge top = make();
      set_org(top, ll); set_dst(top, lr);
      bond(top, base);  top = nxt(top);
      bond(top, rcand); top = nxt(top);
      bond(top, lcand);
      if (axis == 1) {
        // restore left-most / right-most handles (:5809-5833)
        int flp = org(farleft), fla = apx(farleft);
        int frp = dst(farright), fra = apx(farright);
        Edge chk = sym(farleft); int cv = apx(chk);
        while (x(cv) < x(flp)) {
          farleft = prv(chk);
          fla = flp; flp = cv;
          chk = sym(farleft); cv = apx(chk);
        }
        while (x(fra) > x(frp)) {
          farright = sym(prv(farright));
          frp = fra; fra = apx(farright);
        }
        (void)fla;
      }
      return;
    }
    if (!ldone) {
      // peel non-Delaunay edges off the left hull (:5837-5880)
      Edge ne = sym(prv(lcand));
      int na = apx(ne);
      if (na >= 0) {
        bool bad = incirc(ll, lr, ul, na) > 0;
        while (bad) {
          ne = nxt(ne); Edge topc = sym(ne);
          ne = nxt(ne); Edge sidec = sym(ne);
          bond(ne, topc);
          bond(lcand, sidec);
          lcand = nxt(lcand); Edge outerc = sym(lcand);
          ne = prv(ne);
          bond(ne, outerc);
          set_org(lcand, ll); set_dst(lcand, -1); set_apx(lcand, na);
          set_org(ne, -1);    set_dst(ne, ul);    set_apx(ne, na);
          ul = na;
          ne = sidec; na = apx(ne);
          bad = (na >= 0) ? (incirc(ll, lr, ul, na) > 0) : false;
          if (fault) return;
        }
      }
    }
    if (!rdone) {
      // peel non-Delaunay edges off the right hull (:5883-5926)
      Edge ne = sym(nxt(rcand));
      int na = apx(ne);
      if (na >= 0) {
        bool bad = incirc(ll, lr, ur, na) > 0;
        while (bad) {
          ne = prv(ne); Edge topc = sym(ne);
          ne = prv(ne); Edge sidec = sym(ne);
          bond(ne, topc);
          bond(rcand, sidec);
          rcand = prv(rcand); Edge outerc = sym(rcand);
          ne = nxt(ne);
          bond(ne, outerc);
          set_org(rcand, -1); set_dst(rcand, lr); set_apx(rcand, na);
          set_org(ne, ur);    set_dst(ne, -1);    set_apx(ne, na);
          ur = na;
          ne = sidec; na = apx(ne);
          bad = (na >= 0) ? (incirc(ll, lr, ur, na) > 0) : false;
          if (fault) return;
        }
      }
    }
    if (ldone || (!rdone && incirc(ul, ll, lr, ur) > 0)) {
      // new cross edge ll -> ur (:5930-5937)
      bond(base, rcand);
      base = prv(rcand);
      set_dst(base, ll);
      lr = ur;
      rcand = sym(base);
      ur = apx(rcand);
    } else {
      // new cross edge ul -> lr (:5938-5946)
      bond(base, lcand);
      base = nxt(lcand);
      set_org(base, lr);
      ll = ul;
      lcand = sym(base);
      ul = apx(lcand);
    }
    if (fault) return;
  }
}

// triangle.cpp:5953-6103
void DCMesh::build(int* a, int n, int axis, Edge& farleft, Edge& farright) {
  if (fault) return;
  if (n == 2) {
    farleft = make();  set_org(farleft, a[0]);  set_dst(farleft, a[1]);
    farright = make(); set_org(farright, a[1]); set_dst(farright, a[0]);
    bond(farleft, farright);
    farleft = prv(farleft); farright = nxt(farright); bond(farleft, farright);
    farleft = prv(farleft); farright = nxt(farright); bond(farleft, farright);
    farleft = prv(farright);
    return;
  }
  if (n == 3) {
    Edge mid = make(), t1 = make(), t2 = make(), t3 = make();
    int area = ccw(a[0], a[1], a[2]);
    if (area == 0) {
      // collinear: two edges, four ghosts (:6000-6029)
      set_org(mid, a[0]); set_dst(mid, a[1]);
      set_org(t1, a[1]);  set_dst(t1, a[0]);
      set_org(t2, a[2]);  set_dst(t2, a[1]);
      set_org(t3, a[1]);  set_dst(t3, a[2]);
      bond(mid, t1); bond(t2, t3);
      mid = nxt(mid); t1 = prv(t1); t2 = nxt(t2); t3 = prv(t3);
      bond(mid, t3); bond(t1, t2);
      mid = nxt(mid); t1 = prv(t1); t2 = nxt(t2); t3 = prv(t3);
      bond(mid, t1); bond(t2, t3);
      farleft = t1; farright = t2;
    } else {
      // one real triangle `mid`, three ghosts (:6030-6078)
      set_org(mid, a[0]); set_dst(t1, a[0]); set_org(t3, a[0]);
      int p = area > 0 ? a[1] : a[2], q = area > 0 ? a[2] : a[1];
      set_dst(mid, p); set_org(t1, p); set_dst(t2, p);
      set_apx(mid, q); set_org(t2, q); set_dst(t3, q);
      bond(mid, t1); mid = nxt(mid);
      bond(mid, t2); mid = nxt(mid);
      bond(mid, t3);
      t1 = prv(t1); t2 = nxt(t2); bond(t1, t2);
      t1 = prv(t1); t3 = prv(t3); bond(t1, t3);
      t2 = nxt(t2); t3 = prv(t3); bond(t2, t3);
      farleft = t1;
      farright = area > 0 ? t2 : nxt(farleft);
    }
    return;
  }
  int half = n >> 1;
  Edge il, ir;
  build(a, half, 1 - axis, farleft, il);
  build(a + half, n - half, 1 - axis, ir, farright);
  if (fault) return;
  merge(farleft, il, ir, farright, axis);
}

}  // namespace

// Returns the triangle count (may exceed cap; only cap are written), or -1 on unsupported input
// (non-integral coordinates, fewer than two distinct points).
extern "C" int32_t orc_triangulate(const float* xy, int32_t n, int32_t* corners, int32_t cap) {
  if (n < 3) return -1;                            // Triangle would triexit (triangle.cpp:7648)
  std::vector<int64_t> X(n), Y(n);
  for (int i = 0; i < n; i++) {
    float fx = xy[2 * i], fy = xy[2 * i + 1];
    if (fx != std::floor(fx) || fy != std::floor(fy) || std::fabs(fx) > 1e6f || std::fabs(fy) > 1e6f) return -1;
    X[i] = (int64_t)fx; Y[i] = (int64_t)fy;
  }
  DCMesh m; m.X = X.data(); m.Y = Y.data();
  m.adj.reserve(6 * (size_t)n * 3); m.corner.reserve(6 * (size_t)n * 3);
  std::vector<int> order(n);
  for (int i = 0; i < n; i++) order[i] = i;        // pool traversal == input order (triangle.cpp:6171-6174)
  m.sort_xy(order.data(), n);
  int k = 0;                                        // drop duplicates, first in sorted order survives (:6179-6194)
  for (int j = 1; j < n; j++)
    if (!(X[order[k]] == X[order[j]] && Y[order[k]] == Y[order[j]])) order[++k] = order[j];
  k++;
  if (k < 2) return -1;                             // reference recurses forever here
  int half = k >> 1;                                // alternating cuts (:6197-6206)
  if (k - half >= 2) {
    if (half >= 2) m.alternate(order.data(), half, 1);
    m.alternate(order.data() + half, k - half, 1);
  }
  Edge hl, hr;
  m.build(order.data(), k, 0, hl, hr);
  if (m.fault) return -1;
  // ghosts are exactly the triangles with a missing corner; real ones leave in creation order
  int32_t nt = 0;
  size_t T = m.corner.size() / 3;
  for (size_t t = 0; t < T; t++) {
    int c0 = m.corner[3 * t], c1 = m.corner[3 * t + 1], c2 = m.corner[3 * t + 2];
    if (c0 < 0 || c1 < 0 || c2 < 0) continue;
    if (nt < cap) { corners[3 * nt] = c1; corners[3 * nt + 1] = c2; corners[3 * nt + 2] = c0; }
    nt++;
  }
  return nt;
}
