// delaunay.h — host-side triangulation of the support points (product code).
//
// ELAS triangulates its support points with Shewchuk's Triangle (reference: src/elas/elas.cpp:445-505
// calling triangulate("zQB"), src/elas/triangle.cpp:8499).  Support points sit on a 5-pixel lattice,
// so the Delaunay triangulation is not unique and every downstream disparity depends on the exact
// tie-breaks of Triangle's divide-and-conquer with alternating cuts.  This class is a re-entrant,
// allocation-free (after reserve) implementation of that same decision sequence on integer
// coordinates, so that one instance per host worker thread can run concurrently — the reference
// keeps its RNG seed and predicate constants in globals (triangle.cpp:541-550) and cannot.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace jnav {

class Delaunay {
 public:
  // Triangulate n points (x[i], y[i]), |coord| < 2^15.  Writes (org,dest,apex) vertex indices of
  // every triangle, in Triangle's output order, to tri (capacity 3*2*n ints).  Returns the number
  // of triangles, or -1 when fewer than 2 distinct points exist.
  int run(const int32_t* x, const int32_t* y, int n, int32_t* tri);
  // The same when the alternating-cut arrangement of the n DISTINCT vertices is already known (the GPU computes it from the
  // support list, kernels.hip k_arrange): arrangement[i] = vertex at position i of what arrange() + split() would leave in
  // order_.  Only the hull recursion runs.
  int run_arranged(const int32_t* x, const int32_t* y, int n, const uint16_t* arrangement, int32_t* tri);
  // The arrangement run() would use (test hook): returns false when vertices coincide (run() then replays Triangle's sort).
  bool arrangement(const int32_t* x, const int32_t* y, int n, uint16_t* out);

  // The same in three phases, for a caller with idle threads (a lone pair, a small batch of large frames):
  //   prepare()   serial: sort, duplicate removal, the first one or two alternating cuts -> 1, 2 or 4 independent parts
  //   subtree(i)  the parts may run concurrently on different threads (they touch disjoint ranges of every array)
  //   finish()    serial: the one or three hull merges between the parts, then the output
  // Triangle numbers (= the reference's creation order, which decides doubly covered pixels downstream) are kept by
  // giving every part, and every merge between parts, its own slot range in the order the sequential recursion would
  // have reached it.  prepare returns the number of parts, 0 when there is nothing to do (finish then returns -1).
  int prepare(const int32_t* x, const int32_t* y, int n, int want_parts);
  void subtree(int part);
  int finish(int32_t* tri);

 private:
  typedef uint32_t H;                       // oriented triangle handle: (triangle << 2) | edge
  const int32_t* x_ = nullptr;
  const int32_t* y_ = nullptr;
  std::vector<int32_t> link_;               // 4 per triangle (3 used): handle across that edge; indexed by the handle itself
  std::vector<int32_t> vert_;               // 4 per triangle (3 used): vertex or -1 (ghost corner)
  std::vector<int32_t> order_, by_y_, tmp_, bucket_;
  std::vector<uint8_t> left_;
  uint64_t lcg_ = 1;
  struct Ctx { int next; };                 // next free triangle slot of a slot range
  struct Part { int lo, hi, axis, slot0, used; H farleft, farright; };
  Part part_[4];
  int nparts_ = 0, k_ = 0;
  int zslot_[3] = {0, 0, 0};                // slot ranges (2 slots each) of the merges between parts, in creation order

  H fresh(Ctx& c);
  inline H across(H h) const { return (H)link_[h]; }
  static inline unsigned up(unsigned e) { return e == 2 ? 0u : e + 1u; }      // edge 0->1->2->0
  static inline unsigned down(unsigned e) { return e == 0 ? 2u : e - 1u; }    // edge 0->2->1->0
  static inline H ccw_edge(H h) { return (h & ~3u) | up(h & 3); }
  static inline H cw_edge(H h)  { return (h & ~3u) | down(h & 3); }
  inline int32_t& v_org(H h)  { return vert_[ccw_edge(h)]; }
  inline int32_t& v_dest(H h) { return vert_[cw_edge(h)]; }
  inline int32_t& v_apex(H h) { return vert_[h]; }
  inline void glue(H a, H b) { link_[a] = (int32_t)b; link_[b] = (int32_t)a; }

  inline int orient(int a, int b, int c) const;
  inline int in_circle(int a, int b, int c, int d) const;
  inline bool precedes(int a, int b, int axis) const;
  unsigned draw(unsigned choices);
  void partition(int32_t* a, int n, int axis, int& l, int& r);
  void quicksort(int32_t* a, int n);
  bool sort_distinct(int32_t* a, int n);
  void arrange(int32_t* a, int n);
  void cut(int lo, int hi, int axis);
  void split(int lo, int hi, int axis);
  void conquer(int32_t* a, int n, int axis, H& farleft, H& farright, Ctx& c);
  void zip(H& farleft, H& innerleft, H& innerright, H& farright, int axis, Ctx& c);
};

}  // namespace jnav
