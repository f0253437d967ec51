// host_stage.h — the serial middle of ELAS that stays on the host (product code).
//
// Between GPU stage A (descriptors + support matching) and GPU stage B (dense matching and
// post-processing) the reference runs four short, branchy, order-dependent steps per frame:
//   removeInconsistentSupportPoints / removeRedundantSupportPoints   elas.cpp:153-235 (in-place, scan-order dependent)
//   computeDelaunayTriangulation x2                                   elas.cpp:445-505 (Triangle D&C)
//   computeDisparityPlanes x2                                         elas.cpp:507-577 (Gauss-Jordan, double)
//   createGrid x2                                                     elas.cpp:579-659
// One HostWorker per thread; frames of a batch are farmed out over a pool.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#include "jn_types.h"
#include "delaunay.h"

namespace jnav {

struct HostParams {
  int32_t W, H;
  int32_t disp_max, step;
  int32_t incon_window_size, incon_threshold, incon_min_support;
  int32_t grid_size, gw, gh, cw, ch;
};

class HostWorker {
 public:
  explicit HostWorker(const HostParams& hp);
  // d_can: this frame's candidate lattice [ch][cw] as produced by the GPU (modified in place).
  // payload: pinned staging for this frame (capacity payload_capacity(hp)); info: filled in.
  void run(int16_t* d_can, uint8_t* payload, FrameInfo* info);
  static size_t payload_capacity(const HostParams& hp);

 private:
  HostParams hp_;
  Delaunay dt_;
  std::vector<int32_t> su_, sv_, sd_, sx_;       // support points (u, v, d) and u-d
  std::vector<int32_t> tri_;
  std::vector<uint32_t> mark_;
  void filter_inconsistent(int16_t* D) const;
  void filter_redundant(int16_t* D, int max_dist, int thresh, bool vertical) const;
  int  make_side(int side, TriRec* out);
  void make_grid(int side, uint32_t* bits);
};

// Plane through three support points (Gauss-Jordan with full pivoting in double, matrix.cpp:414-502
// semantics).  Exposed for tests.
bool solve_plane(const double rows[3][3], const double rhs[3], float out[3]);

}  // namespace jnav
