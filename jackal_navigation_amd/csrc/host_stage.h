// host_stage.h — the serial middle of ELAS that stays on the host (product code).
//
// Between GPU stage A (descriptors + support matching) and GPU stage B (dense matching and
// post-processing) the reference runs four short, branchy, order-dependent steps per frame:
//   removeInconsistentSupportPoints / removeRedundantSupportPoints   elas.cpp:153-235 (in-place, scan-order dependent)
//   computeDelaunayTriangulation x2                                   elas.cpp:445-505 (Triangle D&C)
// (computeDisparityPlanes and createGrid, elas.cpp:507-659, run on the GPU from the support points
// and corner indices this stage emits.)
// One HostWorker per thread; frames (phase 1) and frame sides (phase 2) are farmed out over a pool.
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
  int32_t add_corners = 0;     // elas.cpp:435: six border points join the support points (MIDDLEBURY preset)
};

// Support points of one frame, shared between the two phases of the host stage.
struct FrameScratch {
  std::vector<int32_t> u, v, d, x;               // x = u - d (right-image column)
};

class HostWorker {
 public:
  explicit HostWorker(const HostParams& hp);
  // Phase 1 (one task per frame): filter this frame's candidate lattice d_can [ch][cw] in place
  // (elas.cpp:416-422) and list the support points (elas.cpp:425-431) into `fs`; sets info->ok/nsup.
  // `filtered`: the GPU already ran the filters (k_support_filters); only the list is built.
  void filter_and_list(int16_t* d_can, FrameInfo* info, FrameScratch* fs, bool filtered = false) const;
  // Between the phases: give the frame its place in the batch payload (frames are packed back to
  // back so that the whole batch goes to the GPU in one copy).  Returns the bytes the frame occupies.
  static size_t place(FrameInfo* info, size_t base_offset);
  // Phase 2 (one task per frame and side): Delaunay triangulation (elas.cpp:445-505) of the points
  // (u,v) for side 0 or (u-d,v) for side 1; corner indices go to the payload (side 0 also writes
  // the support points there).  `payload` is the base the offsets in `info` refer to.
  void triangulate_side(int side, const FrameScratch& fs, uint8_t* payload, FrameInfo* info);
  // The same from the GPU's list (k_support_list), with no shared scratch: the task builds the coordinates it needs from
  // the (uc, vc, d) triples itself, so a batch is one flat set of frame-side tasks.  `info` must hold ok / nsup and
  // its payload offsets already.
  // `arrangement` (optional): the alternating-cut arrangement of this side's vertices computed on the GPU (k_arrange); with it
  // only the hull recursion of the triangulation runs here.
  void triangulate_side_from_list(int side, const int16_t* triples, uint8_t* payload, FrameInfo* info, const uint16_t* arrangement = nullptr);
  // The same split into phases for callers with idle threads (jn_api.cpp decides): coordinates + Delaunay::prepare into
  // the caller's per-(frame, side) state, then Delaunay::subtree per part on any worker, then Delaunay::finish.
  struct SideState { Delaunay dt; std::vector<int32_t> xs, ys; int parts = 0; };
  void side_prepare(int side, const int16_t* triples, uint8_t* payload, const FrameInfo* info, SideState* st, int want_parts) const;
  static void side_finish(int side, uint8_t* payload, FrameInfo* info, SideState* st);
  static size_t payload_capacity(const HostParams& hp);   // worst case for one frame
  // addCornerSupportPoints (elas.cpp:237-267): the four image corners with the disparity of their nearest support point
  // (first minimum of the squared distance in list order), plus the two right-hand corners shifted by their disparity
  // for the right image.  Appends 6 entries to u/v/d (which hold n points).
  static void corner_points(int W, int H, int n, std::vector<int32_t>& u, std::vector<int32_t>& v, std::vector<int32_t>& d);
  enum { kCornerPoints = 6 };

 private:
  HostParams hp_;
  Delaunay dt_;
  std::vector<int32_t> xs_, ys_, ds_;            // coordinates of one frame side (triangulate_side_from_list)
  mutable std::vector<int16_t> tr_;              // transposed lattice for the horizontal redundancy pass
  void filter_inconsistent(int16_t* D) const;
  void filter_redundant(int16_t* D, int max_dist, int thresh, bool vertical) const;
};

}  // namespace jnav
