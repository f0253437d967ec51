// jn_types.h — plain structs shared by the host stage (C++) and the HIP kernels.
#pragma once
#include <stdint.h>

namespace jnav {

// One Delaunay triangle, ready for the rasteriser/dense-matching kernels.  The host stage does the
// corner sort and the slope/intercept divisions of reference elas.cpp:847-868 in IEEE float, the
// GPU only evaluates a*u+b (elas.cpp:878-879, :893-894) and the plane (elas.cpp:722).
struct TriRec {
  int16_t Au, Bu, Cu;        // corner columns after the ascending-u sort (elas.cpp:847-859)
  uint16_t flags;            // bit0: plane prior valid (elas.cpp:872)
  float ACa, ACb, ABa, ABb, BCa, BCb;   // edge lines v = a*u + b (elas.cpp:862-868)
  float pa, pb, pc;          // disparity plane of this side (elas.cpp:817-827)
  int16_t vmin, vmax;        // conservative inclusive row range of the rasterised triangle (for binning)
};
static_assert(sizeof(TriRec) == 48, "TriRec layout");

enum { kGridWords = 8 };     // 256-bit candidate set per grid cell (disp_max <= 255)
enum { kTileW = 32, kTileH = 8 };   // dense matching works on 32x8 pixel tiles = one 256-thread workgroup
enum { kBinCap = 64 };       // triangle candidates kept per tile; longer lists fall back to a full scan

// One candidate triangle of one 32x8 tile: byte x of `rows` has bit r set iff the reference's raster
// loops (elas.cpp:874-901) visit pixel (tile_u0 + x, tile_v0 + r) for triangle t.
// Padded to 64 bytes and 16-byte aligned: an entry is written as four 16-byte stores into one cache line
// (13 scattered dword stores per entry made k_bin store-bound).
struct alignas(16) BinEntry {
  int32_t t;
  uint32_t rows[kTileW / 4];
  float pa, pb, pc;          // the triangle's disparity plane and validity flag ride along so that the
  uint32_t flags;            // matcher needs no dependent TriRec fetch after the lookup
  uint32_t pad[3];
};
static_assert(sizeof(BinEntry) == 64, "BinEntry layout");
enum { kBinWords = 16 };     // sizeof(BinEntry) / 4
enum { kBinLds = 16 };       // list entries per tile the matcher resolves from LDS (one 16-bit cover word per pixel); longer lists (never seen: mean 7, max 17 per 32x8 tile at 720p) are read from global memory

// Per-frame bookkeeping uploaded before GPU stage B.  The frame payload the host stage produces is
//   [support points: nsup x (u,v,d) int32][left corners: ntri[0] x 3 int32][right corners: ntri[1] x 3 int32]
struct FrameInfo {
  int32_t ok;                // 0: fewer than 3 support points -> outputs stay untouched (elas.cpp:66-71)
  int32_t nsup;
  int32_t ntri[2];           // left, right
  int64_t sup_offset;        // byte offset of the support points inside the frame payload
  int64_t corner_offset[2];  // byte offset of each side's triangle corner indices
  int64_t reserved;          // GPU triangulation route: what the frame's support list held, clipped or not (k_delaunay writes it; the host sizes the next launches by it)
};

// Kernel-side view of the tunables (Elas::parameters subset + derived constants).
struct DevParams {
  int32_t W, H, pitch;       // image width/height, bytes per internal image row
  int32_t disp_max;
  int32_t disp_min;                 // max(param.disp_min, 0): first disparity the support matching tries (elas.cpp:323; nothing else reads it)
  int32_t support_texture, step, lr_threshold;
  float   support_threshold;
  int32_t cw, ch;            // candidate lattice size (elas.cpp:384-387)
  int32_t grid_size, gw, gh;
  int32_t match_texture;
  int32_t radius;            // plane_radius (elas.cpp:806)
  int32_t P[8];              // prior table P[|d-d_plane|] (elas.cpp:802-805), entries 0..radius
  float   speckle_sim;
  int32_t speckle_size;
  int32_t gap_width;
  int32_t add_corners;       // elas.cpp:1169, :1253: gap interpolation also extrapolates to the image borders
  uint32_t grid_magic;       // floor(2^32 / grid_size) + 1: x / grid_size == __umulhi(x, grid_magic) for 0 <= x < 2^32 / grid_size
};

}  // namespace jnav
