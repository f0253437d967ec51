// host_stage.cpp — see host_stage.h.  Compile with -ffp-contract=off: plane and edge arithmetic
// must round exactly like the reference's non-FMA build.
#include "host_stage.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>

namespace jnav {

HostWorker::HostWorker(const HostParams& hp) : hp_(hp) {
  const size_t maxsup = (size_t)hp.cw * hp.ch;
  su_.reserve(maxsup); sv_.reserve(maxsup); sd_.reserve(maxsup); sx_.reserve(maxsup);
  tri_.resize(6 * maxsup + 64);
  mark_.resize((size_t)hp.gw * hp.gh * kGridWords);
}

size_t HostWorker::payload_capacity(const HostParams& hp) {
  const size_t maxsup = (size_t)hp.cw * hp.ch;
  return 2 * (size_t)hp.gw * hp.gh * kGridWords * sizeof(uint32_t) + 2 * (2 * maxsup + 8) * sizeof(TriRec);
}

// elas.cpp:153-179.  Column-major sweep over the lattice; a point survives if at least
// incon_min_support lattice points (itself included) in the (2w+1)^2 window agree within
// incon_threshold.  Deletions take effect immediately, so the sweep order is part of the result.
void HostWorker::filter_inconsistent(int16_t* D) const {
  const int cw = hp_.cw, ch = hp_.ch, win = hp_.incon_window_size, tol = hp_.incon_threshold;
  for (int u = 0; u < cw; u++) {
    const int u0 = std::max(u - win, 0), u1 = std::min(u + win, cw - 1);
    for (int v = 0; v < ch; v++) {
      const int d = D[v * cw + u];
      if (d < 0) continue;
      const int v0 = std::max(v - win, 0), v1 = std::min(v + win, ch - 1);
      int agree = 0;
      for (int vv = v0; vv <= v1; vv++) {
        const int16_t* row = D + vv * cw;
        for (int uu = u0; uu <= u1; uu++) {
          const int e = row[uu];
          agree += (e >= 0) & (std::abs(d - e) <= tol);
        }
      }
      if (agree < hp_.incon_min_support) D[v * cw + u] = -1;
    }
  }
}

// elas.cpp:181-235.  A point is redundant when, walking up to max_dist lattice steps in BOTH
// directions along one axis, a point within `thresh` disparity is met first.
void HostWorker::filter_redundant(int16_t* D, int max_dist, int thresh, bool vertical) const {
  const int cw = hp_.cw, ch = hp_.ch;
  const int stride = vertical ? cw : 1;
  for (int u = 0; u < cw; u++)
    for (int v = 0; v < ch; v++) {
      int16_t* p = D + v * cw + u;
      const int d = *p;
      if (d < 0) continue;
      const int pos = vertical ? v : u, len = vertical ? ch : cw;
      bool both = true;
      for (int dir = -1; dir <= 1 && both; dir += 2) {
        bool found = false;
        for (int j = 1; j <= max_dist; j++) {
          const int q = pos + dir * j;
          if (q < 0 || q >= len) break;
          const int e = p[dir * j * stride];
          if (e >= 0 && std::abs(d - e) <= thresh) { found = true; break; }
        }
        both = found;
      }
      if (both) *p = -1;
    }
}

bool solve_plane(const double rows[3][3], const double rhs[3], float out[3]) {
  double A[3][3], b[3];
  memcpy(A, rows, sizeof(A)); memcpy(b, rhs, sizeof(b));
  int used[3] = {0, 0, 0};
  for (int step = 0; step < 3; step++) {
    double best = 0.0; int pr = 0, pc = 0;
    for (int r = 0; r < 3; r++) {
      if (used[r] == 1) continue;
      for (int c = 0; c < 3; c++)
        if (used[c] == 0 && std::fabs(A[r][c]) >= best) { best = std::fabs(A[r][c]); pr = r; pc = c; }
    }
    ++used[pc];
    if (pr != pc) {
      for (int c = 0; c < 3; c++) std::swap(A[pr][c], A[pc][c]);
      std::swap(b[pr], b[pc]);
    }
    if (std::fabs(A[pc][pc]) < 1e-20) return false;
    const double inv = 1.0 / A[pc][pc];
    A[pc][pc] = 1.0;
    for (int c = 0; c < 3; c++) A[pc][c] *= inv;
    b[pc] *= inv;
    for (int r = 0; r < 3; r++) {
      if (r == pc) continue;
      const double f = A[r][pc];
      A[r][pc] = 0.0;
      for (int c = 0; c < 3; c++) A[r][c] -= A[pc][c] * f;
      b[r] -= b[pc] * f;
    }
  }
  out[0] = (float)b[0]; out[1] = (float)b[1]; out[2] = (float)b[2];
  return true;
}

// Triangulate one side and emit raster-ready records.  side 0: points (u,v); side 1: (u-d,v).
int HostWorker::make_side(int side, TriRec* out) {
  const int n = (int)su_.size();
  const int32_t* xs = side ? sx_.data() : su_.data();
  const int nt = dt_.run(xs, sv_.data(), n, tri_.data());
  if (nt < 0) return 0;
  for (int t = 0; t < nt; t++) {
    const int32_t* c = &tri_[3 * t];
    // both plane fits (elas.cpp:507-577): left-image coordinates and right-image coordinates
    float pl[2][3];
    for (int s = 0; s < 2; s++) {
      double rows[3][3], rhs[3];
      for (int r = 0; r < 3; r++) {
        rows[r][0] = s ? sx_[c[r]] : su_[c[r]]; rows[r][1] = sv_[c[r]]; rows[r][2] = 1; rhs[r] = sd_[c[r]];
      }
      if (!solve_plane(rows, rhs, pl[s])) pl[s][0] = pl[s][1] = pl[s][2] = 0;
    }
    TriRec& o = out[t];
    const float* mine = pl[side];
    const float other_a = pl[1 - side][0];
    o.pa = mine[0]; o.pb = mine[1]; o.pc = mine[2];
    o.flags = (std::fabs(mine[0]) < 0.7 && std::fabs(other_a) < 0.7) ? 1 : 0;       // elas.cpp:872
    float tu[3], tv[3];
    for (int k = 0; k < 3; k++) { tu[k] = (float)xs[c[k]]; tv[k] = (float)sv_[c[k]]; }
    for (int j = 0; j < 3; j++)                                                     // elas.cpp:847-854
      for (int k = 0; k < j; k++)
        if (tu[k] > tu[j]) { std::swap(tu[j], tu[k]); std::swap(tv[j], tv[k]); }
    float ABa = 0, ACa = 0, BCa = 0;                                                // elas.cpp:862-868
    if ((int32_t)tu[0] != (int32_t)tu[1]) ABa = (tv[0] - tv[1]) / (tu[0] - tu[1]);
    if ((int32_t)tu[0] != (int32_t)tu[2]) ACa = (tv[0] - tv[2]) / (tu[0] - tu[2]);
    if ((int32_t)tu[1] != (int32_t)tu[2]) BCa = (tv[1] - tv[2]) / (tu[1] - tu[2]);
    o.ABa = ABa; o.ACa = ACa; o.BCa = BCa;
    o.ABb = tv[0] - ABa * tu[0]; o.ACb = tv[0] - ACa * tu[0]; o.BCb = tv[1] - BCa * tu[1];
    o.Au = (int16_t)tu[0]; o.Bu = (int16_t)tu[1]; o.Cu = (int16_t)tu[2];
    o.pad = 0;
  }
  return nt;
}

// elas.cpp:579-659 on bitsets: mark d-1..d+1 per support point, OR-dilate 3x3 over the FLATTENED
// cell index exactly like the reference's pointer walk (border columns wrap, first/last cell row
// stay empty), and hand the GPU one 256-bit candidate set per cell.
void HostWorker::make_grid(int side, uint32_t* bits) {
  const int gw = hp_.gw, gh = hp_.gh, gs = hp_.grid_size, dmax = hp_.disp_max;
  const size_t cells = (size_t)gw * gh;
  std::fill(mark_.begin(), mark_.begin() + cells * kGridWords, 0u);
  const int n = (int)su_.size();
  for (int i = 0; i < n; i++) {
    const int d = sd_[i];
    const int x = side ? (int)std::floor((float)(su_[i] - d) / (float)gs) : (int)std::floor((float)(su_[i] / gs));
    const int y = (int)std::floor((float)sv_[i] / (float)gs);
    if (x < 0 || x >= gw || y < 0 || y >= gh) continue;
    uint32_t* cell = &mark_[((size_t)y * gw + x) * kGridWords];
    for (int dd = std::max(d - 1, 0); dd <= std::min(d + 1, dmax); dd++) cell[dd >> 5] |= 1u << (dd & 31);
  }
  memset(bits, 0, cells * kGridWords * sizeof(uint32_t));
  const long first = gw + 1, last = (long)cells - gw - 1;        // flattened cells that receive a result
  const long off[9] = {-gw - 1, -gw, -gw + 1, -1, 0, 1, gw - 1, gw, gw + 1};
  for (long c = first; c < last; c++) {
    uint32_t* o = bits + c * kGridWords;
    for (int k = 0; k < 9; k++) {
      const uint32_t* s = &mark_[(c + off[k]) * kGridWords];
      for (int w = 0; w < kGridWords; w++) o[w] |= s[w];
    }
  }
}

void HostWorker::run(int16_t* d_can, uint8_t* payload, FrameInfo* info) {
  const int cw = hp_.cw, ch = hp_.ch, step = hp_.step;
  filter_inconsistent(d_can);                                    // elas.cpp:416
  filter_redundant(d_can, 5, 1, true);                           // elas.cpp:421
  filter_redundant(d_can, 5, 1, false);                          // elas.cpp:422
  su_.clear(); sv_.clear(); sd_.clear(); sx_.clear();
  for (int uc = 1; uc < cw; uc++)                                // elas.cpp:425-431 (u-major order)
    for (int vc = 1; vc < ch; vc++) {
      const int d = d_can[vc * cw + uc];
      if (d < 0) continue;
      su_.push_back(uc * step); sv_.push_back(vc * step); sd_.push_back(d); sx_.push_back(uc * step - d);
    }
  memset(info, 0, sizeof(*info));
  info->nsup = (int32_t)su_.size();
  if (su_.size() < 3) { info->ok = 0; return; }                  // elas.cpp:66-71
  const size_t grid_bytes = (size_t)hp_.gw * hp_.gh * kGridWords * sizeof(uint32_t);
  size_t off = 0;
  for (int s = 0; s < 2; s++) {
    info->grid_offset[s] = (int64_t)off;
    make_grid(s, reinterpret_cast<uint32_t*>(payload + off));
    off += grid_bytes;
  }
  for (int s = 0; s < 2; s++) {
    info->tri_offset[s] = (int64_t)off;
    info->ntri[s] = make_side(s, reinterpret_cast<TriRec*>(payload + off));
    off += (size_t)info->ntri[s] * sizeof(TriRec);
  }
  info->ok = 1;
}

}  // namespace jnav
