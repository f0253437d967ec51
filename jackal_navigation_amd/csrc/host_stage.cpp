// host_stage.cpp — see host_stage.h.
#include "host_stage.h"
#include <immintrin.h>
#include <cstdlib>
#include <cstring>
#include <algorithm>

namespace jnav {

HostWorker::HostWorker(const HostParams& hp) : hp_(hp) {}

void HostWorker::corner_points(int W, int H, int n, std::vector<int32_t>& u, std::vector<int32_t>& v, std::vector<int32_t>& d) {
  int bu[6] = {0, 0, W - 1, W - 1, 0, 0}, bv[6] = {0, H - 1, 0, H - 1, 0, 0}, bd[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    int best = 10000000;
    for (int j = 0; j < n; j++) {
      const int du = bu[i] - u[j], dv = bv[i] - v[j], dist = du * du + dv * dv;
      if (dist < best) { best = dist; bd[i] = d[j]; }
    }
  }
  bu[4] = bu[2] + bd[2]; bv[4] = bv[2]; bd[4] = bd[2];            // :258-259: right-image twins of the two right-hand corners
  bu[5] = bu[3] + bd[3]; bv[5] = bv[3]; bd[5] = bd[3];
  u.resize(n + 6); v.resize(n + 6); d.resize(n + 6);
  for (int i = 0; i < 6; i++) { u[n + i] = bu[i]; v[n + i] = bv[i]; d[n + i] = bd[i]; }
}

size_t HostWorker::payload_capacity(const HostParams& hp) {
  const size_t maxsup = (size_t)hp.cw * hp.ch + kCornerPoints;
  return maxsup * 3 * sizeof(int32_t) + 2 * (2 * maxsup + 8) * 3 * sizeof(int32_t) + 256;
}

// elas.cpp:153-179.  Column-major sweep over the lattice; a point survives if at least
// incon_min_support lattice points (itself included) in the (2w+1)^2 window agree within
// incon_threshold.  Deletions take effect immediately, so the sweep order is part of the result.
// The window rows are contiguous int16: each row is one masked 16-lane compare (AVX2).
void HostWorker::filter_inconsistent(int16_t* D) const {
  const int cw = hp_.cw, ch = hp_.ch, win = hp_.incon_window_size, tol = hp_.incon_threshold;
  const bool simd = (2 * win + 1) <= 16;
  const __m256i vtol = _mm256_set1_epi16((short)tol), vneg = _mm256_set1_epi16(-1);
  for (int u = 0; u < cw; u++) {
    const int u0 = std::max(u - win, 0), u1 = std::min(u + win, cw - 1);
    const int span = u1 - u0 + 1;
    for (int v = 0; v < ch; v++) {
      const int d = D[v * cw + u];
      if (d < 0) continue;
      const int v0 = std::max(v - win, 0), v1 = std::min(v + win, ch - 1);
      int agree = 0;
      if (simd && u0 + 16 <= cw) {
        // Only "at least incon_min_support or not" matters, so the rows are visited from the point's own row outwards
        // and the scan stops as soon as the threshold is reached — after one or two rows wherever the lattice is
        // dense; only the sparse points that end up deleted see all 2w+1 rows.
        const __m256i vd = _mm256_set1_epi16((short)d);
        const unsigned span_bits = span >= 16 ? 0xFFFFFFFFu : ((1u << (2 * span)) - 1u);   // two mask bits per int16 lane
        const int need2 = 2 * hp_.incon_min_support;
        int agree2 = 0;
        for (int k = 0; k <= win && agree2 < need2; k++)
          for (int sgn = 0; sgn < (k ? 2 : 1); sgn++) {
            const int vv = sgn ? v + k : v - k;
            if (vv < v0 || vv > v1) continue;
            const __m256i e = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(D + vv * cw + u0));
            const __m256i diff = _mm256_abs_epi16(_mm256_sub_epi16(vd, e));
            const __m256i ok = _mm256_andnot_si256(_mm256_cmpgt_epi16(diff, vtol), _mm256_cmpgt_epi16(e, vneg));
            agree2 += __builtin_popcount((unsigned)_mm256_movemask_epi8(ok) & span_bits);
          }
        agree = agree2 >> 1;
      } else {
        for (int vv = v0; vv <= v1; vv++) {
          const int16_t* row = D + vv * cw;
          for (int uu = u0; uu <= u1; uu++) {
            const int e = row[uu];
            agree += (e >= 0) & (std::abs(d - e) <= tol);
          }
        }
      }
      if (agree < hp_.incon_min_support) D[v * cw + u] = -1;
    }
  }
}

// elas.cpp:181-235.  A point is redundant when, walking up to max_dist lattice steps in BOTH
// directions along one axis, a point within `thresh` disparity is met.  The sweep is in place: steps
// "before" the point see this pass's deletions, steps "after" it do not.
// Lines across the axis are independent, so `rows x cols` is walked row by row, 16 columns per AVX2 op
// (the vertical pass as is; the horizontal pass on a transposed copy).
static void redundant_down_rows(int16_t* D, int rows, int cols, int max_dist, int thresh) {
  const __m256i vneg = _mm256_set1_epi16(-1), vth = _mm256_set1_epi16((short)thresh);
  for (int r = 0; r < rows; r++) {
    int16_t* row = D + (size_t)r * cols;
    int c = 0;
    for (; c + 16 <= cols; c += 16) {
      const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(row + c));
      __m256i before = _mm256_setzero_si256(), after = before;
      for (int k = 1; k <= max_dist; k++) {
        if (r - k >= 0) {
          const __m256i e = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(row + c - (ptrdiff_t)k * cols));
          before = _mm256_or_si256(before, _mm256_andnot_si256(_mm256_cmpgt_epi16(_mm256_abs_epi16(_mm256_sub_epi16(d, e)), vth),
                                                               _mm256_cmpgt_epi16(e, vneg)));
        }
        if (r + k < rows) {
          const __m256i e = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(row + c + (ptrdiff_t)k * cols));
          after = _mm256_or_si256(after, _mm256_andnot_si256(_mm256_cmpgt_epi16(_mm256_abs_epi16(_mm256_sub_epi16(d, e)), vth),
                                                             _mm256_cmpgt_epi16(e, vneg)));
        }
      }
      const __m256i kill = _mm256_and_si256(_mm256_and_si256(before, after), _mm256_cmpgt_epi16(d, vneg));
      _mm256_storeu_si256(reinterpret_cast<__m256i*>(row + c), _mm256_blendv_epi8(d, vneg, kill));
    }
    for (; c < cols; c++) {                                  // columns past the last full vector
      const int d = row[c];
      if (d < 0) continue;
      bool before = false, after = false;
      for (int k = 1; k <= max_dist; k++) {
        if (r - k >= 0) { const int e = row[c - (ptrdiff_t)k * cols]; before |= e >= 0 && std::abs(d - e) <= thresh; }
        if (r + k < rows) { const int e = row[c + (ptrdiff_t)k * cols]; after |= e >= 0 && std::abs(d - e) <= thresh; }
      }
      if (before && after) row[c] = -1;
    }
  }
}
static void transpose16(const int16_t* src, int rows, int cols, int16_t* dst) {   // dst [cols][rows]
  constexpr int B = 32;
  for (int r0 = 0; r0 < rows; r0 += B)
    for (int c0 = 0; c0 < cols; c0 += B) {
      const int r1 = std::min(r0 + B, rows), c1 = std::min(c0 + B, cols);
      for (int r = r0; r < r1; r++)
        for (int c = c0; c < c1; c++) dst[(size_t)c * rows + r] = src[(size_t)r * cols + c];
    }
}
void HostWorker::filter_redundant(int16_t* D, int max_dist, int thresh, bool vertical) const {
  const int cw = hp_.cw, ch = hp_.ch;
  if (vertical) { redundant_down_rows(D, ch, cw, max_dist, thresh); return; }
  tr_.resize((size_t)cw * ch);
  transpose16(D, ch, cw, tr_.data());                        // lattice columns become rows
  redundant_down_rows(tr_.data(), cw, ch, max_dist, thresh);
  transpose16(tr_.data(), cw, ch, D);
}

void HostWorker::filter_and_list(int16_t* d_can, FrameInfo* info, FrameScratch* fs, bool filtered) const {
  const int cw = hp_.cw, ch = hp_.ch, step = hp_.step;
  if (!filtered) {
    filter_inconsistent(d_can);                                  // elas.cpp:416
    filter_redundant(d_can, 5, 1, true);                         // elas.cpp:421
    filter_redundant(d_can, 5, 1, false);                        // elas.cpp:422
  }
  fs->u.clear(); fs->v.clear(); fs->d.clear(); fs->x.clear();
  for (int uc = 1; uc < cw; uc++)                                // elas.cpp:425-431 (u-major order)
    for (int vc = 1; vc < ch; vc++) {
      const int d = d_can[vc * cw + uc];
      if (d < 0) continue;
      const int u = uc * step, v = vc * step;
      fs->u.push_back(u); fs->v.push_back(v); fs->d.push_back(d); fs->x.push_back(u - d);
    }
  if (hp_.add_corners) {                                         // elas.cpp:435
    const int n = (int)fs->u.size();
    corner_points(hp_.W, hp_.H, n, fs->u, fs->v, fs->d);
    fs->x.resize(n + kCornerPoints);
    for (int i = n; i < n + kCornerPoints; i++) fs->x[i] = fs->u[i] - fs->d[i];
  }
  memset(info, 0, sizeof(*info));
  info->nsup = (int32_t)fs->u.size();
  info->ok = fs->u.size() >= 3;                                  // elas.cpp:66-71
}

size_t HostWorker::place(FrameInfo* info, size_t base) {
  if (!info->ok) return 0;
  const size_t n = (size_t)info->nsup;
  const size_t sup_bytes = n * 3 * sizeof(int32_t), side_bytes = (2 * n + 8) * 3 * sizeof(int32_t);   // <= 2n-5 triangles per side
  info->sup_offset = (int64_t)base;
  info->corner_offset[0] = (int64_t)(base + sup_bytes);
  info->corner_offset[1] = (int64_t)(base + sup_bytes + side_bytes);
  return (sup_bytes + 2 * side_bytes + 255) / 256 * 256;
}

void HostWorker::triangulate_side(int side, const FrameScratch& fs, uint8_t* payload, FrameInfo* info) {
  if (!info->ok) return;
  const int n = (int)fs.u.size();
  if (side == 0) {
    int32_t* uvd = reinterpret_cast<int32_t*>(payload + info->sup_offset);
    for (int i = 0; i < n; i++) { uvd[3 * i] = fs.u[i]; uvd[3 * i + 1] = fs.v[i]; uvd[3 * i + 2] = fs.d[i]; }
  }
  int32_t* corners = reinterpret_cast<int32_t*>(payload + info->corner_offset[side]);
  const int nt = dt_.run(side ? fs.x.data() : fs.u.data(), fs.v.data(), n, corners);
  info->ntri[side] = nt < 0 ? 0 : nt;
}

// Support points of one frame from the GPU's (uc, vc, d) lattice triples: pixel coordinates, plus the corner points when
// the preset asks for them.  info->nsup counts both (jn_api.cpp adds kCornerPoints to the GPU's count).
static void points_from_list(const HostParams& hp, const int16_t* t, int nsup, std::vector<int32_t>& u, std::vector<int32_t>& v, std::vector<int32_t>& d) {
  const int nlist = hp.add_corners ? nsup - HostWorker::kCornerPoints : nsup, step = hp.step;
  u.resize(nlist); v.resize(nlist); d.resize(nlist);
  for (int i = 0; i < nlist; i++) { u[i] = t[3 * i] * step; v[i] = t[3 * i + 1] * step; d[i] = t[3 * i + 2]; }
  if (hp.add_corners) HostWorker::corner_points(hp.W, hp.H, nlist, u, v, d);
}
static void write_support(uint8_t* payload, const FrameInfo* info, const std::vector<int32_t>& u, const std::vector<int32_t>& v, const std::vector<int32_t>& d) {
  int32_t* uvd = reinterpret_cast<int32_t*>(payload + info->sup_offset);
  for (int i = 0; i < info->nsup; i++) { uvd[3 * i] = u[i]; uvd[3 * i + 1] = v[i]; uvd[3 * i + 2] = d[i]; }
}

void HostWorker::triangulate_side_from_list(int side, const int16_t* t, uint8_t* payload, FrameInfo* info, const uint16_t* arrangement) {
  if (!info->ok) return;
  const int n = info->nsup;
  points_from_list(hp_, t, n, xs_, ys_, ds_);
  if (side == 0) write_support(payload, info, xs_, ys_, ds_);
  else for (int i = 0; i < n; i++) xs_[i] -= ds_[i];                       // (u - d, v), elas.cpp:466-467
  int32_t* corners = reinterpret_cast<int32_t*>(payload + info->corner_offset[side]);
  const int nt = arrangement ? dt_.run_arranged(xs_.data(), ys_.data(), n, arrangement, corners) : dt_.run(xs_.data(), ys_.data(), n, corners);
  info->ntri[side] = nt < 0 ? 0 : nt;
}

void HostWorker::side_prepare(int side, const int16_t* t, uint8_t* payload, const FrameInfo* info, SideState* st, int want_parts) const {
  st->parts = 0;
  if (!info->ok) return;
  const int n = info->nsup;
  std::vector<int32_t> d;
  points_from_list(hp_, t, n, st->xs, st->ys, d);
  if (side == 0) write_support(payload, info, st->xs, st->ys, d);
  else for (int i = 0; i < n; i++) st->xs[i] -= d[i];                      // (u - d, v), elas.cpp:466-467
  st->parts = st->dt.prepare(st->xs.data(), st->ys.data(), n, want_parts);
}

void HostWorker::side_finish(int side, uint8_t* payload, FrameInfo* info, SideState* st) {
  if (!info->ok) return;
  int32_t* corners = reinterpret_cast<int32_t*>(payload + info->corner_offset[side]);
  const int nt = st->parts ? st->dt.finish(corners) : -1;
  info->ntri[side] = nt < 0 ? 0 : nt;
}

}  // namespace jnav
