// oracle/bm_oracle.cpp — scalar CPU definition of the block-matching mode (TEST INFRASTRUCTURE ONLY).
//
// PARITY UNPINNED / SELF-REFERENTIAL: the reference repository has no block matcher (its only matcher is libelas,
// SURVEY.md 0.1), so there is no reference code, test or golden vector to pin this against.  This file IS the definition
// the HIP kernels (jackal_navigation_amd/csrc/bm.hip) are compared with; it follows include/jn_bm.h line by line.  What
// it can be checked against, and is in tests/test_bm_oracle.py: a literal five-loop evaluation of the cost on small
// images, the synthetic scenes' ground-truth disparities, and hand-computable cases.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {

typedef struct orc_bm_params { int32_t num_disparities, block_radius, prefilter_cap, lr_max_diff, subpixel, cost_function; } orc_bm_params;   // cost_function 1: squared differences (jn_bm.h JN_BM_COST_SSD)

void orc_sgm_prefilter(const uint8_t* I, int32_t W, int32_t H, int32_t cap, uint8_t* g);     // sgm_oracle.cpp: the same prefilter

// The cost by its definition: five nested loops.  side 0: CL(x,y,d), side 1: CR(x,y,d).
static int32_t bm_cost_fn(const uint8_t* gL, const uint8_t* gR, int32_t W, int32_t H, int32_t r, int32_t side, int32_t x, int32_t y, int32_t d, int32_t squared) {
  auto cl = [&](int v) { return std::min(std::max(v, 0), W - 1); };
  auto cr = [&](int v) { return std::min(std::max(v, 0), H - 1); };
  int c = 0;
  for (int j = -r; j <= r; j++)
    for (int i = -r; i <= r; i++) {
      const size_t row = (size_t)cr(y + j) * W;
      const int e = side == 0 ? (int)gL[row + cl(x + i)] - (int)gR[row + cl(x + i - d)] : (int)gR[row + cl(x + i)] - (int)gL[row + cl(x + i + d)];
      c += squared ? e * e : std::abs(e);
    }
  return c;
}
int32_t orc_bm_cost(const uint8_t* gL, const uint8_t* gR, int32_t W, int32_t H, int32_t r, int32_t side, int32_t x, int32_t y, int32_t d) {
  return bm_cost_fn(gL, gR, W, H, r, side, x, y, d, 0);
}
int32_t orc_bm_cost_ssd(const uint8_t* gL, const uint8_t* gR, int32_t W, int32_t H, int32_t r, int32_t side, int32_t x, int32_t y, int32_t d) {
  return bm_cost_fn(gL, gR, W, H, r, side, x, y, d, 1);
}

// All costs of one side, [H][W][D] u16, by separable sums of the absolute-difference image of every d (the same numbers
// as orc_bm_cost, tests compare the two): AD_d(x,y) = |a(x) - b(x -/+ d)| needs the clamp INSIDE the window, so the
// horizontal sum runs over clamped x+i for a and clamped x+i-/+d for b, not over a clamped AD image.
static void bm_costs(const uint8_t* gL, const uint8_t* gR, int W, int H, int D, int r, int side, int squared, std::vector<uint32_t>& out) {
  out.assign((size_t)W * H * D, 0);
  auto cl = [&](int v) { return std::min(std::max(v, 0), W - 1); };
  std::vector<int> hrow((size_t)W * H);
  for (int d = 0; d < D; d++) {
    for (int y = 0; y < H; y++) {
      const uint8_t* a = (side == 0 ? gL : gR) + (size_t)y * W;
      const uint8_t* b = (side == 0 ? gR : gL) + (size_t)y * W;
      const int s = side == 0 ? -d : d;
      for (int x = 0; x < W; x++) {
        int c = 0;
        for (int i = -r; i <= r; i++) { const int e = (int)a[cl(x + i)] - (int)b[cl(x + i + s)]; c += squared ? e * e : std::abs(e); }
        hrow[(size_t)y * W + x] = c;
      }
    }
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        int c = 0;
        for (int j = -r; j <= r; j++) c += hrow[(size_t)std::min(std::max(y + j, 0), H - 1) * W + x];
        out[((size_t)y * W + x) * D + d] = (uint32_t)c;
      }
  }
}

// Whole mode.  disp [H][W] int16.  Returns 0, or -1 for parameters outside the definition.
int32_t orc_bm_process(const orc_bm_params* p, const uint8_t* L, const uint8_t* R, int32_t W, int32_t H, int16_t* disp) {
  const int D = p->num_disparities, r = p->block_radius;
  if (D < 1 || D > 256 || r < 1 || r > 7 || p->prefilter_cap < 1 || p->prefilter_cap > 31) return -1;
  const size_t px = (size_t)W * H;
  std::vector<uint8_t> gL(px), gR(px);
  orc_sgm_prefilter(L, W, H, p->prefilter_cap, gL.data());
  orc_sgm_prefilter(R, W, H, p->prefilter_cap, gR.data());
  if (p->cost_function != 0 && p->cost_function != 1) return -1;
  std::vector<uint32_t> CL, CR;
  bm_costs(gL.data(), gR.data(), W, H, D, r, 0, p->cost_function, CL);
  bm_costs(gL.data(), gR.data(), W, H, D, r, 1, p->cost_function, CR);
  auto argmin = [&](const uint32_t* c) { int best = 0; for (int d = 1; d < D; d++) if (c[d] < c[best]) best = d; return best; };   // smallest d attaining the minimum
  std::vector<int> dR(px);
  for (size_t i = 0; i < px; i++) dR[i] = argmin(&CR[i * D]);
  const int scale = p->subpixel ? 16 : 1;
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const uint32_t* c = &CL[((size_t)y * W + x) * D];
      const int d = argmin(c);
      bool ok = true;
      if (p->lr_max_diff >= 0) ok = x - d >= 0 && std::abs(d - dR[(size_t)y * W + x - d]) <= p->lr_max_diff;
      int out = -scale;
      if (ok) {
        out = d * scale;
        if (p->subpixel && d > 0 && d < D - 1) {
          const int den = std::max((int)c[d - 1] + (int)c[d + 1] - 2 * (int)c[d], 1);
          out = 16 * d + (16 * ((int)c[d - 1] - (int)c[d + 1]) + den) / (2 * den);
        }
      }
      disp[(size_t)y * W + x] = (int16_t)out;
    }
  return 0;
}

}  // extern "C"
