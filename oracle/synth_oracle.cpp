// oracle/synth_oracle.cpp — TEST INFRASTRUCTURE.  Synthetic stereo pair of SURVEY.md Appendix A
// and the FNV hash used there; lets the tests re-derive the survey's known-answer hashes.
#include "oracle.h"
#include <vector>
#include <algorithm>

extern "C" void orc_synth_pair(int32_t W, int32_t H, int32_t sceneD, uint32_t seed, uint8_t* L, uint8_t* R) {
  uint32_t rng = seed;
  auto next = [&rng]() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return rng; };
  const int TW = W + 512, BW = TW / 4 + 1, BH = H / 4 + 1;
  std::vector<uint8_t> blk((size_t)BW * BH);
  for (auto& b : blk) b = (uint8_t)(next() & 255);
  std::vector<uint8_t> tex((size_t)TW * H);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < TW; x++) {
      int v = (blk[(size_t)(y / 4) * BW + x / 4] * 3 + (int)(next() & 63)) / 4 + 16;
      tex[(size_t)y * TW + x] = (uint8_t)std::min(v, 255);
    }
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      int d = (int)((double)y / H * (sceneD * 0.6)) + 2;
      if (x > W / 3 && x < W / 2 && y > H / 3 && y < 2 * H / 3) d = (int)(sceneD * 0.7);
      R[(size_t)y * W + x] = tex[(size_t)y * TW + x + 256];
      L[(size_t)y * W + x] = tex[(size_t)y * TW + x - d + 256];
    }
}

extern "C" uint64_t orc_fnv1a64_u32(const uint32_t* w, int64_t n) {
  uint64_t h = 1469598103934665603ull;
  for (int64_t i = 0; i < n; i++) h = (h ^ w[i]) * 1099511628211ull;
  return h;
}
