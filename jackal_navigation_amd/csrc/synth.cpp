// synth.cpp — synthetic rectified stereo pair used by bench.py and the GPU tests (product utility).
// Generator specified in SURVEY.md Appendix A: 4x4-block random texture, disparity ramp from the
// horizon down plus a fronto-parallel box; xorshift32 stream seeded per frame.
#include "../../include/jn_stereo.h"
#include <vector>

extern "C" void jn_synth_pair(int32_t W, int32_t H, int32_t scene_disp, uint32_t seed, uint8_t* L, uint8_t* R) {
  uint32_t s = seed;
  auto rnd = [&s]() -> uint32_t { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
  const int tw = W + 512, bw = tw / 4 + 1, bh = H / 4 + 1;
  std::vector<uint8_t> coarse((size_t)bw * bh);
  for (size_t i = 0; i < coarse.size(); i++) coarse[i] = (uint8_t)(rnd() & 255u);
  std::vector<uint8_t> tex((size_t)tw * H);
  for (int y = 0; y < H; y++) {
    const uint8_t* c = &coarse[(size_t)(y >> 2) * bw];
    uint8_t* t = &tex[(size_t)y * tw];
    for (int x = 0; x < tw; x++) {
      const int val = (c[x >> 2] * 3 + (int)(rnd() & 63u)) / 4 + 16;
      t[x] = (uint8_t)(val > 255 ? 255 : val);
    }
  }
  const int box_d = (int)(scene_disp * 0.7);
  for (int y = 0; y < H; y++) {
    const uint8_t* t = &tex[(size_t)y * tw] + 256;
    const int ramp = (int)((double)y / H * (scene_disp * 0.6)) + 2;
    const bool box_row = y > H / 3 && y < 2 * H / 3;
    for (int x = 0; x < W; x++) {
      const int d = (box_row && x > W / 3 && x < W / 2) ? box_d : ramp;
      R[(size_t)y * W + x] = t[x];
      L[(size_t)y * W + x] = t[x - d];
    }
  }
}
