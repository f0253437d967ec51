"""ThreadSanitizer over the host-stage task pool (csrc/pool.h): several slot threads submit frame
tasks and frame-side tasks concurrently, exactly as jn_api.cpp's slot workers do; results must equal
the serial ones.  CPU only (no HIP in pool.h / host_stage / delaunay)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "jackal_navigation_amd", "csrc")

DRIVER = r'''
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
#include "pool.h"
using namespace jnav;
int main() {
  HostParams hp; hp.W = 400; hp.H = 300; hp.disp_max = 63; hp.step = 5; hp.incon_window_size = 5; hp.incon_threshold = 5;
  hp.incon_min_support = 5; hp.grid_size = 20; hp.gw = 20; hp.gh = 15; hp.cw = 80; hp.ch = 60;
  const int frames = 6, slots = 3, rounds = 4;
  // random candidate lattices (smooth disparity + holes), one set per slot
  std::vector<std::vector<int16_t>> can(slots * frames, std::vector<int16_t>((size_t)hp.cw * hp.ch));
  std::mt19937 g(7);
  for (auto& c : can)
    for (int v = 0; v < hp.ch; v++)
      for (int u = 0; u < hp.cw; u++) c[v * hp.cw + u] = (u == 0 || v == 0) ? 0 : (g() % 10 < 2 ? -1 : (int16_t)(10 + v / 3 + (int)(g() % 3)));
  const size_t cap = HostWorker::payload_capacity(hp);
  // serial reference
  std::vector<std::vector<uint8_t>> ref_payload(slots * frames, std::vector<uint8_t>(cap));
  std::vector<FrameInfo> ref_info(slots * frames);
  {
    HostWorker w(hp);
    for (int i = 0; i < slots * frames; i++) {
      std::vector<int16_t> c = can[i]; FrameScratch fs;
      w.filter_and_list(c.data(), &ref_info[i], &fs);
      HostWorker::place(&ref_info[i], 0);
      w.triangulate_side(0, fs, ref_payload[i].data(), &ref_info[i]);
      w.triangulate_side(1, fs, ref_payload[i].data(), &ref_info[i]);
    }
  }
  Pool pool(5, hp, POOL_SPIN_US);                            // 0: workers sleep at once (batch handles); > 0: they and run() poll first (latency handles)
  int bad = 0;
  std::vector<std::thread> th;
  std::vector<int> bad_per(slots, 0);
  for (int s = 0; s < slots; s++)
    th.emplace_back([&, s] {
      std::vector<FrameScratch> scratch(frames);
      std::vector<FrameInfo> info(frames);
      std::vector<std::vector<uint8_t>> payload(frames, std::vector<uint8_t>(cap));
      for (int r = 0; r < rounds; r++) {
        std::vector<std::vector<int16_t>> c(frames);
        for (int i = 0; i < frames; i++) c[i] = can[s * frames + i];
        pool.run(frames, [&](HostWorker& w, int i) { w.filter_and_list(c[i].data(), &info[i], &scratch[i]); });
        for (int i = 0; i < frames; i++) HostWorker::place(&info[i], 0);
        pool.run(2 * frames, [&](HostWorker& w, int k) { w.triangulate_side(k & 1, scratch[k >> 1], payload[k >> 1].data(), &info[k >> 1]); });
        for (int i = 0; i < frames; i++) {
          const FrameInfo& a = info[i]; const FrameInfo& b = ref_info[s * frames + i];
          if (a.ok != b.ok || a.nsup != b.nsup || a.ntri[0] != b.ntri[0] || a.ntri[1] != b.ntri[1]) { bad_per[s]++; continue; }
          const size_t used = a.ok ? (size_t)a.corner_offset[1] + (size_t)a.ntri[1] * 12 : 0;
          // compare only the regions that carry data (the gap between the two corner arrays is unspecified)
          if (a.ok && (memcmp(payload[i].data(), ref_payload[s * frames + i].data(), (size_t)a.corner_offset[0] + (size_t)a.ntri[0] * 12) != 0 ||
                       memcmp(payload[i].data() + a.corner_offset[1], ref_payload[s * frames + i].data() + a.corner_offset[1], (size_t)a.ntri[1] * 12) != 0))
            bad_per[s]++;
          (void)used;
        }
      }
    });
  for (auto& t : th) t.join();
  for (int s = 0; s < slots; s++) bad += bad_per[s];
  printf("pool run: %d mismatches\n", bad);
  return bad ? 1 : 0;
}
'''


@pytest.mark.timeout(600)
@pytest.mark.parametrize("spin_us", [0, 200])
def test_pool_under_tsan(tmp_path, spin_us):
    src = tmp_path / "tsan_driver.cpp"
    src.write_text(DRIVER)
    exe = tmp_path / "tsan_driver"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-mavx2", "-ffp-contract=off", "-fsanitize=thread", "-DPOOL_SPIN_US=%d" % spin_us, "-I", CSRC, str(src),
           os.path.join(CSRC, "host_stage.cpp"), os.path.join(CSRC, "delaunay.cpp"), "-o", str(exe), "-lpthread"]
    subprocess.run(cmd, check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-4000:])
    assert "0 mismatches" in out.stdout and "ThreadSanitizer" not in out.stderr
