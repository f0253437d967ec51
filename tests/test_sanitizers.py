"""AddressSanitizer + UBSan over the CPU-side C++ (host stage, Delaunay, and the oracle restatement).
GPU sanitizers are not available on the MI355X pool, so the memory-safety net is the host build."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "jackal_navigation_amd", "csrc")

DRIVER = r'''
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "host_stage.h"
#include "oracle.h"
using namespace jnav;
int main(int argc, char** argv) {
  const int W = atoi(argv[1]), H = atoi(argv[2]), sceneD = atoi(argv[3]), dmax = atoi(argv[4]);
  std::vector<uint8_t> L((size_t)W * H), R((size_t)W * H);
  orc_synth_pair(W, H, sceneD, 4242, L.data(), R.data());
  orc_params p; orc_params_default(&p, 0); p.disp_max = dmax;
  // oracle end to end under the sanitizers
  std::vector<float> D1((size_t)W * H, 0.f), D2((size_t)W * H, 0.f);
  if (orc_elas_process(&p, L.data(), R.data(), D1.data(), D2.data(), W, H, W) != 0) return 2;
  // product host stage on the oracle's candidate lattice
  std::vector<uint8_t> d1((size_t)16 * W * H), d2((size_t)16 * W * H);
  orc_descriptor(L.data(), W, H, W, d1.data()); orc_descriptor(R.data(), W, H, W, d2.data());
  int cw, ch; orc_candidates(&p, d1.data(), d2.data(), W, H, nullptr, &cw, &ch);
  std::vector<int16_t> can((size_t)cw * ch);
  orc_candidates(&p, d1.data(), d2.data(), W, H, can.data(), &cw, &ch);
  HostParams hp; hp.W = W; hp.H = H; hp.disp_max = dmax; hp.step = 5; hp.incon_window_size = 5; hp.incon_threshold = 5;
  hp.incon_min_support = 5; hp.grid_size = 20; hp.gw = (W + 19) / 20; hp.gh = (H + 19) / 20; hp.cw = cw; hp.ch = ch;
  HostWorker w(hp); FrameInfo fi; FrameScratch fs;
  std::vector<uint8_t> payload(HostWorker::payload_capacity(hp));
  w.filter_and_list(can.data(), &fi, &fs);
  HostWorker::place(&fi, 0);
  w.triangulate_side(0, fs, payload.data(), &fi);
  w.triangulate_side(1, fs, payload.data(), &fi);
  // the two Delaunay implementations must agree
  std::vector<int32_t> uvd((size_t)3 * fi.nsup + 3), c((size_t)6 * fi.nsup + 48); std::vector<float> pl((size_t)12 * fi.nsup + 96);
  const int n = orc_support(&p, d1.data(), d2.data(), W, H, uvd.data(), fi.nsup + 1);
  if (n != fi.nsup) return 3;
  for (int side = 0; side < 2; side++) {
    const int nt = orc_triangles(uvd.data(), n, side, c.data(), pl.data(), 2 * n + 16);
    if (nt != fi.ntri[side]) return 4;
    if (memcmp(c.data(), payload.data() + fi.corner_offset[side], (size_t)nt * 12) != 0) return 5;
  }
  printf("sanitized run ok: %d support points, %d/%d triangles\n", fi.nsup, fi.ntri[0], fi.ntri[1]);
  return 0;
}
'''


@pytest.mark.timeout(600)
def test_host_and_oracle_under_asan_ubsan(tmp_path):
    src = tmp_path / "san_driver.cpp"
    src.write_text(DRIVER)
    exe = tmp_path / "san_driver"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-mavx2", "-msse3", "-ffp-contract=off", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-I", CSRC, "-I", os.path.join(ROOT, "oracle"), str(src),
           os.path.join(CSRC, "host_stage.cpp"), os.path.join(CSRC, "delaunay.cpp"),
           os.path.join(ROOT, "oracle", "elas_oracle.cpp"), os.path.join(ROOT, "oracle", "delaunay_oracle.cpp"),
           os.path.join(ROOT, "oracle", "node_oracle.cpp"), os.path.join(ROOT, "oracle", "synth_oracle.cpp"), "-o", str(exe)]
    subprocess.run(cmd, check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for args in (("320", "180", "48", "255"), ("333", "201", "30", "95")):
        out = subprocess.run([str(exe), *args], capture_output=True, text=True, env=env, timeout=300)
        assert out.returncode == 0, (args, out.returncode, out.stdout[-2000:], out.stderr[-4000:])
        assert "sanitized run ok" in out.stdout
