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


JPEG_FUZZ = r'''
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "jpeg_host.h"
static uint32_t rng = 2463534242u;
static inline uint32_t xr() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return rng; }
int main(int argc, char** argv) {
  int counts[8] = {0};
  for (int f = 1; f < argc; f++) {
    FILE* fp = fopen(argv[f], "rb"); if (!fp) return 2;
    std::vector<uint8_t> good(1 << 20); good.resize(fread(good.data(), 1, good.size(), fp)); fclose(fp);
    for (int it = 0; it < 1500; it++) {
      // exactly-sized heap copy: any read past the end is an ASan report
      std::vector<uint8_t> b = good;
      switch (it % 6) {
        case 0: for (uint32_t k = 0, n = 1 + xr() % 8; k < n; k++) b[2 + xr() % (b.size() < 700 ? b.size() - 2 : 698)] = (uint8_t)xr(); break;
        case 1: for (uint32_t k = 0, n = 1 + xr() % 30; k < n; k++) b[2 + xr() % (b.size() - 2)] = (uint8_t)xr(); break;
        case 2: b.resize(2 + xr() % (b.size() - 2)); break;
        case 3: case 4: {                                                     // a random DHT / DQT / SOF / SOS / DRI body after SOI
          static const uint8_t ms[5] = {0xC4, 0xDB, 0xC0, 0xDA, 0xDD};
          std::vector<uint8_t> seg; const uint32_t len = xr() % 300;
          seg.push_back(0xFF); seg.push_back(it % 6 == 4 ? 0xC4 : ms[xr() % 5]); seg.push_back((uint8_t)((len + 2) >> 8)); seg.push_back((uint8_t)(len + 2));
          for (uint32_t k = 0; k < len; k++) seg.push_back((uint8_t)(it % 6 == 4 && k == 0 ? xr() % 2 * 16 + xr() % 4 : xr()));
          b.insert(b.begin() + 2, seg.begin(), seg.end());
        } break;
        default: {                                                           // SOF dimensions
          for (size_t i = 2; i + 9 < b.size(); i++) if (b[i] == 0xFF && (b[i + 1] == 0xC0)) { b[i + 5] = (uint8_t)xr(); b[i + 6] = (uint8_t)xr(); b[i + 7] = (uint8_t)xr(); b[i + 8] = (uint8_t)xr(); break; }
        }
      }
      uint8_t* exact = (uint8_t*)malloc(b.size()); memcpy(exact, b.data(), b.size());
      jnav::JpegFrame fr; std::vector<int16_t> coef;
      const jn_status st = jnav::jpeg_parse_and_decode(exact, b.size(), fr, coef);
      int w, h; jn_jpeg_info(exact, (int64_t)b.size(), &w, &h);
      free(exact);
      if (st == JN_OK && (coef.size() != (size_t)fr.bw * fr.bh * 64 || fr.width > jnav::kJpegMaxDim || fr.height > jnav::kJpegMaxDim)) return 3;
      counts[st & 7]++;
    }
  }
  printf("jpeg fuzz ok: ok %d unsupported %d invalid %d\n", counts[JN_OK], counts[JN_ERR_UNSUPPORTED], counts[JN_ERR_INVALID]);
  return 0;
}
'''


@pytest.mark.timeout(900)
def test_jpeg_host_decoder_fuzzed_under_asan_ubsan(tmp_path):
    """the entropy decoder takes compressed camera frames from the network.  4500 mutated files (header and
    scan byte flips, truncations, random DHT/DQT/SOF/SOS/DRI bodies, random frame sizes) through jpeg_host.cpp built with
    ASan + UBSan, each from an exactly-sized heap block."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz"))
    files = []
    for name in ("ragged_35x21_q95_422", "q90_422", "restart_blocks_q80_444"):
        f = tmp_path / (name + ".jpg")
        f.write_bytes(bytes(z[name + "__jpeg"]))
        files.append(str(f))
    src = tmp_path / "jpeg_fuzz.cpp"
    src.write_text(JPEG_FUZZ)
    exe = tmp_path / "jpeg_fuzz"
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I", CSRC,
                    str(src), os.path.join(CSRC, "jpeg_host.cpp"), "-o", str(exe)], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([str(exe), *files], capture_output=True, text=True, env=env, timeout=800)
    assert out.returncode == 0 and "jpeg fuzz ok" in out.stdout, (out.returncode, out.stdout[-500:], out.stderr[-4000:])
