// Phase timing of the product triangulation (csrc/delaunay.cpp) on the support points of one 1280x720 frame side:
//   python3 scripts/probes/delaunay_phases.py     (writes the points with the checker, builds this file with g++, runs it)
#include <chrono>
#include <algorithm>
#include <cstdio>
#include <vector>
#include <cstdint>
#define private public
#include "delaunay.h"
using namespace jnav;
int main(int argc, char** argv) {
  FILE* f = fopen(argc > 1 ? argv[1] : "/tmp/sup.txt", "r"); std::vector<int32_t> x, y; int a, b, c;
  while (fscanf(f, "%d %d %d", &a, &b, &c) == 3) { x.push_back(a); y.push_back(b); }
  int n = x.size(); std::vector<int32_t> tri(6 * n + 16);
  Delaunay d;
  auto now = [] { return std::chrono::steady_clock::now(); };
  double tp = 1e9, ts = 1e9, tc = 1e9, tf = 1e9; int N = 1500;
  for (int it = 0; it < N + 20; it++) {
    auto t0 = now();
    d.prepare(x.data(), y.data(), n, 1);
    auto t1 = now();
    Delaunay::Part& p = d.part_[0];
    d.split(p.lo, p.hi, p.axis);
    auto t2 = now();
    Delaunay::Ctx cx{p.slot0};
    d.conquer(d.order_.data() + p.lo, p.hi - p.lo, p.axis, p.farleft, p.farright, cx);
    p.used = cx.next - p.slot0;
    auto t3 = now();
    int nt = d.finish(tri.data());
    auto t4 = now();
    if (it >= 20) { tp = std::min(tp, std::chrono::duration<double, std::micro>(t1 - t0).count()); ts = std::min(ts, std::chrono::duration<double, std::micro>(t2 - t1).count());
      tc = std::min(tc, std::chrono::duration<double, std::micro>(t3 - t2).count()); tf = std::min(tf, std::chrono::duration<double, std::micro>(t4 - t3).count()); }
    if (it == 0) printf("n %d tri %d\n", n, nt);
  }
  printf("MIN: prepare (sort + arrange) %.1f us, split %.1f us, conquer %.1f us, output %.1f us, total %.1f us\n", tp, ts, tc, tf, tp + ts + tc + tf);
}
