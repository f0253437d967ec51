"""Generates tests/golden/bm_hashes.txt: FNV-1a-64 (over the int16 map viewed as uint32 words) of the block-matching ORACLE's
output on Appendix-A pairs.  SELF-REFERENTIAL goldens (the reference has no block matcher): they pin the HIP path and
bench.py's self-check to oracle/bm_oracle.cpp, nothing more.  Run from the repo root:  python tests/golden/make_bm_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Oracle, BmOracle  # noqa: E402

o, s = Oracle(), BmOracle()
rows = ["# W H scene_disp D block_radius subpixel seed fnv1a64(int16 disparity map as u32 words) -- oracle/bm_oracle.cpp, default cap/lr"]
for (W, H, scene, D, r, sub) in ((640, 480, 64, 64, 4, 0), (640, 480, 64, 64, 4, 1), (1280, 720, 128, 128, 4, 0), (1280, 720, 128, 128, 4, 1)):
    L, R = o.synth_pair(W, H, scene, 12345)
    d = s.process(s.params(D, r, subpixel=sub), L, R)
    rows.append("%d %d %d %d %d %d 12345 %016x" % (W, H, scene, D, r, sub, o.fnv(d.view(np.uint32))))
    print(rows[-1])
open(os.path.join(ROOT, "tests", "golden", "bm_hashes.txt"), "w").write("\n".join(rows) + "\n")
