"""Generates tests/golden/bm_ssd_hashes.txt: FNV-1a-64 (over the int16 map viewed as uint32 words) of the block-matching ORACLE's output
with cost_function = 1 (sum of squared differences, include/jn_bm.h JN_BM_COST_SSD) on Appendix-A pairs.  SELF-REFERENTIAL goldens (the
reference has no block matcher): they pin the matrix-core path (csrc/bm_mfma.hip) to oracle/bm_oracle.cpp, nothing more.
Run from the repo root:  python tests/golden/make_bm_ssd_golden.py      (the 1920x1080 D=256 case needs ~5 GB and a minute)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Oracle, BmOracle  # noqa: E402

o, s = Oracle(), BmOracle()
rows = ["# W H scene_disp D block_radius subpixel seed fnv1a64(int16 disparity map as u32 words) -- oracle/bm_oracle.cpp, cost_function 1 (SSD), default cap/lr"]
for (W, H, scene, D, r, sub) in ((640, 480, 64, 64, 4, 0), (640, 480, 64, 64, 4, 1), (1280, 720, 128, 128, 4, 0), (1920, 1080, 256, 256, 4, 1)):
    L, R = o.synth_pair(W, H, scene, 12345)
    d = s.process(s.params(D, r, subpixel=sub, cost_function=1), L, R)
    rows.append("%d %d %d %d %d %d 12345 %016x" % (W, H, scene, D, r, sub, o.fnv(d.view(np.uint32))))
    print(rows[-1], "valid %.3f" % (d >= 0).mean())
open(os.path.join(ROOT, "tests", "golden", "bm_ssd_hashes.txt"), "w").write("\n".join(rows) + "\n")
