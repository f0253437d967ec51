"""Generates tests/golden/sgm_hashes.txt: FNV-1a-64 (over the int16 map viewed as uint32 words) of the SGM ORACLE's output
on Appendix-A pairs.  SELF-REFERENTIAL goldens (the reference has no SGM): they pin the HIP path and bench.py's
self-check to oracle/sgm_oracle.cpp, nothing more.  Run from the repo root:  python tests/golden/make_sgm_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Oracle, SgmOracle  # noqa: E402

o, s = Oracle(), SgmOracle()
rows = ["# W H scene_disp D subpixel seed fnv1a64(int16 disparity map as u32 words) -- oracle/sgm_oracle.cpp, default P1/P2/cap/lr"]
for (W, H, scene, D, sub) in ((640, 480, 64, 64, 0), (1280, 720, 128, 128, 0), (1280, 720, 128, 128, 1)):
    L, R = o.synth_pair(W, H, scene, 12345)
    d = s.process(s.params(D, subpixel=sub), L, R)
    rows.append("%d %d %d %d %d 12345 %016x" % (W, H, scene, D, sub, o.fnv(d.view(np.uint32))))
    print(rows[-1])
open(os.path.join(ROOT, "tests", "golden", "sgm_hashes.txt"), "w").write("\n".join(rows) + "\n")
