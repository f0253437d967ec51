#!/usr/bin/env python3
"""Generates tests/golden/reference_middlebury_hashes.txt from the REAL reference (oracle/_ref/libelas_ref.so) with the
MIDDLEBURY preset (elas.h:118-145: add_corners, ipol_gap_width 5000, median filter, both sides post-processed).

The reference reads descriptor bytes it never initialises (descriptor.cpp:29; readers listed in the docstring of
oracle.binding.Reference); `Reference()` runs it in a worker whose allocations are zero-filled — what a freshly mapped buffer
holds anyway, and the definition the product and the oracle use.  Run in the dev container only:

    python tests/golden/make_middlebury_golden.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.binding import Oracle, Reference  # noqa: E402
from scenes import make_scene  # noqa: E402

o, r = Oracle(), Reference()
lines = []
for (W, H, sd, dmax, seed) in ((320, 180, 48, 255, 12345), (400, 300, 60, 127, 7), (640, 480, 64, 63, 12345), (1280, 720, 128, 127, 12345)):
    L, R = o.synth_pair(W, H, sd, seed)
    D1, D2 = r.process(r.params(1, disp_max=dmax), L, R)
    lines.append("synth %d %d %d %d %d %016x %016x" % (W, H, sd, dmax, seed, o.fnv(D1), o.fnv(D2)))
    print(lines[-1])
for kind in ("strips", "patches", "slanted", "blobs"):
    W, H, dmax, seed = 320, 240, 79, 21
    L, R = make_scene(kind, W, H, dmax, seed)
    D1, D2 = r.process(r.params(1, disp_max=dmax), L, R)
    lines.append("%s %d %d %d %d %d %016x %016x" % (kind, W, H, dmax, dmax, seed, o.fnv(D1), o.fnv(D2)))
    print(lines[-1])
with open(os.path.join(ROOT, "tests", "golden", "reference_middlebury_hashes.txt"), "w") as f:
    f.write("# kind W H scene_disp disp_max seed fnv1a64(D1) fnv1a64(D2) -- reference src/elas, MIDDLEBURY preset, D pre-filled 0,\n"
            "# uninitialised allocations zero-filled (oracle.binding.Reference worker); kind synth = Appendix-A generator, else tests/scenes.py\n")
    f.write("\n".join(lines) + "\n")
