#!/usr/bin/env python3
"""Generates tests/golden/scan_golden.json: for the benchmark pairs (Appendix-A generator, seed 12345) the FNV-1a-64 of the u8 depth
map and the 90 scan bins + 4 extrema that the node's tail makes of the REFERENCE's D1 — compiled reference (oracle/_ref, zero-filled
allocations, see oracle.binding.Reference) -> convertTo(CV_8U) -> cacheDisparityValues LUT -> publishObstacleScan, the last three
as restated in oracle/node_oracle.cpp (OpenCV / ROS are not installed: those are definitions, point_cloud.cpp:422, :104-147,
:213-296).  bench.py checks the timed path's frame 0 against it.  Run in the dev container only:

    python tests/golden/make_scan_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Oracle, Reference  # noqa: E402

o, r = Oracle(), Reference()
out = {"_what": "W H scene_disp disp_max -> fnv1a64 of the u8 map (as u32 words), bins[90], meta[4] (angle_min, angle_max, range_min, range_max); seed 12345"}
for (W, H, sd, dmax) in ((320, 180, 48, 255), (640, 480, 64, 63), (1280, 720, 128, 127), (1920, 1080, 256, 255)):
    L, R = o.synth_pair(W, H, sd, 12345)
    D1, _ = r.process(r.params(0, disp_max=dmax), L, R)
    u8 = o.to_u8(D1)
    sp = o.scan_params(W, H)
    bins, meta, used = o.scan(sp, u8, o.valid_lut(sp, W, H))
    out["%d %d %d %d" % (W, H, sd, dmax)] = {"u8_fnv": "%016x" % o.fnv(np.ascontiguousarray(u8).view(np.uint32)), "bins": [float(b) for b in bins],
                                             "meta": [float(m) for m in meta], "pixels_used": int(used)}
    print(W, H, out["%d %d %d %d" % (W, H, sd, dmax)]["u8_fnv"], int((bins < 1e9 - 1).sum()), "bins hit")
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "scan_golden.json"), "w"), indent=0)
