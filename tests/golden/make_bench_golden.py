#!/usr/bin/env python3
"""Generates tests/golden/bench_batch_golden.json: what the REFERENCE makes of every pair of bench.py's headline batch (1280x720,
scene disparities <= 128, disp_max 127, Appendix-A generator, seeds 12345 .. 12376): FNV-1a-64 of D1 from the compiled reference
(oracle/_ref, zero-filled allocations, see oracle.binding.Reference), and the node's tail of that D1 — u8 map hash, the 90 bins and
the 4 extrema — as restated in oracle/node_oracle.cpp (OpenCV / ROS are not installed: those are definitions, point_cloud.cpp:422,
:104-147, :213-296).  bench.py checks ALL frames of ALL slots against it after the timed region.  Run in the dev container only:

    python tests/golden/make_bench_golden.py            # the headline batch -> bench_batch_golden.json
    python tests/golden/make_bench_golden.py vga        # the 640x480 D=64 batch of the line's vga_config (round 6: every frame of
                                                        # every slot is checked there too) -> bench_vga_golden.json
"""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

VGA = len(sys.argv) > 1 and sys.argv[1] == "vga"
W, H, SD, DMAX, SEED0, COUNT = (640, 480, 64, 63, 12345, 32) if VGA else (1280, 720, 128, 127, 12345, 32)


def one(seed):
    from oracle.binding import Oracle, Reference
    o, r = Oracle(), Reference()
    L, R = o.synth_pair(W, H, SD, seed)
    D1, _ = r.process(r.params(0, disp_max=DMAX), L, R)
    u8 = o.to_u8(D1)
    sp = o.scan_params(W, H)
    bins, meta, _ = o.scan(sp, u8, o.valid_lut(sp, W, H))
    return seed, {"d1_fnv": "%016x" % o.fnv(D1), "u8_fnv": "%016x" % o.fnv(np.ascontiguousarray(u8).view(np.uint32)),
                  "bins": [float(b) for b in bins], "meta": [float(m) for m in meta]}


if __name__ == "__main__":
    out = {"_what": "seed -> fnv1a64(D1 as u32 words), fnv1a64(u8 map as u32 words), bins[90], meta[4] of the reference on synth_pair(%d, %d, %d, seed), disp_max %d"
                    % (W, H, SD, DMAX), "config": [W, H, SD, DMAX]}
    with ProcessPoolExecutor(4) as ex:
        for seed, rec in ex.map(one, range(SEED0, SEED0 + COUNT)):
            out[str(seed)] = rec
            print(seed, rec["d1_fnv"], rec["u8_fnv"])
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "bench_vga_golden.json" if VGA else "bench_batch_golden.json"), "w"), indent=0)
