"""Known-answer hashes of the compiled reference (oracle/_ref, zero-filled heap) with param.subsampling = 1: half-size maps.
Run where /root/reference exists:  python3 tests/golden/make_subsampling_golden.py   -> tests/golden/reference_subsampling_hashes.txt"""
import os, sys
here = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(here))
sys.path.insert(0, ROOT)
from oracle.binding import Oracle, Reference

CASES = [(320, 240, 40, 79, 21, {}), (640, 480, 64, 63, 12345, {"postprocess_only_left": 0}),
         (256, 200, 40, 63, 1, {"filter_median": 1, "postprocess_only_left": 0}),
         (400, 304, 60, 127, 6, {"candidate_stepsize": 4, "ipol_gap_width": 7, "speckle_size": 50, "postprocess_only_left": 0}),
         (322, 182, 48, 255, 12345, {"filter_adaptive_mean": 0}), (1280, 720, 128, 127, 12345, {"postprocess_only_left": 0})]
o, r = Oracle(), Reference()
lines = []
for (W, H, sd, dmax, seed, kw) in CASES:
    L, R = o.synth_pair(W, H, sd, seed)
    D1, D2 = r.process(r.params(0, disp_max=dmax, subsampling=1, **kw), L, R)
    assert D1.shape == (H // 2, W // 2)
    opts = ",".join("%s=%s" % kv for kv in sorted(kw.items())) or "-"
    lines.append("%d %d %d %d %d %s %016x %016x" % (W, H, sd, dmax, seed, opts, o.fnv(D1), o.fnv(D2)))
    print(lines[-1])
with open(os.path.join(here, "reference_subsampling_hashes.txt"), "w") as f:
    f.write("# W H scene_disp disp_max seed options fnv1a64(D1) fnv1a64(D2) -- reference src/elas, ROBOTICS params + subsampling=1 (+ options), maps (W/2)x(H/2), uninitialised allocations zero-filled\n")
    f.write("\n".join(lines) + "\n")
