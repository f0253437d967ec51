#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libelas_ref.so, i.e. the
reference's own src/elas compiled from /root/reference by oracle/Makefile), run through
oracle.binding.Reference: a worker process whose allocations are zero-filled, because libelas reads
descriptor bytes it never wrote (readers listed in that class's docstring).  Run in the dev
container only; the fixtures are data (inputs + the reference's outputs per stage), no source.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Oracle, Reference  # noqa: E402

CASES = [  # name, W, H, scene disparity, disp_max, seed
    ("elas_160x120_d63_seed7", 160, 120, 32, 63, 7),
    ("elas_320x180_d255_seed12345", 320, 180, 48, 255, 12345),
]

# FNV-1a-64 of final D1 for the survey's larger cases, re-derived here from the reference
HASH_CASES = [(320, 180, 48, 255), (640, 480, 64, 63), (1280, 720, 128, 127), (1920, 1080, 256, 255)]


def main():
    o, r = Oracle(), Reference()
    here = os.path.dirname(os.path.abspath(__file__))
    for name, W, H, sd, dmax, seed in CASES:
        L, R = o.synth_pair(W, H, sd, seed)
        p = r.params(0, disp_max=dmax)
        out = {"L": L, "R": R, "disp_max": np.int32(dmax)}
        with r.open(p, L, R) as s:
            out["desc1_inner"] = s.descriptor(0)[3:H - 3, 3:W - 3]
            sup = s.support()
            out["support"] = sup
            raw = []
            for side in (0, 1):
                c, pl = s.triangles(side, len(sup))
                out["corners%d" % side] = c
                out["planes%d" % side] = pl
                g = s.grid(side)
                out["grid_counts%d" % side] = g[:, :, 0].astype(np.int16)
                out["grid%d" % side] = g.astype(np.int16)
                raw.append(s.dense(side))
            out["raw1"], out["raw2"] = raw
            a1, a2 = s.lr_check(raw[0], raw[1])
            out["lr1"], out["lr2"] = a1, a2
            b = s.speckle(a1)
            out["speckle1"] = b
            c_ = s.gap(b)
            out["gap1"] = c_
            out["final1_stagewise"] = s.adaptive_mean(c_)
        D1, D2 = r.process(p, L, R)
        assert np.array_equal(D1.view(np.uint32), out["final1_stagewise"].view(np.uint32))
        out["D1"], out["D2"] = D1, D2
        np.savez_compressed(os.path.join(here, name + ".npz"), **out)
        print(name, "support", len(sup), "hash %016x" % o.fnv(D1))
    lines = []
    for W, H, sd, dmax in HASH_CASES:
        L, R = o.synth_pair(W, H, sd, 12345)
        D1, D2 = r.process(r.params(0, disp_max=dmax), L, R)
        lines.append("%d %d %d %d %016x %016x" % (W, H, sd, dmax, o.fnv(D1), o.fnv(D2)))
        print(lines[-1])
    with open(os.path.join(here, "reference_hashes.txt"), "w") as f:
        f.write("# W H scene_disp disp_max fnv1a64(D1) fnv1a64(D2) -- reference src/elas, ROBOTICS params, seed 12345, D pre-filled 0, uninitialised allocations zero-filled\n")
        f.write("\n".join(lines) + "\n")
    # the other scene kinds (tests/scenes.py), both sides post-processed
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenes import make_scene, KINDS
    lines = []
    for kind in KINDS:
        for (W, H, dmax, seed) in ((320, 240, 79, 11), (640, 360, 127, 12)):
            L, R = make_scene(kind, W, H, dmax, seed)
            D1, D2 = r.process(r.params(0, disp_max=dmax, postprocess_only_left=0), L, R)
            lines.append("%s %d %d %d %d %016x %016x" % (kind, W, H, dmax, seed, o.fnv(D1), o.fnv(D2)))
            print(lines[-1])
    with open(os.path.join(here, "reference_scene_hashes.txt"), "w") as f:
        f.write("# kind W H disp_max seed fnv1a64(D1) fnv1a64(D2) -- reference src/elas, ROBOTICS params with postprocess_only_left=0,\n"
                "# tests/scenes.py make_scene(kind, W, H, disp_max, seed), D pre-filled 0, uninitialised allocations zero-filled\n")
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
