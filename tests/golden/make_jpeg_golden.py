#!/usr/bin/env python3
"""Generates tests/golden/jpeg_cases.npz: JPEG byte strings and the grey image libjpeg reconstructs from them
(Pillow = libjpeg-turbo, grey output colour space via draft('L'): luminance only, default "slow integer" IDCT — what
cv::imdecode(..., CV_LOAD_IMAGE_GRAYSCALE) returns at point_cloud.cpp:436, :478; OpenCV itself is not in this image).
The fixtures are data: inputs (encoded frames) and expected outputs (the grey image for small cases, its SHA-256 for all).

    python tests/golden/make_jpeg_golden.py
"""
import hashlib
import io
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Oracle  # noqa: E402


def decode_gray(data):
    im = Image.open(io.BytesIO(data))
    im.draft("L", im.size)
    im.load()
    assert im.mode == "L", im.mode
    return np.asarray(im).copy()


def scene(o, W, H, seed, colour=True):
    L, R = o.synth_pair(W, H, 40, seed)
    if not colour:
        return Image.fromarray(L)
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    rgb = np.stack([L, np.clip(L.astype(int) + 40 * np.sin(xx / 17.0), 0, 255).astype(np.uint8),
                    np.clip(R.astype(int) + rng.integers(-20, 20, (H, W)), 0, 255).astype(np.uint8)], axis=2)
    return Image.fromarray(rgb)


def main():
    o = Oracle()
    cases = {}
    spec = [  # name, W, H, colour, save kwargs
        ("webcam_640x360_q80_420", 640, 360, True, dict(quality=80, subsampling=2)),          # launch/stereo.launch:4-5 frame size
        ("q90_422", 320, 180, True, dict(quality=90, subsampling=1)),
        ("q75_444_optimized", 320, 180, True, dict(quality=75, subsampling=0, optimize=True)),
        ("ragged_333x201_q60_420", 333, 201, True, dict(quality=60, subsampling=2)),
        ("ragged_35x21_q95_422", 35, 21, True, dict(quality=95, subsampling=1)),
        ("grey_q85", 320, 240, False, dict(quality=85)),
        ("q100_444", 160, 120, True, dict(quality=100, subsampling=0)),
        ("q5_420", 160, 120, True, dict(quality=5, subsampling=2)),
        ("restart_rows_q80_420", 320, 180, True, dict(quality=80, subsampling=2, restart_marker_rows=1)),
        ("restart_blocks_q80_444", 200, 100, True, dict(quality=80, subsampling=0, restart_marker_blocks=7)),
    ]
    for name, W, H, colour, kw in spec:
        buf = io.BytesIO()
        scene(o, W, H, 100 + len(cases) // 2, colour).save(buf, "JPEG", **kw)
        data = buf.getvalue()
        cases[name + "__jpeg"] = np.frombuffer(data, np.uint8)
        gray = decode_gray(data)
        cases[name + "__shape"] = np.array(gray.shape, np.int32)
        cases[name + "__sha256"] = np.frombuffer(hashlib.sha256(gray.tobytes()).digest(), np.uint8)
        if gray.size <= 160 * 120:
            cases[name + "__gray"] = gray                       # small cases keep the full expected image, the others its digest
        print(name, len(data), "bytes")
    buf = io.BytesIO()
    scene(o, 160, 120, 7).save(buf, "JPEG", quality=80, progressive=True)
    cases["progressive__jpeg"] = np.frombuffer(buf.getvalue(), np.uint8)                       # must be refused
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz"), **cases)


if __name__ == "__main__":
    main()
