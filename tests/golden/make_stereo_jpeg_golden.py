#!/usr/bin/env python3
"""Generates tests/golden/stereo_jpeg_pair.npz: one synthetic stereo pair at the webcam's 640x360 (launch/stereo.launch:4-5)
as the two JPEG byte strings the camera driver would publish on webcam/left|right/image_raw/compressed, plus the SHA-256 of the
grey frames libjpeg reconstructs from them (Pillow = libjpeg-turbo, draft('L'): what cv::imdecode(GRAYSCALE) returns at
point_cloud.cpp:436, :478).  Input of the whole-frame test (JPEG bytes -> decode -> rectify -> ELAS -> scan).

    python tests/golden/make_stereo_jpeg_golden.py
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
    return np.asarray(im).copy()


def raw_frame(o, rect, K, D, R, P, rng):
    """A 640x360 sensor frame whose rectification (the node's maps for this camera) shows `rect` (320x180): the map gives, per
    rectified pixel, its source position on the sensor; inverted on the sensor grid by scattered-data interpolation, the
    rectified image is then sampled bilinearly there.  Sensor pixels outside the rectified field of view get noise."""
    from scipy.interpolate import griddata
    from scipy.ndimage import map_coordinates
    H, W = rect.shape
    mx, my = o.undistort_map(list(K), list(D), list(R), list(P), W, H)
    vv, uu = np.mgrid[0:H, 0:W]
    pts = np.stack([mx.ravel(), my.ravel()], axis=1)
    yy, xx = np.mgrid[0:360, 0:640]
    u = griddata(pts, uu.ravel().astype(np.float64), (xx, yy), method="linear")
    v = griddata(pts, vv.ravel().astype(np.float64), (xx, yy), method="linear")
    inside = ~np.isnan(u)
    raw = rng.integers(0, 256, (360, 640)).astype(np.float64)
    samp = map_coordinates(rect.astype(np.float64), [np.nan_to_num(v), np.nan_to_num(u)], order=1, mode="nearest")
    raw[inside] = samp[inside]
    return np.clip(np.rint(raw), 0, 255).astype(np.uint8)


def main():
    import jackal_navigation_amd as jn                      # host-only calls: the calibration constants and stereoRectify
    from jackal_navigation_amd import node
    o = Oracle()
    c = node.stereo_calib()
    r = node.stereo_rectify(c, 320, 180)
    L, R = o.synth_pair(320, 180, 40, 4242)                  # what the rectified pair should show
    rng = np.random.default_rng(7)
    out = {}
    for name, img, K, D, Rr, P in (("left", L, c.K1, c.D1, r.R1, r.P1), ("right", R, c.K2, c.D2, r.R2, r.P2)):
        big = raw_frame(o, img, K, D, Rr, P, rng)
        rgb = np.stack([big, big, big], axis=2)
        buf = io.BytesIO()
        Image.fromarray(rgb).save(buf, "JPEG", quality=85, subsampling=2)
        data = buf.getvalue()
        g = decode_gray(data)
        out[name + "__jpeg"] = np.frombuffer(data, np.uint8)
        out[name + "__sha256"] = np.frombuffer(hashlib.sha256(g.tobytes()).digest(), np.uint8)
        print(name, len(data), "bytes", g.shape)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "stereo_jpeg_pair.npz"), **out)


if __name__ == "__main__":
    main()
