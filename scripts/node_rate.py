"""Whole-frame rate of the node path from compressed images to the LaserScan, the way point_cloud.cpp runs it (one frame at
a time: point_cloud.cpp:431-490 -> :406-429 -> :213-296), on one GPU:  python3 scripts/node_rate.py [frames]
Input: the two JPEG frames of tests/golden/stereo_jpeg_pair.npz (640x360 webcam frames); per frame: entropy decode (host) +
IDCT (GPU) x2, remap x2 with the shipped calibration's maps, ELAS on device pointers + u8 map + LUT scan (one submit_scan),
bins and extrema copied to the host, message assembled.  Prints ms per frame and the split."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jackal_navigation_amd as jn  # noqa: E402
from jackal_navigation_amd import node  # noqa: E402
from jackal_navigation_amd.device import DeviceArray  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    pair = not (len(sys.argv) > 2 and sys.argv[2] == "serial")          # `serial`: the two eyes one after the other (round 2)
    z = np.load(os.path.join(ROOT, "tests", "golden", "stereo_jpeg_pair.npz"))
    jl, jr = np.ascontiguousarray(z["left__jpeg"]), np.ascontiguousarray(z["right__jpeg"])
    from jackal_navigation_amd import _lib
    L = _lib.load()
    raws = [DeviceArray((360, 640), np.uint8), DeviceArray((360, 640), np.uint8)]       # persistent, as a node would keep them
    ww, hh = C.c_int32(), C.c_int32()
    W, H = 320, 180
    c = node.stereo_calib()
    r = node.stereo_rectify(c, W, H)
    maps = [node.init_undistort_rectify_map(list(K), list(D), list(Rr), list(P), W, H)
            for K, D, Rr, P in ((c.K1, c.D1, r.R1, r.P1), (c.K2, c.D2, r.R2, r.P2))]
    sp = node.scan_params(W, H)
    lut = node.build_valid_disp_lut(sp, W, H)
    rect = [DeviceArray((H, W), np.uint8), DeviceArray((H, W), np.uint8)]
    d1 = DeviceArray.from_numpy(np.zeros((1, H, W), np.float32)); d2 = DeviceArray.from_numpy(np.zeros((1, H, W), np.float32))
    u8 = DeviceArray((1, H, W), np.uint8); bins = DeviceArray((1, sp.bins), np.float64); meta = DeviceArray((1, 4), np.float64)
    st = (C.c_int32 * 1)()
    t_dec = t_remap = t_match = t_msg = 0.0
    with jn.Elas(jn.Elas.parameters(0), W, H, max_batch=1, host_threads=8) as e:
        def frame():
            nonlocal t_dec, t_remap, t_match, t_msg
            t0 = time.perf_counter()
            if pair:                                                                # both eyes, entropy decodes on two threads
                _lib.check(L.jn_jpeg_decode_gray_pair(0, jl.ctypes.data, jl.size, jr.ctypes.data, jr.size, raws[0].ptr, raws[1].ptr, 640, 360,
                                                      C.byref(ww), C.byref(hh)), "jn_jpeg_decode_gray_pair")
            else:
                for buf, out in ((jl, raws[0]), (jr, raws[1])):                     # cv::imdecode(GRAYSCALE), point_cloud.cpp:436, :478
                    _lib.check(L.jn_jpeg_decode_gray(0, buf.ctypes.data, buf.size, out.ptr, 640, 360, C.byref(ww), C.byref(hh)), "jn_jpeg_decode_gray")
            t1 = time.perf_counter()
            for i in range(2):
                node.remap(1, raws[i].ptr, 640, 360, 640, 640 * 360, maps[i][0].ptr, maps[i][1].ptr, rect[i].ptr, W, H, W, W * H)
            t2 = time.perf_counter()
            e.submit_scan(0, 1, rect[0].ptr, rect[1].ptr, W, H * W, d1.ptr, d2.ptr, sp, lut.ptr, u8.ptr, bins.ptr, meta.ptr, st)
            e.wait(0)
            t3 = time.perf_counter()
            msg = node.laser_scan_message(bins.numpy()[0], meta.numpy()[0], seq=0)
            t4 = time.perf_counter()
            t_dec += t1 - t0; t_remap += t2 - t1; t_match += t3 - t2; t_msg += t4 - t3
            return msg
        for _ in range(20):
            msg = frame()
        t_dec = t_remap = t_match = t_msg = 0.0
        t0 = time.perf_counter()
        for _ in range(frames):
            msg = frame()
        el = time.perf_counter() - t0
    print("[%s decode] " % ("two-thread" if pair else "serial") + "node path, 640x360 JPEG pair -> 320x180 ELAS (disp_max 255) -> 90-bin scan: %.3f ms per frame = %.0f frames/s  "
          "(decode x2 %.3f, remap x2 %.3f, ELAS + u8 + scan %.3f, D2H + message %.3f); %d ranges, status %d" %
          (el / frames * 1e3, frames / el, t_dec / frames * 1e3, t_remap / frames * 1e3, t_match / frames * 1e3, t_msg / frames * 1e3,
           len(msg["ranges"]), st[0]))


if __name__ == "__main__":
    main()
