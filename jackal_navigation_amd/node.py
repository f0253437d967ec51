"""Host-side mirror of the `point_cloud` node functions that follow Elas::process
(reference src/obstacle_avoidance/point_cloud.cpp), over libjn_stereo.so.

    generateDisparityMap  (:406-429)  -> disparity_to_u8
    cacheDisparityValues  (:104-147)  -> build_valid_disp_lut
    publishObstacleScan   (:213-296)  -> obstacle_scan / disparity_scan (+ laser_scan_message)
    publishPointCloud -g  (:298-404)  -> point_cloud
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ScanParams
from .device import DeviceArray

INF = 1e9   # point_cloud.cpp:54


def scan_params(width, height):
    sp = ScanParams()
    _lib.load().jn_scan_params_default(C.byref(sp), width, height)
    return sp


def disparity_to_u8(dD, dOut, n, device=0):
    _lib.check(_lib.load().jn_disparity_to_u8(device, dD, dOut, n), "jn_disparity_to_u8")


def build_valid_disp_lut(sp, width, height, device=0):
    lut = DeviceArray((height, width, 2), np.uint8, device)
    _lib.check(_lib.load().jn_build_valid_disp_lut(device, C.byref(sp), width, height, lut.ptr), "jn_build_valid_disp_lut")
    return lut


def obstacle_scan(sp, n, dDisp, dLut, width, height, dBins, dMeta, device=0):
    _lib.check(_lib.load().jn_obstacle_scan(device, C.byref(sp), n, dDisp, dLut, width, height, dBins, dMeta), "jn_obstacle_scan")


def obstacle_scan_cloud(sp, n, dDisp, width, height, dBins, dMeta, device=0):
    """-g mode: publishPointCloud + publishObstacleScan(vector<Point3d>) (point_cloud.cpp:321-352, :149-211)."""
    _lib.check(_lib.load().jn_obstacle_scan_cloud(device, C.byref(sp), n, dDisp, width, height, dBins, dMeta), "jn_obstacle_scan_cloud")


def disparity_scan(sp, n, dD, dLut, width, height, dDispU8, dBins, dMeta, device=0):
    _lib.check(_lib.load().jn_disparity_scan(device, C.byref(sp), n, dD, dLut, width, height, dDispU8, dBins, dMeta), "jn_disparity_scan")


def compact_ranges(bins):
    bins = np.ascontiguousarray(bins, np.float64)
    out = np.zeros(bins.shape[0], np.float32)
    k = _lib.load().jn_compact_ranges(bins.ctypes.data, bins.shape[0], out.ctypes.data)
    return out[:k].copy()


def laser_scan_message(bins, meta, seq=0):
    """The sensor_msgs/LaserScan fields exactly as point_cloud.cpp:271-283 fills them."""
    return {
        "header": {"seq": int(seq), "frame_id": "jackal"},
        "angle_min": np.float32(meta[0]), "angle_max": np.float32(meta[1]),
        "range_min": np.float32(meta[2]), "range_max": np.float32(meta[3]),
        "angle_increment": np.float32(3.1415 / 180.), "scan_time": np.float32(0.001), "time_increment": np.float32(0.1),
        "ranges": compact_ranges(bins),
    }


def point_cloud(sp, dDisp, width, height, device=0):
    xyz = DeviceArray((height * width, 3), np.float32, device)
    cnt = C.c_int64(0)
    _lib.check(_lib.load().jn_point_cloud(device, C.byref(sp), dDisp, width, height, xyz.ptr, C.byref(cnt)), "jn_point_cloud")
    out = xyz.numpy()[:cnt.value].copy()
    xyz.free()
    return out


def stereo_calib():
    """K1,K2,D1,D2,R,T of calibration/amrl_jackal_webcam_stereo.yml (640x360)."""
    c = _lib.StereoCalib()
    _lib.load().jn_stereo_calib_default(C.byref(c))
    return c


def stereo_rectify(calib, new_width, new_height):
    """cv::stereoRectify(..., CV_CALIB_ZERO_DISPARITY, 0, newImageSize) — point_cloud.cpp:543-544."""
    r = _lib.Rectification()
    _lib.check(_lib.load().jn_stereo_rectify(C.byref(calib), new_width, new_height, C.byref(r)), "jn_stereo_rectify")
    return r


def init_undistort_rectify_map(K, D, R, P, width, height, device=0):
    """cv::initUndistortRectifyMap(K, D, R, P, size, CV_32F) — point_cloud.cpp:553-554; returns device float maps."""
    mx = DeviceArray((height, width), np.float32, device)
    my = DeviceArray((height, width), np.float32, device)
    arr = [np.ascontiguousarray(np.asarray(a, np.float64)) for a in (K, D, R, P)]
    _lib.check(_lib.load().jn_init_undistort_rectify_map(device, arr[0].ctypes.data, arr[1].ctypes.data, arr[2].ctypes.data,
                                                         arr[3].ctypes.data, width, height, mx.ptr, my.ptr), "jn_init_undistort_rectify_map")
    return mx, my


def remap(n, dSrc, src_w, src_h, src_pitch, src_stride, dMapX, dMapY, dDst, width, height, dst_pitch, dst_stride, device=0):
    """cv::remap(src, dst, mapx, mapy, INTER_LINEAR) — point_cloud.cpp:440, :481 (batched, device pointers)."""
    _lib.check(_lib.load().jn_remap_bilinear(device, n, dSrc, src_w, src_h, src_pitch, src_stride, dMapX, dMapY, dDst, width, height,
                                             dst_pitch, dst_stride), "jn_remap_bilinear")


def jpeg_info(data):
    """(width, height) of a JPEG frame; raises JnError(JN_ERR_UNSUPPORTED) for progressive / arithmetic / 12-bit files."""
    buf = np.frombuffer(bytes(data), np.uint8)
    w, h = C.c_int32(), C.c_int32()
    _lib.check(_lib.load().jn_jpeg_info(buf.ctypes.data, buf.size, C.byref(w), C.byref(h)), "jn_jpeg_info")
    return w.value, h.value


def imdecode_gray(data, device=0):
    """cv::imdecode(data, CV_LOAD_IMAGE_GRAYSCALE) — point_cloud.cpp:436, :478; returns the grey frame as a DeviceArray."""
    buf = np.frombuffer(bytes(data), np.uint8)
    w, h = jpeg_info(buf)
    out = DeviceArray((h, w), np.uint8, device)
    ww, hh = C.c_int32(), C.c_int32()
    _lib.check(_lib.load().jn_jpeg_decode_gray(device, buf.ctypes.data, buf.size, out.ptr, w, h, C.byref(ww), C.byref(hh)), "jn_jpeg_decode_gray")
    return out


def imdecode_gray_pair(left, right, device=0):
    """Both eyes of a stereo frame, entropy-decoded on two threads (jn_jpeg_decode_gray_pair); returns two DeviceArrays."""
    bl, br = np.frombuffer(bytes(left), np.uint8), np.frombuffer(bytes(right), np.uint8)
    w, h = jpeg_info(bl)
    outs = DeviceArray((h, w), np.uint8, device), DeviceArray((h, w), np.uint8, device)
    ww, hh = C.c_int32(), C.c_int32()
    _lib.check(_lib.load().jn_jpeg_decode_gray_pair(device, bl.ctypes.data, bl.size, br.ctypes.data, br.size, outs[0].ptr, outs[1].ptr, w, h,
                                                    C.byref(ww), C.byref(hh)), "jn_jpeg_decode_gray_pair")
    return outs


def synth_pair(width, height, scene_disp, seed=12345):
    L = np.zeros((height, width), np.uint8)
    R = np.zeros((height, width), np.uint8)
    _lib.load().jn_synth_pair(width, height, scene_disp, seed, L.ctypes.data, R.ctypes.data)
    return L, R
