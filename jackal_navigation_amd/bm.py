"""Host-side mirror of the block-matching mode (include/jn_bm.h) over libjn_stereo.so.

The reference has no block matcher (its only matcher is libelas); this mode is defined in include/jn_bm.h and slots in where
generateDisparityMap (point_cloud.cpp:406-429) calls Elas::process: rectified pair in, disparity map out."""
import ctypes as C

from . import _lib


class BmParams(C.Structure):
    _fields_ = [("num_disparities", C.c_int32), ("block_radius", C.c_int32), ("prefilter_cap", C.c_int32),
                ("lr_max_diff", C.c_int32), ("subpixel", C.c_int32), ("cost_function", C.c_int32)]


COST_SAD, COST_SSD = 0, 1


class BmTimes(C.Structure):
    _fields_ = [("prefilter", C.c_float), ("match", C.c_float), ("finish", C.c_float), ("total", C.c_float)]


def _bind():
    L = _lib.load()
    if not getattr(L, "_bm_bound", False):
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        L.jn_bm_params_default.argtypes = [C.POINTER(BmParams)]
        L.jn_bm_params_default.restype = None
        L.jn_bm_create.argtypes = [C.POINTER(BmParams), i32, i32, i32, i32, C.POINTER(vp)]
        L.jn_bm_destroy.argtypes = [vp]
        L.jn_bm_destroy.restype = None
        L.jn_bm_process_batch.argtypes = [vp, i32, vp, vp, i32, i64, vp]
        L.jn_bm_process_scan.argtypes = [vp, i32, vp, vp, i32, i64, vp, vp, vp, vp, vp, vp]
        L.jn_bm_submit_scan.argtypes = [vp, i32, i32, vp, vp, i32, i64, vp, vp, vp, vp, vp, vp]
        L.jn_bm_wait.argtypes = [vp, i32]
        L.jn_bm_last_times.argtypes = [vp, C.POINTER(BmTimes)]
        L.jn_sgm_disparity_to_u8.argtypes = [i32, vp, i32, vp, i64]
        L._bm_bound = True
    return L


BM_EXPORTS = ["jn_bm_params_default", "jn_bm_create", "jn_bm_destroy", "jn_bm_process_batch", "jn_bm_process_scan", "jn_bm_submit_scan", "jn_bm_wait",
              "jn_bm_last_times"]


class Bm:
    @staticmethod
    def parameters(**overrides):
        p = BmParams()
        _bind().jn_bm_params_default(C.byref(p))
        for k, v in overrides.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        return p

    def __init__(self, param, width, height, max_batch=1, device=0):
        self._L = _bind()
        self.param, self.width, self.height, self.max_batch, self.device = param, int(width), int(height), int(max_batch), int(device)
        h = C.c_void_p()
        _lib.check(self._L.jn_bm_create(C.byref(param), width, height, max_batch, device, C.byref(h)), "jn_bm_create")
        self._h = h

    def process_batch(self, n, dI1, dI2, pitch, image_stride, dDisp):
        _lib.check(self._L.jn_bm_process_batch(self._h, n, dI1, dI2, pitch, image_stride, dDisp), "jn_bm_process_batch")

    def process_scan(self, n, dI1, dI2, pitch, image_stride, dDisp, scan_params, dLut, dU8, dBins, dMeta):
        """Matcher + u8 map + LUT scan on one stream with one synchronisation (jn_bm_process_scan)."""
        _lib.check(self._L.jn_bm_process_scan(self._h, n, dI1, dI2, pitch, image_stride, dDisp, C.byref(scan_params), dLut, dU8, dBins, dMeta),
                   "jn_bm_process_scan")

    def submit_scan(self, slot, n, dI1, dI2, pitch, image_stride, dDisp, scan_params=None, dLut=None, dU8=None, dBins=None, dMeta=None):
        """jn_bm_submit_scan: queue a batch on `slot` (0..5) and return; without scan_params only the disparities are produced."""
        _lib.check(self._L.jn_bm_submit_scan(self._h, slot, n, dI1, dI2, pitch, image_stride, dDisp,
                                             C.byref(scan_params) if scan_params is not None else None, dLut, dU8, dBins, dMeta), "jn_bm_submit_scan")

    def wait(self, slot):
        _lib.check(self._L.jn_bm_wait(self._h, slot), "jn_bm_wait")

    def last_times(self):
        t = BmTimes()
        _lib.check(self._L.jn_bm_last_times(self._h, C.byref(t)), "jn_bm_last_times")
        return {k: float(getattr(t, k)) for k, _ in t._fields_}

    def to_u8(self, dDisp, dOut, n):
        _lib.check(self._L.jn_sgm_disparity_to_u8(self.device, dDisp, self.param.subpixel, dOut, n), "jn_sgm_disparity_to_u8")   # same output format

    def close(self):
        if getattr(self, "_h", None):
            self._L.jn_bm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
