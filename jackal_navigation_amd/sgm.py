"""Host-side mirror of the SGM mode (include/jn_sgm.h) over libjn_stereo.so.

The reference has no SGM (its only matcher is libelas); this mode is defined in include/jn_sgm.h and slots in where
generateDisparityMap (point_cloud.cpp:406-429) calls Elas::process: rectified pair in, disparity map out."""
import ctypes as C

from . import _lib


class SgmParams(C.Structure):
    _fields_ = [("num_disparities", C.c_int32), ("P1", C.c_int32), ("P2", C.c_int32), ("prefilter_cap", C.c_int32),
                ("lr_max_diff", C.c_int32), ("subpixel", C.c_int32)]


class SgmTimes(C.Structure):
    _fields_ = [("prefilter", C.c_float), ("paths", C.c_float), ("wta", C.c_float), ("total", C.c_float)]


def _bind():
    L = _lib.load()
    if not getattr(L, "_sgm_bound", False):
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        L.jn_sgm_params_default.argtypes = [C.POINTER(SgmParams)]
        L.jn_sgm_params_default.restype = None
        L.jn_sgm_create.argtypes = [C.POINTER(SgmParams), i32, i32, i32, i32, C.POINTER(vp)]
        L.jn_sgm_destroy.argtypes = [vp]
        L.jn_sgm_destroy.restype = None
        L.jn_sgm_process_batch.argtypes = [vp, i32, vp, vp, i32, i64, vp]
        L.jn_sgm_last_times.argtypes = [vp, C.POINTER(SgmTimes)]
        L.jn_sgm_disparity_to_u8.argtypes = [i32, vp, i32, vp, i64]
        L.jn_sgm_debug_ptr.argtypes = [vp, i32, C.POINTER(i32 * 5)]
        L.jn_sgm_debug_ptr.restype = vp
        L.jn_sgm_submit_scan.argtypes = [vp, i32, i32, vp, vp, i32, i64, vp, vp, vp, vp, vp, vp]
        L.jn_sgm_wait.argtypes = [vp, i32]
        L._sgm_bound = True
    return L


SGM_EXPORTS = ["jn_sgm_params_default", "jn_sgm_create", "jn_sgm_destroy", "jn_sgm_process_batch", "jn_sgm_last_times",
               "jn_sgm_disparity_to_u8", "jn_sgm_debug_ptr", "jn_sgm_submit_scan", "jn_sgm_wait"]


class Sgm:
    @staticmethod
    def parameters(**overrides):
        p = SgmParams()
        _bind().jn_sgm_params_default(C.byref(p))
        for k, v in overrides.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        return p

    def __init__(self, param, width, height, max_batch=1, device=0):
        self._L = _bind()
        self.param, self.width, self.height, self.max_batch, self.device = param, int(width), int(height), int(max_batch), int(device)
        h = C.c_void_p()
        _lib.check(self._L.jn_sgm_create(C.byref(param), width, height, max_batch, device, C.byref(h)), "jn_sgm_create")
        self._h = h

    def process_batch(self, n, dI1, dI2, pitch, image_stride, dDisp):
        _lib.check(self._L.jn_sgm_process_batch(self._h, n, dI1, dI2, pitch, image_stride, dDisp), "jn_sgm_process_batch")

    def submit_scan(self, slot, n, dI1, dI2, pitch, image_stride, dDisp, scan_params=None, dLut=None, dU8=None, dBins=None, dMeta=None):
        """Asynchronous: the whole mode (+ u8 map + LUT scan when scan_params is given) queued on the slot's stream (jn_sgm_submit_scan)."""
        _lib.check(self._L.jn_sgm_submit_scan(self._h, slot, n, dI1, dI2, pitch, image_stride, dDisp,
                                              C.byref(scan_params) if scan_params is not None else None, dLut, dU8, dBins, dMeta), "jn_sgm_submit_scan")

    def wait(self, slot):
        _lib.check(self._L.jn_sgm_wait(self._h, slot), "jn_sgm_wait")

    def last_times(self):
        t = SgmTimes()
        _lib.check(self._L.jn_sgm_last_times(self._h, C.byref(t)), "jn_sgm_last_times")
        return {k: float(getattr(t, k)) for k, _ in t._fields_}

    def to_u8(self, dDisp, dOut, n):
        _lib.check(self._L.jn_sgm_disparity_to_u8(self.device, dDisp, self.param.subpixel, dOut, n), "jn_sgm_disparity_to_u8")

    def debug_ptr(self, which):
        """(device pointer, info) of an intermediate buffer of the last batch (include/jn_sgm.h jn_sgm_debug_ptr)."""
        info = (C.c_int32 * 5)()
        ptr = self._L.jn_sgm_debug_ptr(self._h, which, C.byref(info))
        return ptr, list(info)

    def close(self):
        if getattr(self, "_h", None):
            self._L.jn_sgm_destroy(self._h)
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
