"""Host-side mirror of the reference's `Elas` interface (src/elas/elas.h:52-162) over libjn_stereo.so.

    param = Elas.parameters(Elas.ROBOTICS); param.postprocess_only_left = True    # point_cloud.cpp:416-417
    elas = Elas(param, width, height)
    elas.process(I1, I2, D1, D2, dims)                                            # point_cloud.cpp:419

`process` keeps the reference's contract: caller-allocated float32 D1/D2 of width*height, dims =
(width, height, bytes-per-line), outputs untouched when fewer than 3 support points are found (the
reference prints an error and returns; this mirror additionally returns the status code).
Batched/pipelined entry points take device pointers (DeviceArray.ptr or torch tensor .data_ptr()).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ElasParams, StageTimes


class Elas:
    ROBOTICS = 0      # elas.h:57
    MIDDLEBURY = 1

    @staticmethod
    def parameters(setting=0, **overrides):
        """Elas::parameters(setting) — elas.h:85-145."""
        p = ElasParams()
        _lib.load().jn_elas_params_default(C.byref(p), setting)
        for k, v in overrides.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        return p

    def __init__(self, param, width, height, max_batch=1, device=0, host_threads=0, slots=1):
        self._L = _lib.load()
        self.param, self.width, self.height = param, int(width), int(height)
        self.max_batch, self.device, self.slots = int(max_batch), int(device), int(slots)
        h = C.c_void_p()
        _lib.check(self._L.jn_elas_create(C.byref(param), width, height, max_batch, device, host_threads, slots, C.byref(h)),
                   "jn_elas_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_comm", None) is not None:      # detach first (waits for the batches in flight), then let go of the communicator
                self._L.jn_elas_set_comm(self._h, None)
            self._L.jn_elas_destroy(self._h)
            self._h = None
        self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- Elas::process(I1, I2, D1, D2, dims) ---------------------------------------------------
    def process(self, I1, I2, D1, D2, dims=None):
        if dims is None:
            dims = (self.width, self.height, I1.strides[0] if I1.ndim == 2 else self.width)
        for a, t in ((I1, np.uint8), (I2, np.uint8), (D1, np.float32), (D2, np.float32)):
            if a.dtype != t or not a.flags["C_CONTIGUOUS"]:
                raise TypeError("expected C-contiguous %s array" % np.dtype(t).name)
        cdims = (C.c_int32 * 3)(*[int(x) for x in dims])
        st = self._L.jn_elas_process(self._h, I1.ctypes.data, I2.ctypes.data, D1.ctypes.data, D2.ctypes.data, C.byref(cdims))
        if st not in (_lib.JN_OK, _lib.JN_ERR_FEW_SUPPORT):
            raise _lib.JnError(st, "jn_elas_process")
        return st

    # -- batched, device pointers -----------------------------------------------------------------
    def process_batch(self, n, dI1, dI2, pitch, image_stride, dD1, dD2):
        status = (C.c_int32 * n)()
        _lib.check(self._L.jn_elas_process_batch(self._h, n, dI1, dI2, pitch, image_stride, dD1, dD2, status), "jn_elas_process_batch")
        return list(status)

    def submit(self, slot, n, dI1, dI2, pitch, image_stride, dD1, dD2, status=None):
        _lib.check(self._L.jn_elas_submit(self._h, slot, n, dI1, dI2, pitch, image_stride, dD1, dD2, status), "jn_elas_submit")

    def submit_host(self, slot, Ls, Rs, D1, D2, status=None):
        """jn_elas_submit_host: numpy arrays [n][H][W] (uint8 in, float32 out, C-contiguous) staged by the slot's worker; the
        arrays must stay alive until wait(slot)."""
        n, H, W = Ls.shape
        assert Ls.flags["C_CONTIGUOUS"] and Rs.flags["C_CONTIGUOUS"] and D1.flags["C_CONTIGUOUS"] and D2.flags["C_CONTIGUOUS"]
        assert Ls.dtype.itemsize == 1 and D1.dtype.itemsize == 4 and D1.shape == Ls.shape == Rs.shape == D2.shape
        _lib.check(self._L.jn_elas_submit_host(self._h, slot, n, Ls.ctypes.data, Rs.ctypes.data, W, H * W, D1.ctypes.data, D2.ctypes.data, status),
                   "jn_elas_submit_host")

    def submit_scan(self, slot, n, dI1, dI2, pitch, image_stride, dD1, dD2, sp, dLut, dDispU8, dBins, dMeta, status=None):
        """submit() plus the node's tail (u8 depth map + LUT obstacle scan of D1) on the same stream."""
        _lib.check(self._L.jn_elas_submit_scan(self._h, slot, n, dI1, dI2, pitch, image_stride, dD1, dD2, C.byref(sp), dLut,
                                               dDispU8, dBins, dMeta, status), "jn_elas_submit_scan")

    def set_comm(self, comm):
        """Attach a parallel.ScanComm (or None): scan batches then end with the cross-rig MIN reduce (jn_elas_set_comm)."""
        _lib.check(self._L.jn_elas_set_comm(self._h, comm._h if comm is not None else None), "jn_elas_set_comm")
        self._comm = comm            # the library keeps the raw jn_comm*: the ScanComm must outlive its use there (its __del__ destroys the communicator)

    def merge_order(self, cap=4096):
        """Submission numbers of the last scan batches in the order their merges were queued (jn_elas_merge_order; a testing aid)."""
        import ctypes as C
        buf = (C.c_uint64 * cap)()
        k = self._L.jn_elas_merge_order(self._h, buf, cap)
        return list(buf[:k])

    def merge_time(self, slot=0):
        import ctypes as C
        ms = C.c_float(0)
        _lib.check(self._L.jn_elas_merge_time(self._h, slot, C.byref(ms)), "jn_elas_merge_time")
        return ms.value

    def wait(self, slot):
        _lib.check(self._L.jn_elas_wait(self._h, slot), "jn_elas_wait")

    def last_times(self, slot=0):
        t = StageTimes()
        _lib.check(self._L.jn_elas_last_times(self._h, slot, C.byref(t)), "jn_elas_last_times")
        return t.as_dict()

    def bin_stats(self, slot=0):
        """(longest triangle list of a 32x8 tile, tiles with more than 16 entries, tiles beyond 64) of the slot's last batch (jn_elas_bin_stats; a testing aid)."""
        out = (C.c_int32 * 3)()
        _lib.check(self._L.jn_elas_bin_stats(self._h, slot, C.byref(out)), "jn_elas_bin_stats")
        return int(out[0]), int(out[1]), int(out[2])

    def route_stats(self, slot=0):
        """(triangulates on the GPU, batches of the slot that fell back to the host stage, descriptors from the Sobel planes) — jn_elas_route_stats, a testing aid."""
        out = (C.c_int32 * 3)()
        _lib.check(self._L.jn_elas_route_stats(self._h, slot, C.byref(out)), "jn_elas_route_stats")
        return int(out[0]), int(out[1]), int(out[2])

    def kernel_time(self, slot=0, kernel=b"k_dense"):
        ms, cnt = C.c_float(), C.c_int32()
        _lib.check(self._L.jn_elas_kernel_time(self._h, slot, kernel, C.byref(ms), C.byref(cnt)), "jn_elas_kernel_time")
        return ms.value, cnt.value
