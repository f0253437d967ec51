"""Device-memory plumbing over the C-ABI (no torch needed): numpy <-> HBM."""
import ctypes as C

import numpy as np

from . import _lib


class DeviceArray:
    """A typed block of device memory owned through jn_device_malloc/jn_device_free."""

    def __init__(self, shape, dtype, device=0):
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.device = device
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _lib.check(_lib.load().jn_device_malloc(device, max(self.nbytes, 1), C.byref(p)), "jn_device_malloc")
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a, device=0):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype, device)
        d.upload(a)
        return d

    def upload(self, a):
        a = np.ascontiguousarray(a, self.dtype)
        assert a.nbytes == self.nbytes
        _lib.check(_lib.load().jn_memcpy_h2d(self.device, self.ptr, a.ctypes.data_as(C.c_void_p), self.nbytes), "jn_memcpy_h2d")

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        _lib.check(_lib.load().jn_memcpy_d2h(self.device, out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes), "jn_memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            _lib.load().jn_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def device_count():
    n = C.c_int32(0)
    _lib.load().jn_device_count(C.byref(n))
    return n.value
