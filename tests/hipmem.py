"""Minimal device-buffer helper for the GPU tests: hipMalloc / hipMemcpy through ctypes on the SAME HIP
runtime the library links (libamdhip64.so.7).  The tests deliberately avoid torch for device memory:
torch wheels bundle their own HIP runtime, and initialising a second runtime late in a process that has
already driven the GPU through the first one is fragile."""
import ctypes as C

import numpy as np

_hip = C.CDLL("libamdhip64.so.7")
_hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
_hip.hipFree.argtypes = [C.c_void_p]
_hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
_hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
_H2D, _D2H = 1, 2


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with hipError %d" % (what, rc))


class DeviceArray:
    """A device allocation shaped like a numpy array."""

    def __init__(self, shape, dtype, fill=None):
        self.shape, self.dtype = tuple(np.atleast_1d(shape)), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _chk(_hip.hipMalloc(C.byref(p), max(self.nbytes, 1)), "hipMalloc")
        self.ptr = p.value
        if fill is not None:
            self.upload(np.full(self.shape, fill, self.dtype))

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        d.upload(a)
        return d

    def upload(self, a):
        a = np.ascontiguousarray(a, self.dtype)
        _chk(_hip.hipMemcpy(self.ptr, a.ctypes.data, self.nbytes, _H2D), "hipMemcpy H2D")  # synchronous

    def numpy(self):
        _chk(_hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
        out = np.empty(self.shape, self.dtype)
        _chk(_hip.hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, _D2H), "hipMemcpy D2H")
        return out

    def free(self):
        if self.ptr:
            _hip.hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
