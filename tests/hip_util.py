"""Device buffers for tests that call the launch-level C ABI (v2p_stitch_launch) directly: the HIP runtime through ctypes,
no torch (initialising torch after the library has taken the device fails on the test boxes)."""
import ctypes

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        _hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        _hip.hipFree.argtypes = [ctypes.c_void_p]
    return _hip


class DevBuf:
    """`nbytes` of device memory with `pad` zeroed bytes either side of the payload (ptr points at the payload)."""

    def __init__(self, nbytes: int, pad: int = 64, fill: int = 0):
        self.nbytes, self.pad = int(nbytes), pad
        p = ctypes.c_void_p()
        assert hip().hipMalloc(ctypes.byref(p), self.nbytes + 2 * pad + 16) == 0
        self.base = p.value
        assert hip().hipMemset(self.base, fill, self.nbytes + 2 * pad + 16) == 0
        self.ptr = self.base + pad

    @classmethod
    def of(cls, arr: np.ndarray, pad: int = 64):
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes, pad)
        if arr.nbytes:
            assert hip().hipMemcpy(b.ptr, arr.ctypes.data, arr.nbytes, 1) == 0
        return b

    def download(self) -> np.ndarray:
        out = np.empty(self.nbytes, dtype=np.uint8)
        assert hip().hipDeviceSynchronize() == 0
        if self.nbytes:
            assert hip().hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, 2) == 0
        return out

    def free(self):
        if self.base:
            hip().hipFree(self.base)
            self.base = 0
