"""Test helpers: device arrays through the C ABI's own allocator (no torch needed at this level)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from pailliercryptolib_python_amd import _native


class DevArray:
    """A uint32/int32 matrix in device memory owned by the test."""

    def __init__(self, host: np.ndarray | None = None, shape=None, dtype=np.uint32, device: int = 0):
        self.device = device
        self.lib = _native.load()
        if host is not None:
            host = np.ascontiguousarray(host)
            shape, dtype = host.shape, host.dtype
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _native.check(self.lib.pai_malloc(device, max(self.nbytes, 4), C.byref(p)))
        self.ptr = p
        if host is not None and self.nbytes:
            _native.check(self.lib.pai_memcpy_h2d(device, self.ptr, host.ctypes.data_as(C.c_void_p), self.nbytes, None))

    def get(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            _native.check(self.lib.pai_memcpy_d2h(self.device, out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes, None))
        return out

    def free(self):
        if self.ptr:
            self.lib.pai_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def ints_to_limbs(vals, L):
    out = np.zeros((len(vals), L), dtype=np.uint32)
    for i, v in enumerate(vals):
        out[i] = np.frombuffer(int(v).to_bytes(4 * L, "little"), dtype="<u4")
    return out


def limbs_to_ints(arr):
    arr = np.ascontiguousarray(arr, dtype="<u4")
    return [int.from_bytes(row.tobytes(), "little") for row in arr]


def rand_below(rng, bound: int, count: int):
    nbytes = (bound.bit_length() + 7) // 8 + 8
    return [int.from_bytes(rng.bytes(nbytes), "little") % bound for _ in range(count)]


def host_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)
