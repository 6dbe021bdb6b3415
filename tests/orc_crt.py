"""ctypes wrapper of oracle/_build/liboracle_crt.so (the GF(M61^2) x GF(M31^2) / PFA oracle).
Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "_build", "liboracle_crt.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_build/liboracle_crt.so"])
        L = C.CDLL(path)
        L.orcc_transform_size.restype = C.c_size_t; L.orcc_transform_size.argtypes = [C.c_uint32, C.c_uint]
        L.orcc_create.restype = C.c_void_p; L.orcc_create.argtypes = [C.c_uint32, C.c_uint, C.c_size_t]
        L.orcc_destroy.argtypes = [C.c_void_p]
        L.orcc_size.restype = C.c_size_t; L.orcc_size.argtypes = [C.c_void_p]
        L.orcc_widths.argtypes = [C.c_void_p, C.c_void_p]
        L.orcc_square_mul.argtypes = [C.c_void_p, C.c_uint32]
        L.orcc_set_u32.argtypes = [C.c_void_p, C.c_uint32]
        L.orcc_sub_u32.argtypes = [C.c_void_p, C.c_uint32]
        L.orcc_get_digits.argtypes = [C.c_void_p, C.c_void_p]
        L.orcc_set_digits.argtypes = [C.c_void_p, C.c_void_p]
        L.orcc_get_precarry.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orcc_get_words.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orcc_get_weighted.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class OracleCrt:
    def __init__(self, p, odd=1, n=0):
        self.L = lib()
        self.h = self.L.orcc_create(p, odd, n)
        if not self.h:
            raise ValueError("no admissible transform for p=%d, odd=%d, n=%d" % (p, odd, n))
        self.p, self.odd, self.n = p, odd, self.L.orcc_size(self.h)

    def close(self):
        if self.h:
            self.L.orcc_destroy(self.h); self.h = None

    def __del__(self):
        self.close()

    def set(self, a): self.L.orcc_set_u32(self.h, a)
    def square_mul(self, a=1): self.L.orcc_square_mul(self.h, a)
    def sub(self, a): self.L.orcc_sub_u32(self.h, a)

    def widths(self):
        w = np.zeros(self.n, dtype=np.uint8)
        self.L.orcc_widths(self.h, w.ctypes.data_as(C.c_void_p))
        return w

    def digits(self):
        d = np.zeros(self.n, dtype=np.uint64)
        self.L.orcc_get_digits(self.h, d.ctypes.data_as(C.c_void_p))
        return d

    def set_digits(self, d):
        d = np.ascontiguousarray(d, dtype=np.uint64)
        self.L.orcc_set_digits(self.h, d.ctypes.data_as(C.c_void_p))

    def precarry(self):
        r61 = np.zeros(self.n, dtype=np.uint64); r31 = np.zeros(self.n, dtype=np.uint32)
        self.L.orcc_get_precarry(self.h, r61.ctypes.data_as(C.c_void_p), r31.ctypes.data_as(C.c_void_p))
        return r61, r31

    def weighted(self):
        """residues of the last squaring's coefficients, scaled but still weighted: the input of the GPU carry sweep"""
        r61 = np.zeros(self.n, dtype=np.uint64); r31 = np.zeros(self.n, dtype=np.uint32)
        self.L.orcc_get_weighted(self.h, r61.ctypes.data_as(C.c_void_p), r31.ctypes.data_as(C.c_void_p))
        return r61, r31

    def words(self):
        """canonical little-endian 32-bit words of the residue (2^p - 1 -> 0)"""
        w = np.zeros((self.p + 31) // 32, dtype=np.uint32)
        self.L.orcc_get_words(self.h, w.ctypes.data_as(C.c_void_p), w.size)
        return w

    def value(self):
        return int.from_bytes(self.words().astype("<u4").tobytes(), "little")
