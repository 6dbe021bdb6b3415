"""ctypes wrapper around oracle/_build/liboracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference's Marin IBDWT path (oracle/oracle.h).  Nothing in
prmers_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
REF_HOST = os.path.join(ORACLE_DIR, "_ref", "ref_host")


def build(force=False):
    """Compile the oracle (and oracle/_ref when the reference tree is present)."""
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
            os.path.join(ORACLE_DIR, "oracle.c")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "all"])
    if os.path.isdir("/root/reference/include/marin") and (force or not os.path.exists(REF_HOST)):
        subprocess.call(["make", "-C", ORACLE_DIR, "-s", "ref"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        # never oversubscribe: a GPU box exposes the host's cores but only a share of them is ours
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        L = C.CDLL(LIB_PATH)
        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 1
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads(int(os.environ.get("ORACLE_THREADS", max(1, min(ncpu, 16)))))
        u64, u32, sz, vp = C.c_uint64, C.c_uint32, C.c_size_t, C.c_void_p
        sig = {
            "orc_mod_add": (u64, [u64, u64]), "orc_mod_sub": (u64, [u64, u64]), "orc_mod_mul": (u64, [u64, u64]),
            "orc_mod_pow": (u64, [u64, u64]), "orc_mod_invert": (u64, [u64]),
            "orc_transform_size": (sz, [u32]),
            "orc_create": (vp, [u32, sz]), "orc_destroy": (None, [vp]), "orc_size": (sz, [vp]),
            "orc_exponent": (u32, [vp]), "orc_widths": (None, [vp, vp]), "orc_weights": (None, [vp, vp, vp]),
            "orc_threads": (C.c_int, []),
            "orc_set_u32": (None, [vp, sz, u32]), "orc_copy": (None, [vp, sz, sz]),
            "orc_square_mul": (None, [vp, sz, u32]), "orc_set_multiplicand": (None, [vp, sz, sz]),
            "orc_mul": (None, [vp, sz, sz, u32]), "orc_sub_u32": (None, [vp, sz, u32]),
            "orc_add": (None, [vp, sz, sz]), "orc_sub_reg": (None, [vp, sz, sz]),
            "orc_get_digits": (None, [vp, sz, vp]), "orc_set_digits": (None, [vp, sz, vp]),
            "orc_get_raw": (None, [vp, sz, vp]), "orc_set_raw": (None, [vp, sz, vp]),
            "orc_digits_res64": (u64, [vp, sz]), "orc_digits_equal_to": (C.c_int, [vp, sz, u64]),
            "orc_digits_equal_to_Mp": (C.c_int, [vp, sz]),
            "orc_word_count": (sz, [u32]), "orc_pack_words": (None, [vp, sz, u32, vp]),
            "orc_prp3_div9": (None, [u32, vp, sz]), "orc_format_res64": (None, [vp, sz, C.c_char_p]),
            "orc_format_res2048": (None, [vp, sz, C.c_char_p]),
            "orc_get_words": (None, [vp, sz, vp, sz]), "orc_set_words": (None, [vp, sz, vp, sz]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One oracle engine: the reference's register machine on the CPU."""

    def __init__(self, p, regs=8):
        self.L = lib()
        self.h = self.L.orc_create(p, regs)
        if not self.h:
            raise RuntimeError("orc_create failed")
        self.p, self.regs = p, regs
        self.n = self.L.orc_size(self.h)

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def widths(self):
        w = np.zeros(self.n, dtype=np.uint8)
        self.L.orc_widths(self.h, _ptr(w))
        return w

    def weights(self):
        w = np.zeros(self.n, dtype=np.uint64)
        wi = np.zeros(self.n, dtype=np.uint64)
        self.L.orc_weights(self.h, _ptr(w), _ptr(wi))
        return w, wi

    def set(self, r, a): self.L.orc_set_u32(self.h, r, a)
    def copy(self, d, s): self.L.orc_copy(self.h, d, s)
    def square_mul(self, r, a=1): self.L.orc_square_mul(self.h, r, a)
    def set_multiplicand(self, d, s): self.L.orc_set_multiplicand(self.h, d, s)
    def mul(self, d, s, a=1): self.L.orc_mul(self.h, d, s, a)
    def sub(self, r, a): self.L.orc_sub_u32(self.h, r, a)
    def add(self, d, s): self.L.orc_add(self.h, d, s)
    def sub_reg(self, d, s): self.L.orc_sub_reg(self.h, d, s)

    def digits(self, r):
        d = np.zeros(self.n, dtype=np.uint64)
        self.L.orc_get_digits(self.h, r, _ptr(d))
        return d

    def set_digits(self, r, d):
        d = np.ascontiguousarray(d, dtype=np.uint64)
        assert d.size == self.n
        self.L.orc_set_digits(self.h, r, _ptr(d))

    def raw(self, r):
        x = np.zeros(self.n, dtype=np.uint64)
        self.L.orc_get_raw(self.h, r, _ptr(x))
        return x

    def res64(self, r):
        d = self.digits(r)
        return self.L.orc_digits_res64(_ptr(d), self.n)

    def words(self, r):
        wc = self.L.orc_word_count(self.p)
        w = np.zeros(wc, dtype=np.uint32)
        self.L.orc_get_words(self.h, r, _ptr(w), wc)
        return w

    def set_words(self, r, w):
        w = np.ascontiguousarray(w, dtype=np.uint32)
        self.L.orc_set_words(self.h, r, _ptr(w), w.size)

    def value(self, r):
        return words_to_int(self.words(r))

    def set_value(self, r, v):
        self.set_words(r, int_to_words(v % ((1 << self.p) - 1), self.p))


def mers_reduce(x, p):
    """x mod 2^p-1 by shift-and-add (fast for the big-integer cross-checks)."""
    M = (1 << p) - 1
    while x > M:
        x = (x & M) + (x >> p)
    return 0 if x == M else x


def words_to_int(w):
    return int.from_bytes(np.ascontiguousarray(w, dtype="<u4").tobytes(), "little")


def int_to_words(v, p):
    wc = (p + 31) // 32
    return np.frombuffer(int(v).to_bytes(wc * 4, "little"), dtype="<u4").copy()


def digits_to_int(d):
    """Integer value of an encoded digit vector (value | width << 32), not reduced."""
    v, s = 0, 0
    for x in np.asarray(d, dtype=np.uint64).tolist():
        v += (x & 0xFFFFFFFF) << s
        s += x >> 32
    return v


def pack_words(d, p):
    L = lib()
    d = np.ascontiguousarray(d, dtype=np.uint64)
    wc = L.orc_word_count(p)
    w = np.zeros(wc, dtype=np.uint32)
    L.orc_pack_words(_ptr(d), d.size, p, _ptr(w))
    return w


def prp_type1_hex(d, p):
    """(res64, res2048) hex of the type-1 PRP residue, as RunPrpOrLlMarin.cpp:446-462 prints them."""
    L = lib()
    w = pack_words(d, p)
    L.orc_prp3_div9(p, _ptr(w), w.size)
    b64 = C.create_string_buffer(17)
    b2048 = C.create_string_buffer(513)
    L.orc_format_res64(_ptr(w), w.size, b64)
    L.orc_format_res2048(_ptr(w), w.size, b2048)
    return b64.value.decode(), b2048.value.decode()


class OracleEngine:
    """The oracle behind the interface of prmers_amd.Engine, so that host-side callers (prmers_amd.prp)
    can be tested without a GPU.  Test infrastructure only."""

    def __init__(self, p, reg_count=8):
        self.o = Oracle(p, reg_count)
        self.p, self.n, self.reg_count = p, self.o.n, reg_count
        self.word_count = (p + 31) // 32

    def close(self): self.o.close()
    def __enter__(self): return self
    def __exit__(self, *a): self.close()
    def get_size(self): return self.n
    def sync(self): pass
    def set(self, dst, a): self.o.set(dst, a)
    def copy(self, dst, src): self.o.copy(dst, src)
    def square_mul(self, src, a=1): self.o.square_mul(src, a)
    def square_mul_n(self, src, count, a=1, sub=0):   # the loop Engine.square_mul_n stands for (prp.py batches plain iterations through it)
        for _ in range(count):
            self.square_mul(src, a)
            if sub:
                self.sub(src, sub)
    def set_multiplicand(self, dst, src): self.o.set_multiplicand(dst, src)
    def mul(self, dst, src, a=1): self.o.mul(dst, src, a)
    def sub(self, src, a): self.o.sub(src, a)
    def add(self, dst, src): self.o.add(dst, src)
    def sub_reg(self, dst, src): self.o.sub_reg(dst, src)
    def digits(self, src): return self.o.digits(src)
    def res64(self, src): return self.o.res64(src)
    def get_int(self, src): return self.o.value(src)
    def set_int(self, dst, v): self.o.set_value(dst, v)
    def is_equal(self, a, b): return self.o.value(a) == self.o.value(b)
    def get_checkpoint_size(self): return self.reg_count * self.n * 8

    def get_checkpoint(self):
        return np.concatenate([self.o.raw(r) for r in range(self.reg_count)]).view(np.uint8)

    def set_checkpoint(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        if buf.size != self.get_checkpoint_size():
            return False
        x = buf.view(np.uint64)
        for r in range(self.reg_count):
            part = np.ascontiguousarray(x[r * self.n:(r + 1) * self.n])
            self.o.L.orc_set_raw(self.o.h, r, _ptr(part))
        return True
