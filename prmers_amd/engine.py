"""Python binding of libmi355_engine.so (ctypes over the C ABI in include/mi355_engine.h).

`Engine` mirrors the reference's `engine` register machine for the Marin path
(include/marin/engine.h:16-303: set / copy / square_mul / set_multiplicand / mul / sub / add /
sub_reg / get_mpz / set_mpz / digit / checkpoint), same names, same argument meaning, errors raised
as EngineError where the reference throws std::runtime_error.  There is no CPU fallback: without the
built HIP library or without a GPU, construction fails.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355_ENGINE_LIB overrides the library path (A/B builds), as AEVUM_ENGINE_LIB does for the reference's plugin
# (src/aevum/EngineAevum.cpp:193-223)
LIB_PATH = os.environ.get("MI355_ENGINE_LIB") or os.path.join(_HERE, "libmi355_engine.so")


class EngineError(RuntimeError):
    pass


_lib = None


def load_library():
    """dlopen the in-tree HIP library; raises if it has not been built (see __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, sz, u32, u64p = C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint64)
    dp = C.POINTER(C.c_double)
    sig = {
        "mi355_engine_version": (C.c_char_p, []),
        "mi355_engine_last_error": (C.c_char_p, []),
        "mi355_engine_resolve_fft": (C.c_int, [u32, C.c_char_p, C.c_char_p, sz]),
        "mi355_engine_create": (vp, [u32, sz, u32, C.c_int, C.c_char_p, C.c_char_p]),
        "mi355_engine_destroy": (None, [vp]),
        "mi355_engine_transform_size": (sz, [vp]),
        "mi355_engine_word_count": (sz, [vp]),
        "mi355_engine_sync": (C.c_int, [vp]),
        "mi355_engine_set_u32": (C.c_int, [vp, sz, u32]),
        "mi355_engine_set_words": (C.c_int, [vp, sz, vp, sz]),
        "mi355_engine_get_words": (C.c_int, [vp, sz, vp, sz]),
        "mi355_engine_copy": (C.c_int, [vp, sz, sz]),
        "mi355_engine_prepare": (C.c_int, [vp, sz, sz]),
        "mi355_engine_square_mul": (C.c_int, [vp, sz, u32]),
        "mi355_engine_mul": (C.c_int, [vp, sz, sz, u32]),
        "mi355_engine_add": (C.c_int, [vp, sz, sz]),
        "mi355_engine_sub_reg": (C.c_int, [vp, sz, sz]),
        "mi355_engine_sub_u32": (C.c_int, [vp, sz, u32]),
        "mi355_engine_equal": (C.c_int, [vp, sz, sz, C.POINTER(C.c_int)]),
        "mi355_engine_get_digits": (C.c_int, [vp, sz, vp, sz]),
        "mi355_engine_set_digits": (C.c_int, [vp, sz, vp, sz]),
        "mi355_engine_res64": (C.c_int, [vp, sz, u64p]),
        "mi355_engine_register_data_size": (sz, [vp]),
        "mi355_engine_get_data": (C.c_int, [vp, sz, vp, sz]),
        "mi355_engine_set_data": (C.c_int, [vp, sz, vp, sz]),
        "mi355_engine_checkpoint_size": (sz, [vp]),
        "mi355_engine_get_checkpoint": (C.c_int, [vp, vp, sz]),
        "mi355_engine_set_checkpoint": (C.c_int, [vp, vp, sz]),
        "mi355_engine_time_square_mul": (C.c_int, [vp, sz, u32, u32, sz, dp, dp, sz]),
        "mi355_engine_kernel_count": (sz, [vp]),
        "mi355_engine_kernel_name": (C.c_char_p, [vp, sz]),
        "mi355_engine_algorithmic_bytes": (sz, [vp]),
        "mi355_engine_selftest": (C.c_int, [sz]),
        "mi355_engine_addsub": (C.c_int, [vp, sz, sz, sz, sz]),
        "mi355_engine_addsub_copy": (C.c_int, [vp, sz, sz, sz, sz, sz, sz]),
        "mi355_engine_mul_add": (C.c_int, [vp, sz, sz, sz, u32]),
        "mi355_engine_square_mul_copy": (C.c_int, [vp, sz, sz, u32]),
        "mi355_engine_square_mul_n": (C.c_int, [vp, sz, u32, sz, u32]),
        "mi355_engine_mul_copy": (C.c_int, [vp, sz, sz, sz, u32]),
        "mi355_crt_carry": (C.c_int, [u32, sz, u32, u32, vp, vp, vp, vp, sz, dp]),
        "mi355_crt_transform_size": (sz, [u32, u32]),
        "mi355_engine_describe": (C.c_int, [vp, C.c_char_p, sz]),
        "mi355_crt_get_raw_digits": (C.c_int, [vp, sz, vp, sz, C.c_int]),
        "mi355_crt_set_raw_digits": (C.c_int, [vp, sz, vp, sz]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)   # AttributeError here = the library does not export what the header declares
        f.restype, f.argtypes = res, args
    _lib = L
    return L


EXPORTS = [
    "mi355_engine_version", "mi355_engine_last_error", "mi355_engine_resolve_fft", "mi355_engine_create",
    "mi355_engine_destroy", "mi355_engine_transform_size", "mi355_engine_word_count", "mi355_engine_sync",
    "mi355_engine_set_u32", "mi355_engine_set_words", "mi355_engine_get_words", "mi355_engine_copy",
    "mi355_engine_prepare", "mi355_engine_square_mul", "mi355_engine_mul", "mi355_engine_add",
    "mi355_engine_sub_reg", "mi355_engine_sub_u32", "mi355_engine_equal", "mi355_engine_get_digits",
    "mi355_engine_set_digits", "mi355_engine_res64", "mi355_engine_register_data_size", "mi355_engine_get_data",
    "mi355_engine_set_data", "mi355_engine_checkpoint_size", "mi355_engine_get_checkpoint",
    "mi355_engine_set_checkpoint", "mi355_engine_time_square_mul", "mi355_engine_kernel_count",
    "mi355_engine_kernel_name", "mi355_engine_algorithmic_bytes", "mi355_engine_selftest",
    "mi355_crt_carry", "mi355_crt_transform_size", "mi355_engine_describe", "mi355_crt_get_raw_digits", "mi355_crt_set_raw_digits",
    "mi355_engine_addsub", "mi355_engine_addsub_copy", "mi355_engine_mul_add", "mi355_engine_square_mul_copy", "mi355_engine_mul_copy", "mi355_engine_square_mul_n",
]


def resolve_plan(p, spec=None):
    """Transform plan text for exponent p (no GPU needed)."""
    L = load_library()
    buf = C.create_string_buffer(256)
    if not L.mi355_engine_resolve_fft(p, spec.encode() if spec else None, buf, 256):
        raise EngineError(L.mi355_engine_last_error().decode())
    return buf.value.decode()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """engine::create_gpu(p, reg_count, device, verbose) on an MI355X (include/marin/engine.h:301)."""

    def __init__(self, p, reg_count=8, device=0, verbose=False, plan=None):
        self.L = load_library()
        self.h = self.L.mi355_engine_create(p, reg_count, device, int(verbose), plan.encode() if plan else None, None)
        if not self.h:
            raise EngineError(self.L.mi355_engine_last_error().decode())
        self.p, self.reg_count = p, reg_count
        self.n = self.L.mi355_engine_transform_size(self.h)
        self.word_count = self.L.mi355_engine_word_count(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.mi355_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ok(self, rc):
        if not rc:
            raise EngineError(self.L.mi355_engine_last_error().decode())

    # --- engine.h surface ---
    def get_size(self): return self.n
    def sync(self): self._ok(self.L.mi355_engine_sync(self.h))
    def set(self, dst, a): self._ok(self.L.mi355_engine_set_u32(self.h, dst, a))
    def copy(self, dst, src): self._ok(self.L.mi355_engine_copy(self.h, dst, src))
    def square_mul(self, src, a=1): self._ok(self.L.mi355_engine_square_mul(self.h, src, a))
    def set_multiplicand(self, dst, src): self._ok(self.L.mi355_engine_prepare(self.h, dst, src))
    def mul(self, dst, src, a=1): self._ok(self.L.mi355_engine_mul(self.h, dst, src, a))
    def sub(self, src, a): self._ok(self.L.mi355_engine_sub_u32(self.h, src, a))
    def add(self, dst, src): self._ok(self.L.mi355_engine_add(self.h, dst, src))
    def sub_reg(self, dst, src): self._ok(self.L.mi355_engine_sub_reg(self.h, dst, src))

    # fused variants (engine.h:65-131): one sweep each
    def addsub(self, sum_out, diff_out, a, b): self._ok(self.L.mi355_engine_addsub(self.h, sum_out, diff_out, a, b))
    def addsub_copy(self, s, d, s_copy, d_copy, a, b): self._ok(self.L.mi355_engine_addsub_copy(self.h, s, d, s_copy, d_copy, a, b))
    def mul_add(self, dst, mul_src, add_src, a=1): self._ok(self.L.mi355_engine_mul_add(self.h, dst, mul_src, add_src, a))
    def square_mul_copy(self, src, dst_copy, a=1): self._ok(self.L.mi355_engine_square_mul_copy(self.h, src, dst_copy, a))
    def square_mul_n(self, src, count, a=1, sub=0):
        """count x { src = src^2 * a; src -= sub }: one cooperative launch on the small transforms."""
        self._ok(self.L.mi355_engine_square_mul_n(self.h, src, a, count, sub))
    def mul_copy(self, dst, src, dst_copy, a=1): self._ok(self.L.mi355_engine_mul_copy(self.h, dst, src, dst_copy, a))

    def is_equal(self, lhs, rhs):
        out = C.c_int(0)
        self._ok(self.L.mi355_engine_equal(self.h, lhs, rhs, C.byref(out)))
        return bool(out.value)

    def pow(self, dst, src, e):
        """dst = src^e, src is erased (engine.h:160-170)."""
        self.set_multiplicand(src, src)
        self.set(dst, 1)
        if e == 0:
            return
        for i in range(int(e).bit_length() - 1, -1, -1):
            self.square_mul(dst)
            if (e >> i) & 1:
                self.mul(dst, src)

    # --- digit / word I/O ---
    def digits(self, src):
        """engine::digit (engine.h:234-296): n values `digit | width << 32`."""
        d = np.zeros(self.n, dtype=np.uint64)
        self._ok(self.L.mi355_engine_get_digits(self.h, src, _ptr(d), self.n))
        return d

    def set_digits(self, dst, d):
        d = np.ascontiguousarray(d, dtype=np.uint64)
        self._ok(self.L.mi355_engine_set_digits(self.h, dst, _ptr(d), d.size))

    def res64(self, src):
        out = C.c_uint64(0)
        self._ok(self.L.mi355_engine_res64(self.h, src, C.byref(out)))
        return out.value

    def words(self, src):
        w = np.zeros(self.word_count, dtype=np.uint32)
        self._ok(self.L.mi355_engine_get_words(self.h, src, _ptr(w), w.size))
        return w

    def set_words(self, dst, w):
        w = np.ascontiguousarray(w, dtype=np.uint32)
        self._ok(self.L.mi355_engine_set_words(self.h, dst, _ptr(w), w.size))

    def get_int(self, src):
        """get_mpz (engine.h:173-203) as a Python int in [0, 2^p-1)."""
        return int.from_bytes(self.words(src).astype("<u4").tobytes(), "little")

    def set_int(self, dst, v):
        """set_mpz (engine.h:206-232); v is reduced mod 2^p-1 first."""
        v = int(v) % ((1 << self.p) - 1)
        self.set_words(dst, np.frombuffer(v.to_bytes(self.word_count * 4, "little"), dtype="<u4"))

    # --- raw images / checkpoints ---
    def get_register_data_size(self): return self.L.mi355_engine_register_data_size(self.h)

    def get_data(self, src):
        buf = np.zeros(self.get_register_data_size(), dtype=np.uint8)
        self._ok(self.L.mi355_engine_get_data(self.h, src, _ptr(buf), buf.size))
        return buf

    def set_data(self, dst, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        rc = self.L.mi355_engine_set_data(self.h, dst, _ptr(buf), buf.size)
        return bool(rc)

    def get_checkpoint_size(self): return self.L.mi355_engine_checkpoint_size(self.h)

    def get_checkpoint(self):
        buf = np.zeros(self.get_checkpoint_size(), dtype=np.uint8)
        self._ok(self.L.mi355_engine_get_checkpoint(self.h, _ptr(buf), buf.size))
        return buf

    def set_checkpoint(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        return bool(self.L.mi355_engine_set_checkpoint(self.h, _ptr(buf), buf.size))

    # --- measurement ---
    def kernel_names(self):
        return [self.L.mi355_engine_kernel_name(self.h, k).decode() for k in range(self.L.mi355_engine_kernel_count(self.h))]

    def time_square_mul(self, reg, iters, a=1, sub=0, per_kernel=False):
        """(total_ms, {kernel: avg_ms}) for `iters` back-to-back squarings, HIP events on the engine stream."""
        total = C.c_double(0)
        k = self.L.mi355_engine_kernel_count(self.h)
        ks = (C.c_double * k)()
        self._ok(self.L.mi355_engine_time_square_mul(self.h, reg, a, sub, iters, C.byref(total),
                                                     ks if per_kernel else None, k if per_kernel else 0))
        return total.value, (dict(zip(self.kernel_names(), list(ks))) if per_kernel else {})

    def algorithmic_bytes(self): return self.L.mi355_engine_algorithmic_bytes(self.h)

    def describe(self):
        """plan text of this engine (kernel set included)"""
        buf = C.create_string_buffer(256)
        self._ok(self.L.mi355_engine_describe(self.h, buf, 256))
        return buf.value.decode()



class CrtEngine(Engine):
    """The GF(M61^2) x GF(M31^2) engine with a prime-factor axis of radix 1, 3 or 9 (the reference's Aevum plugin,
    third_party/aevum/src/EngineApi.h:28-59) -- the same register machine as `Engine`, selected through the fft_spec "crt:odd[:words=N]".
    Digits are plain u64 values here (widths reach 39 bits).  No CPU fallback."""

    def __init__(self, p, odd=1, n=0, device=0, plan=None, reg_count=4):
        """odd = None / "auto": radix 9, 3 or none by the reference's stock / PFA size-ratio gates (README.md:888-926)"""
        spec = ("crt:auto" if odd in (None, "auto") else "crt:%d" % odd) + (":words=%d" % n if n else "") + (":" + plan if plan else "")
        Engine.__init__(self, p, reg_count, device, False, spec)
        self.odd = int(self.describe().split("odd=")[1].split(":")[0]) if odd in (None, "auto") else odd

    def raw_digits(self, src=0):
        """the engine's own digits: plain values in base 2^width_j, canonical (widths reach 39 bits)"""
        d = np.zeros(self.n, dtype=np.uint64)
        self._ok(self.L.mi355_crt_get_raw_digits(self.h, src, _ptr(d), self.n, 1))
        return d

    def set_digits(self, dst, d):
        """plain digit values (the engine's own base 2^width_j digits)"""
        d = np.ascontiguousarray(d, dtype=np.uint64)
        self._ok(self.L.mi355_crt_set_raw_digits(self.h, dst, _ptr(d), d.size))

    def digits(self, src=0):
        """value | width << 32 like engine::get (engine.h:24), for callers that cut a residue into words themselves (prp.py): the canonical
        residue in 32-bit pieces -- this family's own digits do not always fit that encoding"""
        w = self.words(src).astype(np.uint64)
        width = np.full(w.size, 32, dtype=np.uint64)
        if self.p % 32:
            width[-1] = self.p % 32
        return w | (width << np.uint64(32))

    def time_square_mul(self, reg, iters, a=1, sub=0, per_kernel=True):
        """(total_ms of `iters` squarings, {stage: average ms}); every iteration is bracketed by events here"""
        return Engine.time_square_mul(self, reg, iters, a, sub, per_kernel)
