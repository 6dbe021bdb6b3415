#!/usr/bin/env python3
"""Generate tests/golden/big_p_pins.json: big-exponent pins independent of the oracle and the engine.

SURVEY.md 8c "Big-p pins": x0 = 3^(2^30) mod Mp (30 squarings of 3), then K more squarings, with the
res64, the low 2048 bits and a SHA-256 of the canonical little-endian word vector recorded per step, for
the BASELINE exponents C2/C3/C4.  Arithmetic: libgmp through ctypes (mpz_mul + shift/add reduction mod
2^p-1) -- nothing from the reference is imported or executed.  Run: python tests/golden/make_big_p_pins.py
"""
import ctypes
import ctypes.util
import hashlib
import json
import os
import sys

gmp = ctypes.CDLL(ctypes.util.find_library("gmp") or "libgmp.so.10")


class Mpz(ctypes.Structure):
    _fields_ = [("alloc", ctypes.c_int), ("size", ctypes.c_int), ("d", ctypes.c_void_p)]


def fn(name, *argtypes, restype=None):
    f = getattr(gmp, "__gmpz_" + name)
    f.argtypes, f.restype = list(argtypes), restype
    return f


P_ = ctypes.POINTER(Mpz)
init, clear = fn("init", P_), fn("clear", P_)
set_ui = fn("set_ui", P_, ctypes.c_ulong)
mul = fn("mul", P_, P_, P_)
add = fn("add", P_, P_, P_)
sub = fn("sub", P_, P_, P_)
cmp_ = fn("cmp", P_, P_, restype=ctypes.c_int)
tdiv_r_2exp = fn("tdiv_r_2exp", P_, P_, ctypes.c_ulong)
tdiv_q_2exp = fn("tdiv_q_2exp", P_, P_, ctypes.c_ulong)
mul_2exp = fn("mul_2exp", P_, P_, ctypes.c_ulong)
sub_ui = fn("sub_ui", P_, P_, ctypes.c_ulong)
export = fn("export", ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.c_size_t, ctypes.c_int,
            ctypes.c_size_t, P_, restype=ctypes.c_void_p)


def words_of(x, p):
    """canonical little-endian 32-bit words of x (0 <= x < 2^p-1), ceil(p/32) words"""
    nw = (p + 31) // 32
    buf = (ctypes.c_uint32 * (nw + 2))()
    cnt = ctypes.c_size_t(0)
    export(buf, ctypes.byref(cnt), -1, 4, -1, 0, ctypes.byref(x))
    return bytes(buf)[: 4 * nw]


def pins(p, start, count):
    x, t, hi, mp = Mpz(), Mpz(), Mpz(), Mpz()
    for z in (x, t, hi, mp):
        init(ctypes.byref(z))
    set_ui(ctypes.byref(mp), 1)
    mul_2exp(ctypes.byref(mp), ctypes.byref(mp), p)
    sub_ui(ctypes.byref(mp), ctypes.byref(mp), 1)
    set_ui(ctypes.byref(x), 3)
    out = []
    for it in range(1, start + count + 1):
        mul(ctypes.byref(t), ctypes.byref(x), ctypes.byref(x))
        tdiv_q_2exp(ctypes.byref(hi), ctypes.byref(t), p)           # t = hi * 2^p + lo  =>  hi + lo (mod 2^p-1)
        tdiv_r_2exp(ctypes.byref(t), ctypes.byref(t), p)
        add(ctypes.byref(x), ctypes.byref(t), ctypes.byref(hi))
        if cmp_(ctypes.byref(x), ctypes.byref(mp)) >= 0:
            sub(ctypes.byref(x), ctypes.byref(x), ctypes.byref(mp))
        if it >= start:
            w = words_of(x, p)
            out.append({"iteration": it, "res64": "%016X" % int.from_bytes(w[:8], "little"),
                        "low2048": w[:256][::-1].hex().upper(), "sha256_words": hashlib.sha256(w).hexdigest()})
            sys.stderr.write("p=%d it=%d res64=%s\n" % (p, it, out[-1]["res64"]))
    for z in (x, t, hi, mp):
        clear(ctypes.byref(z))
    return out


if __name__ == "__main__":
    doc = {"_source": "libgmp via ctypes (tests/golden/make_big_p_pins.py); x_0 = 3, x_{i+1} = x_i^2 mod 2^p-1; "
                      "words = canonical little-endian 32-bit words, low2048 = hex of the low 2048 bits (most significant first)",
           "start": 30, "pins": {}}
    for p, count in ((9815459, 12), (136279841, 6), (205271257, 6)):
        doc["pins"][str(p)] = pins(p, 30, count)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "big_p_pins.json")
    json.dump(doc, open(path, "w"), indent=1)
    print("wrote", path)
