#!/usr/bin/env python3
"""LDS bank-conflict census of the exchanges of kernels_v5.hip (8-byte slots: ds_write_b64 in four groups of 16 lanes on 16 slot classes, ds_read_b64 in
two groups of 32 lanes on 32) for the slot maps i ^ ((i >> S) & 31), both column shapes (J = 0: 1280 x 4, J = 1: 2560 x 2).  Ideal: stores 4, loads 2
cycles per wave-instruction.  The S values of kernels_v5.hip Shape<J> come from this table.  No GPU needed.  usage: tools/lds_census5.py"""
from collections import Counter
def cyc(slots, groups, classes):
    tot = 0
    for g in groups:
        c, seen = Counter(), set()
        for l in g:
            a = slots[l]
            if a is None: continue
            if a not in seen:
                seen.add(a); c[a % classes] += 1
        tot += max(c.values()) if c else 0
    return tot
WG = [list(range(16*g, 16*g+16)) for g in range(4)]
RG = [list(range(32*g, 32*g+32)) for g in range(2)]
def pats(J):
    C = 4 >> J; LC = 2 - J; L = 256 << J; R1 = 4 << J
    def A_w(t, k): return (t + 640 * (k // C)) * C + (k % C)
    def A_r(t, k):   # prime-factor input map: i1 = (L d0 + 5 r) mod M1
        if t >= 512: return None
        g = t + 512 * (k // 5); return ((L * (k % 5) + 5 * (g // C)) % (5 * L)) * C + (g % C)
    def A_o(t, k):   # radix-5 outputs, slot (k0 L + r) C + c
        if t >= 512: return None
        g = t + 512 * (k // 5); return 1024 * (k % 5) + g
    def dec1(t): return t & 1, (t >> 1) & 7, (t >> 4) & 7, t >> 7
    def B1_r(t, k):
        chi, e3, e2, k0 = dec1(t)
        e1 = k if J else k >> 1; cc = chi if J else 2 * chi + (k & 1)
        return (k0 * L + 64 * e1 + 8 * e2 + e3) * C + cc
    def dec2(t): return t & (C - 1), (t >> LC) & 7, (t >> (LC + 3)) & (R1 - 1), t >> 7
    def B2_r(t, k):
        c, f3, f1, f0 = dec2(t); return (f0 * L + 64 * f1 + 8 * k + f3) * C + c
    def B3_r(t, k):
        c, g2, f1, f0 = dec2(t); return (f0 * L + 64 * f1 + 8 * g2 + k) * C + c
    return {"A": (A_w, 8, A_r, 10), "B1": (A_o, 10, B1_r, 8), "B2": (B1_r, 8, B2_r, 8), "B3": (B2_r, 8, B3_r, 8)}
for J in (0, 1):
    print("J =", J)
    for name, (fw, nw, fr, nr) in pats(J).items():
        res = []
        for S in range(0, 9):
            ph = (lambda i: i ^ ((i >> S) & 31)) if S else (lambda i: i)
            w = r = nwi = nri = 0
            for wave in range(10):
                for k in range(nw):
                    sl = [None if fw(64 * wave + l, k) is None else ph(fw(64 * wave + l, k)) for l in range(64)]
                    if any(s is not None for s in sl): w += cyc(sl, WG, 16); nwi += 1
                for k in range(nr):
                    sl = [None if fr(64 * wave + l, k) is None else ph(fr(64 * wave + l, k)) for l in range(64)]
                    if any(s is not None for s in sl): r += cyc(sl, RG, 32); nri += 1
            res.append("S=%d w%.1f r%.1f" % (S, w / nwi, r / nri))
        print(" ", name, " | ".join(res))
