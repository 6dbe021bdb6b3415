"""Pure-Python model of the MI355X kernel pipeline (index math only; tiny sizes).

Mirrors prmers_amd/csrc/kernels.hip stage by stage -- front (weight + column DFT + twiddle), middle
(row DFT + pointwise + inverse row DFT), back (twiddle + inverse column DFT + unweight + run carry),
carry fix -- so index conventions can be checked against Python integers before touching a GPU.
Development aid; not imported by the product or the tests.
"""
import sys

P = 2**64 - 2**32 + 1


def transform_size(p):
    log2_n, log2_n5 = 1, 2
    while True:
        log2_n += 1
        w = p >> log2_n
        if not ((w + 1) * 2 + log2_n >= 64):
            break
    while True:
        log2_n5 += 1
        w = p // (5 << log2_n5)
        if not ((w + 1) * 2 + (log2_n5 + 2.4) >= 64):
            break
    n2 = (1 << log2_n) if log2_n <= 26 else 1 << 62
    n5 = (5 << log2_n5) if log2_n5 <= 26 else 1 << 62
    return min(n2, n5)


def brev(i, L):
    r = 0
    k = L
    while k > 1:
        r = (r << 1) | (i & 1)
        i >>= 1
        k >>= 1
    return r


class Plan:
    def __init__(self, p, M2=None, C=None):
        self.p = p
        self.n = n = transform_size(p)
        self.m = m = n // 2
        self.r5 = 5 if m % 5 == 0 else 1
        k = (m // self.r5).bit_length() - 1
        if M2 is None:
            M2 = 1 << min(k, 12)
        self.M2 = M2
        self.M1 = m // M2
        assert self.M1 * M2 == m and self.M1 % self.r5 == 0
        self.L1 = self.M1 // self.r5
        if C is None:
            C = max(1, min(M2, 4096 // self.M1))
        self.C = C
        assert M2 % C == 0
        self.om = pow(7, (P - 1) // m, P)          # omega_m
        r = pow(554, (P - 1) // 192 // n, P)        # 2^(1/n)
        q, t = divmod(p, n)
        self.q, self.t = q, t
        # factored weights: s_j = p*j mod n = SA[i1] + SB[2*i2+b] (mod n)
        self.SA = [(2 * M2 * p * i1) % n for i1 in range(self.M1)]
        self.SB = [(p * x) % n for x in range(2 * M2)]
        self.TA = [pow(r, (n - s) % n, P) for s in self.SA]
        self.TB = [pow(r, (n - s) % n, P) for s in self.SB]
        inv_m = pow(m, P - 2, P)
        self.TAi = [pow(x, P - 2, P) * inv_m % P for x in self.TA]
        self.TBi = [pow(x, P - 2, P) for x in self.TB]
        self.I4 = pow(self.om, m // 4, P) if m % 4 == 0 else None

    def w(self, e):
        return pow(self.om, e % self.m, P)

    def width_wrap(self, i1, x):
        """digit width and weight-wrap flag for digit (i1, x = 2*i2+b)."""
        n = self.n
        sa, sb = self.SA[i1], self.SB[x]
        s = sa + sb
        if s >= n:
            s -= n
        width = self.q + (1 if s + self.t > 0 else 0) + (1 if s + self.t > n else 0) - (1 if s > 0 else 0)
        wrap = sa > 0 and sb > 0 and sa + sb <= n
        return width, wrap

    def pos(self, j):
        """memory position (u32 index) of natural digit j in the tile-major register layout."""
        i, b = j >> 1, j & 1
        i1, i2 = divmod(i, self.M2)
        T, c = divmod(i2, self.C)
        return ((T * self.M1 + i1) * self.C + c) * 2 + b

    def freq1(self, pos):
        """column-DFT output position -> k1."""
        blk, q = divmod(pos, self.L1)
        return blk + self.r5 * brev(q, self.L1)


def dif_pow2(x, base, L, stride, root_of, inverse=False):
    """in-place radix-4/2 DIF (or mirrored DIT inverse) on x[base + stride*i], i < L.
    root_of(len, t) = omega_len^t (or its inverse)."""
    I4 = root_of(4, 1) if L >= 4 else None
    lens = []
    ln = L
    while ln >= 4:
        lens.append((4, ln))
        ln >>= 2
    if ln == 2:
        lens.append((2, 2))
    if inverse:
        lens = lens[::-1]
    for radix, ln in lens:
        if radix == 2:
            for blk in range(L // 2):
                i0 = base + stride * (2 * blk)
                i1 = i0 + stride
                u, v = x[i0], x[i1]
                x[i0], x[i1] = (u + v) % P, (u - v) % P
            continue
        q = ln // 4
        for bi in range(L // 4):
            blk, t = divmod(bi, q)
            idx = [base + stride * (blk * ln + t + k * q) for k in range(4)]
            w1, w2, w3 = root_of(ln, t), root_of(ln, 2 * t), root_of(ln, 3 * t)
            if not inverse:
                x0, x1, x2, x3 = (x[i] for i in idx)
                a, b, c, d = (x0 + x2) % P, (x1 + x3) % P, (x0 - x2) % P, (x1 - x3) * I4 % P
                x[idx[0]] = (a + b) % P
                x[idx[1]] = (a - b) * w2 % P
                x[idx[2]] = (c + d) * w1 % P
                x[idx[3]] = (c - d) * w3 % P
            else:
                y0, y1, y2, y3 = (x[i] for i in idx)
                y1w = y1 * w2 % P
                A, B = (y0 + y1w) % P, (y0 - y1w) % P
                y2w, y3w = y2 * w1 % P, y3 * w3 % P
                Cc, D = (y2w + y3w) % P, (y2w - y3w) * I4 % P
                x[idx[0]] = (A + Cc) % P
                x[idx[2]] = (A - Cc) % P
                x[idx[1]] = (B + D) % P
                x[idx[3]] = (B - D) % P


def front(pl, digits):
    """digits: tile-major u32 list -> W (list of [s0, s1] pairs, row pos * M2 + i2)."""
    M1, M2, C, m = pl.M1, pl.M2, pl.C, pl.m
    W = [None] * m
    for T in range(M2 // C):
        X = [[0, 0] for _ in range(M1 * C)]
        for i1 in range(M1):
            for c in range(C):
                i2 = T * C + c
                for b in range(2):
                    d = digits[((T * M1 + i1) * C + c) * 2 + b]
                    _, wrap = pl.width_wrap(i1, 2 * i2 + b)
                    v = d * pl.TA[i1] % P
                    if wrap:
                        v = v * pow(2, P - 2, P) % P
                    X[i1 * C + c][b] = v
        for b in range(2):
            plane = [X[e][b] for e in range(M1 * C)]
            for c in range(C):
                col_dft(pl, plane, c, C, inverse=False)
            for e in range(M1 * C):
                X[e][b] = plane[e]
        for pos in range(M1):
            k1 = pl.freq1(pos)
            for c in range(C):
                i2 = T * C + c
                tw = pl.w(i2 * k1)
                W[pos * M2 + i2] = [X[pos * C + c][b] * tw % P * pl.TB[2 * i2 + b] % P for b in range(2)]
    return W


def col_dft(pl, plane, c, C, inverse):
    M1, r5, L1 = pl.M1, pl.r5, pl.L1
    sgn = -1 if inverse else 1

    def root(ln, t):
        return pow(pl.om, (sgn * t * (pl.m // ln)) % pl.m, P)

    def r5pass():
        w5 = [pow(pl.om, (sgn * k * (pl.m // 5)) % pl.m, P) for k in range(5)]
        for t in range(L1):
            idx = [(L1 * r + t) * C + c for r in range(5)]
            xs = [plane[i] for i in idx]
            if not inverse:
                for k in range(5):
                    s = sum(xs[r] * w5[(r * k) % 5] for r in range(5)) % P
                    plane[idx[k]] = s * pow(pl.om, (t * k * (pl.m // M1)) % pl.m, P) % P
            else:
                xs = [xs[k] * pow(pl.om, (-(t * k) * (pl.m // M1)) % pl.m, P) % P for k in range(5)]
                for r in range(5):
                    plane[idx[r]] = sum(xs[k] * w5[(r * k) % 5] for k in range(5)) % P

    if r5 == 5 and not inverse:
        r5pass()
    if L1 > 1:
        for blk in range(r5):
            dif_pow2(plane, (blk * L1) * C + c, L1, C, root, inverse)
    if r5 == 5 and inverse:
        r5pass()


def middle(pl, W, Y=None):
    """row DFT + pointwise square (or multiply by image Y) + inverse row DFT, in place.
    Y is None -> square; Y == 'fwd' -> forward only (multiplicand image)."""
    M1, M2, m = pl.M1, pl.M2, pl.m
    for pos in range(M1):
        k1 = pl.freq1(pos)
        for b in range(2):
            plane = [W[pos * M2 + i][b] for i in range(M2)]
            dif_pow2(plane, 0, M2, 1, lambda ln, t: pow(pl.om, (t * (m // ln)) % m, P), False)
            for i in range(M2):
                W[pos * M2 + i][b] = plane[i]
        if isinstance(Y, str):
            continue
        for i in range(M2):
            k = k1 + M1 * brev(i, M2)
            rho = pl.w(k)
            u0, u1 = W[pos * M2 + i]
            if Y is None:
                W[pos * M2 + i] = [(u0 * u0 + rho * u1 * u1) % P, 2 * u0 * u1 % P]
            else:
                y0, y1 = Y[pos * M2 + i]
                W[pos * M2 + i] = [(u0 * y0 + rho * u1 * y1) % P, (u0 * y1 + u1 * y0) % P]
        for b in range(2):
            plane = [W[pos * M2 + i][b] for i in range(M2)]
            dif_pow2(plane, 0, M2, 1, lambda ln, t: pow(pl.om, (-t * (m // ln)) % m, P), True)
            for i in range(M2):
                W[pos * M2 + i][b] = plane[i]


def back(pl, W, a=1):
    """-> (digits tile-major, carry-out per run cbuf[T][i1])"""
    M1, M2, C, m = pl.M1, pl.M2, pl.C, pl.m
    digits = [0] * pl.n
    cbuf = [[0] * M1 for _ in range(M2 // C)]
    for T in range(M2 // C):
        X = [[0, 0] for _ in range(M1 * C)]
        for pos in range(M1):
            k1 = pl.freq1(pos)
            for c in range(C):
                i2 = T * C + c
                tw = pl.w(-(i2 * k1))
                X[pos * C + c] = [W[pos * M2 + i2][b] * tw % P * pl.TBi[2 * i2 + b] % P for b in range(2)]
        for b in range(2):
            plane = [X[e][b] for e in range(M1 * C)]
            for c in range(C):
                col_dft(pl, plane, c, C, inverse=True)
            for e in range(M1 * C):
                X[e][b] = plane[e]
        for i1 in range(M1):
            carry = 0
            for c in range(C):
                i2 = T * C + c
                for b in range(2):
                    width, wrap = pl.width_wrap(i1, 2 * i2 + b)
                    u = X[i1 * C + c][b] * pl.TAi[i1] % P
                    if wrap:
                        u = 2 * u % P
                    v = u * a + carry
                    digits[((T * M1 + i1) * C + c) * 2 + b] = v & ((1 << width) - 1)
                    carry = v >> width
            cbuf[T][i1] = carry
    return digits, cbuf


def carry_fix(pl, digits, cbuf):
    M1, M2, C = pl.M1, pl.M2, pl.C
    NT = M2 // C
    for T in range(NT):
        for i1 in range(M1):
            # previous run in digit order
            if T > 0:
                cin = cbuf[T - 1][i1]
            else:
                cin = cbuf[NT - 1][(i1 - 1) % M1]
            base = (T * M1 + i1) * C * 2
            for k in range(2 * C):
                c, b = divmod(k, 2)
                width, _ = pl.width_wrap(i1, 2 * (T * C + c) + b)
                if k == 2 * C - 1:
                    digits[base + k] += cin
                else:
                    v = digits[base + k] + cin
                    digits[base + k] = v & ((1 << width) - 1)
                    cin = v >> width


def value(pl, digits):
    v, s = 0, 0
    for j in range(pl.n):
        i1 = (j >> 1) // pl.M2
        x = 2 * ((j >> 1) % pl.M2) + (j & 1)
        width, _ = pl.width_wrap(i1, x)
        v += digits[pl.pos(j)] << s
        s += width
    assert s == pl.p
    return v % ((1 << pl.p) - 1)


def run(p, M2=None, C=None, iters=12):
    pl = Plan(p, M2, C)
    Mp = (1 << p) - 1
    digits = [0] * pl.n
    digits[0] = 3
    x = 3
    for it in range(iters):
        W = front(pl, digits)
        middle(pl, W)
        digits, cbuf = back(pl, W, 1)
        carry_fix(pl, digits, cbuf)
        x = x * x % Mp
        got = value(pl, digits)
        assert got == x, (p, M2, C, it)
    return pl


if __name__ == "__main__":
    for p, M2, C in [(127, None, None), (127, 2, 1), (127, 2, 2), (521, None, None), (521, 4, 2), (521, 8, 2), (521, 2, 2),
                     (933, None, None), (933, 4, 2), (933, 2, 2), (1801, 8, 4), (1801, 4, 4), (3997, 16, 4), (9941, 16, 4), (9941, 64, 8)]:
        pl = run(p, M2, C)
        print("ok p=%d n=%d M1=%d M2=%d C=%d r5=%d" % (p, pl.n, pl.M1, pl.M2, pl.C, pl.r5))
        sys.stdout.flush()
