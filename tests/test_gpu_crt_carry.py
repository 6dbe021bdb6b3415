"""First HIP kernel of the second field family: the fused unweight + Garner + carry sweep over GF(M61^2) x GF(M31^2)
(prmers_amd/csrc/crt_carry.hip; reference third_party/aevum/src/cl/carry.cl:506-588) through the C ABI, against the CRT oracle
(oracle/oracle_crt.c).  Needs a real MI355X:  python -m pytest tests -m gpu"""
import ctypes as C

import numpy as np
import pytest

import orc_crt

pytestmark = pytest.mark.gpu


def crt_carry(p, n, odd, a, r61, r31, timed=False):
    from prmers_amd.engine import load_library, EngineError
    L = load_library()
    digits = np.zeros(n, dtype=np.uint64)
    residual = np.zeros((n + 7) // 8, dtype=np.uint64)
    ms = C.c_double(0)
    ok = L.mi355_crt_carry(p, n, odd, a, r61.ctypes.data_as(C.c_void_p), r31.ctypes.data_as(C.c_void_p), digits.ctypes.data_as(C.c_void_p),
                           residual.ctypes.data_as(C.c_void_p), 0, C.byref(ms) if timed else None)
    if not ok:
        raise EngineError(L.mi355_engine_last_error().decode())
    return digits, residual, ms.value


def finish(digits, residual, widths):
    """the last carries (a unit here and there) in front of the following run, then the strong carry with wrap-around"""
    d = digits.astype(object)
    n = len(d)
    for run, c in enumerate(residual):
        if c:
            d[((run + 1) * 8) % n] += int(c)
    while True:
        over = [j for j in range(n) if d[j] >> int(widths[j])]
        if not over:
            return np.array(d, dtype=np.uint64)
        for j in over:
            c = d[j] >> int(widths[j])
            d[j] &= (1 << int(widths[j])) - 1
            d[(j + 1) % n] += c


@pytest.mark.parametrize("p,odd,a", [(521, 1, 1), (1279, 3, 1), (9941, 9, 3), (11213, 9, 1), (86243, 9, 1), (216091, 3, 3), (1257787, 9, 1), (3021377, 1, 1)])
def test_crt_carry_matches_the_oracle(p, odd, a):
    o = orc_crt.OracleCrt(p, odd)
    rng = np.random.default_rng(p)
    w = o.widths().astype(np.uint64)
    o.set_digits(rng.integers(0, 1 << 62, o.n, dtype=np.uint64) & ((np.uint64(1) << w) - np.uint64(1)))
    for _ in range(2):
        o.square_mul(a)
        r61, r31 = o.weighted()
        digits, residual, _ = crt_carry(p, o.n, odd, a, r61, r31)
        assert int(residual.max()) <= 8
        assert np.array_equal(finish(digits, residual, w), o.digits()), (p, odd)


def test_crt_carry_at_the_radix_9_size_of_config_4_and_its_rate():
    """p = 205271257 at 9*2^20 words (BASELINE configs[3]): parity with the oracle and the sweep's duration"""
    p, odd, n = 205271257, 9, 9 << 20
    o = orc_crt.OracleCrt(p, odd, n)
    rng = np.random.default_rng(p)
    w = o.widths().astype(np.uint64)
    o.set_digits(rng.integers(0, 1 << 62, o.n, dtype=np.uint64) & ((np.uint64(1) << w) - np.uint64(1)))
    o.square_mul(1)
    r61, r31 = o.weighted()
    digits, residual, ms = crt_carry(p, n, odd, 1, r61, r31, timed=True)
    want = o.digits()
    # vectorised finish: residual carries, then carry passes until nothing is left
    d = digits.copy()
    idx = ((np.nonzero(residual)[0] + 1) * 8) % n
    np.add.at(d, idx, residual[np.nonzero(residual)[0]])
    for _ in range(64):
        c = d >> w
        if not c.any():
            break
        d = (d & ((np.uint64(1) << w) - np.uint64(1))) + np.roll(c, 1)
    assert np.array_equal(d, want)
    print("crt carry sweep: %.3f ms for %d words (%.0f GB/s of 20 B/word)" % (ms, n, 20 * n / ms / 1e6))
    assert ms < 1.0
