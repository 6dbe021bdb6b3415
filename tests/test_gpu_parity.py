"""Parity of the HIP engine (through the C ABI) against the CPU oracle, golden vectors and Python
integers.  Needs a real MI355X:  python -m pytest tests -m gpu"""
import json
import os

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


def Engine(*a, **k):
    from prmers_amd import Engine as E
    return E(*a, **k)


def rand_residue(rng, p):
    return int.from_bytes(rng.bytes((p + 7) // 8), "little") % ((1 << p) - 1)


# exponent, forced plan: single-row plans, two-level plans, radix-5 columns, odd/even log2 sizes
SMALL_CASES = [
    (31, None), (61, None), (89, None), (127, None), (127, "m2=2,c=1"), (127, "m2=2,c=2"),
    (521, None), (521, "m2=4,c=2"), (521, "m2=8,c=2"), (521, "m2=2,c=2"),
    (933, None), (933, "m2=2,c=2"), (1801, "m2=8,c=4"), (1801, "m2=4,c=4"), (3997, "m2=16,c=4"), (3997, None),
    (9941, None), (9941, "m2=16,c=4"), (9941, "m2=64,c=8"), (9941, "m2=4,c=4"), (9941, "m2=128,c=2"),
    (13967, None), (13967, "m2=16,c=8"), (44497, None), (44497, "m2=32,c=4"), (102701, None), (102701, "m2=64,c=2"),
    # shapes served by the register-resident radix-8 kernels: rows of 4096, columns of 512 x 8, 1024 x 4, 2048 x 2
    (300007, "m2=4096"), (300007, "m2=8,c=4"), (300007, "m2=16,c=8"), (300007, "m2=4,c=2"), (216091, None),
    (600011, "m2=32,c=8"), (600011, "m2=8,c=2"), (1200007, "m2=64,c=8"),
    # shapes served by the register-resident radix-4 kernels (kernels_v3.hip, round 4): columns of 256 x 4 (with generic rows and with
    # their own rows of 1024: p = 9815459 below), rows of 1024 over generic columns of 8 .. 64
    (86243, "m2=8,c=4"), (132049, "m2=16,c=4"), (300007, "m2=32,c=4"), (756839, "m2=64,c=4"),
    (300007, "m2=1024"), (600011, "m2=1024"), (1200007, "m2=1024,c=4"), (2976221, "m2=1024"),
    # rows of 2048 with one plane per thread (kernels_v2.hip k2_rows2048_planes: fewer than 512 rows) over generic and radix-4 columns
    (300007, "m2=2048"), (600011, "m2=2048"), (1200007, "m2=2048,c=4"), (2976221, "m2=2048"), (4800007, "m2=2048,c=4"),
    # columns of 1280 = 5 x 256 on the register-resident radix-5 kernels (640 threads per tile), generic and radix-8 rows
    (400063, "m2=8,c=4"), (800283, "m2=16,c=4"), (1600589, "m2=32,c=4"),
    # columns of 2560 = 5 x 512 with runs of two pairs on the same kernels (round 4: the columns of n = 5 2^22 and 5 2^23)
    (800283, "m2=8,c=2"), (1600589, "m2=16,c=2"), (3200123, "m2=32,c=2"),
    # rows of 8192 (two 4096-point halves under one radix-2 level, 1024 threads)
    (300007, "m2=8192"), (600011, "m2=8192"), (1200007, "m2=8192,c=2"),
    # the split column sweeps of n = 5 * 2^26 (radix-5 stage through memory, power-of-two part in LDS) forced at small 5 * 2^k sizes:
    # L1 = 4 .. 2048, generic rows and the register-resident rows of 4096 / 8192
    (1001, "m2=2,split5"), (3585, "m2=4,split5"), (13825, "m2=8,split5"), (53331, "m2=16,split5"), (400063, "m2=8,split5"),
    (800283, "m2=4096,split5"), (1600589, "m2=8192,split5"), (3200123, "m2=8,split5"), (3200123, "m2=4096,split5"),
]


@pytest.mark.parametrize("p,plan", SMALL_CASES)
def test_prp_iterations_match_oracle_and_bigint(p, plan):
    """x <- x^2 from 3: digit vector bit-exact vs the oracle each iteration, value vs Python ints
    (Python big-integer squaring only up to p = 250000: beyond that it would dominate the suite)."""
    iters = min(p, 150)
    Mp = (1 << p) - 1
    bigint = p <= 250000
    o = orc.Oracle(p, 2)
    with Engine(p, 2, plan=plan) as e:
        assert e.n == o.n
        e.set(0, 3); o.set(0, 3)
        x = 3
        for it in range(iters):
            e.square_mul(0); o.square_mul(0)
            if bigint:
                x = x * x % Mp
            if it % 7 == 0 or it > iters - 4:
                assert np.array_equal(e.digits(0), o.digits(0)), (p, plan, it)
                if bigint:
                    assert e.get_int(0) == x
        assert e.res64(0) == o.res64(0)


@pytest.mark.parametrize("p", [89, 127, 521, 607, 1279, 2203])
def test_prime_exponents_end_to_end(p):
    """unit_tests.sh:5-10: 3^(2^p) == 9 for Mersenne-prime exponents."""
    with Engine(p, 2) as e:
        e.set(0, 3)
        for _ in range(p):
            e.square_mul(0)
        d = e.digits(0)
        assert orc.lib().orc_digits_equal_to(d.ctypes.data, e.n, 9) == 1
        r64, _ = orc.prp_type1_hex(d, p)
        assert r64 == "0000000000000001"


@pytest.mark.parametrize("p", GOLD["composite_exponents"])
def test_composite_exponents(p):
    """unit_tests.sh:12-14: composite results, and the full residue vs the oracle."""
    o = orc.Oracle(p, 1)
    with Engine(p, 2) as e:
        e.set(0, 3); o.set(0, 3)
        for _ in range(p):
            e.square_mul(0); o.square_mul(0)
        d = e.digits(0)
        assert np.array_equal(d, o.digits(0))
        assert orc.lib().orc_digits_equal_to(d.ctypes.data, e.n, 9) == 0


def test_m11213_golden_res64():
    """unit_tests.sh:166-178 intermediate Res64 + :152-153 final."""
    want = {int(k): v for k, v in GOLD["m11213_intermediate_res64"].items() if k != "src"}
    with Engine(11213, 2) as e:
        e.set(0, 3)
        for it in range(1, 11214):
            e.square_mul(0)
            if it in want:
                assert "%016X" % e.res64(0) == want[it], it
        d = e.digits(0)
        r64, r2048 = orc.prp_type1_hex(d, 11213)
        assert r64 == GOLD["m11213_final"]["res64"] and r2048 == "0" * 511 + "1"


def test_m100003_golden_res64_res2048():
    """unit_tests.sh:140-141."""
    p = 100003
    with Engine(p, 2) as e:
        e.set(0, 3)
        for _ in range(p):
            e.square_mul(0)
        r64, r2048 = orc.prp_type1_hex(e.digits(0), p)
        assert r64 == GOLD["m100003"]["res64"]
        assert r2048 == GOLD["m100003"]["res2048"]


def test_reg_adapter_contract():
    """tests/test_aevum_reg_adapter.cpp:32-93 through our boundary."""
    g = GOLD["reg_adapter"]
    p = g["p"]
    M = (1 << p) - 1
    with Engine(p, 8) as e:
        e.set(0, 5); e.set(1, 7)
        e.set_multiplicand(2, 1)
        e.mul(0, 2)
        assert e.get_int(0) == g["mul"]
        e.square_mul(0, 3)
        assert e.get_int(0) == g["square_mul3"]
        e.add(0, 1); e.sub_reg(0, 1); e.sub(0, 2)
        assert e.get_int(0) == g["addsub"]
        e.set_int(3, M + 123)
        assert e.get_int(3) == g["set_mpz"]
        one = e.get_data(3)
        e.set(4, 0)
        assert e.set_data(4, one)
        assert e.get_int(4) == 123
        assert e.is_equal(3, 4)
        e.set(4, 124)
        assert not e.is_equal(3, 4)
        e.set(4, 123)
        ck = e.get_checkpoint()
        e.set(0, 1); e.set(3, 1)
        assert e.set_checkpoint(ck)
        assert e.get_int(0) == g["addsub"] and e.get_int(3) == 123
        e.set(5, g["pow_base"])
        e.pow(6, 5, g["pow_exp"])
        assert e.get_int(6) == pow(g["pow_base"], g["pow_exp"], M)
        assert e.res64(3) == 123
        assert not e.set_data(4, one[:-1])          # size mismatch -> false (engine_gpu.h:2138)
        e.sync()


@pytest.mark.parametrize("p,plan", [(127, None), (1279, None), (9941, "m2=16,c=4"), (3997, None), (3997, "m2=4,c=2"),
                                    (86243, None), (216091, None), (300007, "m2=4096"), (300007, "m2=8,c=4")])
def test_ops_random_vs_bigint_and_oracle(p, plan):
    M = (1 << p) - 1
    rng = np.random.default_rng(p)
    o = orc.Oracle(p, 4)
    with Engine(p, 6, plan=plan) as e:
        for _ in range(4):
            x, y = rand_residue(rng, p), rand_residue(rng, p)
            a = int(rng.integers(1, 1000))
            e.set_int(0, x); e.set_int(1, y)
            o.set_value(0, x); o.set_value(1, y)
            assert e.get_int(0) == x and e.get_int(1) == y
            assert np.array_equal(e.digits(0), o.digits(0))
            e.set_multiplicand(2, 1); o.set_multiplicand(2, 1)
            e.mul(0, 2, a); o.mul(0, 2, a)
            assert e.get_int(0) == x * y * a % M
            assert np.array_equal(e.digits(0), o.digits(0))
            e.square_mul(0, a); o.square_mul(0, a)
            z = (x * y * a) ** 2 * a % M
            assert e.get_int(0) == z
            e.add(0, 1)
            assert e.get_int(0) == (z + y) % M
            e.sub_reg(0, 1); e.sub_reg(0, 1)
            assert e.get_int(0) == (z - y) % M
            e.sub(0, 2)
            assert e.get_int(0) == (z - y - 2) % M
            e.copy(3, 0)
            assert e.is_equal(3, 0)
        # set_multiplicand in place, then square_mul of another register still works
        e.set_int(4, 12345)
        e.set_multiplicand(4, 4)
        e.set_int(5, 777)
        e.mul(5, 4)
        assert e.get_int(5) == 12345 * 777 % M
        # LL: x -> x^2 - 2 from 4 (RunPrpOrLlMarin.cpp:229,321-324)
        e.set(0, 4)
        s = 4
        for _ in range(30):
            e.square_mul(0); e.sub(0, 2)
            s = (s * s - 2) % M
        assert e.get_int(0) == s


def test_edge_values():
    """0, 1, Mp (== 0), Mp - 1 and borrow chains through sub."""
    p = 1279
    M = (1 << p) - 1
    with Engine(p, 4) as e:
        e.set(0, 0)
        e.square_mul(0)
        assert e.get_int(0) == 0
        e.set(0, 1)
        e.sub(0, 2)                      # 1 - 2 = Mp - 1: borrow ripples through every digit
        assert e.get_int(0) == M - 1
        e.square_mul(0)
        assert e.get_int(0) == 1
        e.set_int(1, M - 1)
        e.set(2, 1)
        e.add(1, 2)                      # == Mp == 0
        assert e.get_int(1) == 0
        e.set_int(1, M - 1)
        e.square_mul(1, 3)               # (-1)^2 * 3
        assert e.get_int(1) == 3
        e.set(3, 0xFFFFFFFF)
        assert e.get_int(3) == 0xFFFFFFFF
        with pytest.raises(Exception):
            e.square_mul(0, 0)           # factor 0 rejected (EngineApi.cpp:255)
        with pytest.raises(Exception):
            e.set(9, 1)                  # register out of range
        e.set_multiplicand(2, 1)
        with pytest.raises(Exception):
            e.square_mul(2)              # a multiplicand is not a residue


@pytest.mark.parametrize("p", [9815459, 19000013, 50000017, 100000007, 136279841, 205271257, 250000013, 332000003])
def test_extreme_digits_full_size(p):
    """Every digit at its maximum (x = Mp - 1 = -1: the largest convolution sums and the longest carry chains the transform can see) at the
    full-size shapes of every register-resident kernel set: (-1)^2 = 1, (-1)^2 * 3 = 3, then 3^2 - 2 through the deferred subtraction; digit
    vectors against the oracle and the values by hand."""
    o = orc.Oracle(p, 2)
    w = o.widths().astype(np.uint64)
    d = ((np.uint64(1) << w) - np.uint64(1)) | (w << np.uint64(32))
    d[0] -= np.uint64(1)                                   # Mp - 1
    with Engine(p, 2) as e:
        e.set_digits(0, d); o.set_digits(0, d)
        e.square_mul(0); o.square_mul(0)
        got = e.digits(0)
        assert np.array_equal(got, o.digits(0))
        assert int(got[0] & np.uint64(0xFFFFFFFF)) == 1 and not np.any(got[1:] & np.uint64(0xFFFFFFFF))
        e.set_digits(0, d); o.set_digits(0, d)
        e.square_mul(0, 3); o.square_mul(0, 3)
        e.sub(0, 2); o.sub(0, 2)
        e.square_mul(0); o.square_mul(0)                   # (3 - 2)^2 = 1
        got = e.digits(0)
        assert np.array_equal(got, o.digits(0))
        assert int(got[0] & np.uint64(0xFFFFFFFF)) == 1 and not np.any(got[1:] & np.uint64(0xFFFFFFFF))
        assert e.res64(0) == 1


def test_c2_9815459_first_iterations():
    """BASELINE config C2 (n = 2^19): first 60 squarings from 3, digits vs the oracle, then Gerbicz-style
    identity d * x^? skipped here (see test_prp_driver)."""
    p = 9815459
    o = orc.Oracle(p, 1)
    with Engine(p, 2) as e:
        assert e.n == 1 << 19
        e.set(0, 3); o.set(0, 3)
        for it in range(60):
            e.square_mul(0); o.square_mul(0)
            if it in (0, 20, 40, 59):
                assert np.array_equal(e.digits(0), o.digits(0)), it


@pytest.mark.parametrize("p,n", [(9815459, 1 << 19), (4800007, 1 << 18), (50000017, 5 << 19), (19000013, 1 << 20)])
def test_radix4_set_full_size_operations(p, n):
    """The shapes of the register-resident radix-4 kernels at full-size digits (kernels_v3.hip): C2 (columns of 256 x 4 + rows of 1024),
    n = 2^18 (rows of 1024 over generic columns of 128), n = 5 2^19 (rows of 1024 under the radix-5 columns) and n = 2^20 (columns of
    256 x 4 around the plane-per-thread rows of 2048, kernels_v2.hip k2_rows2048_planes): squarings with a factor,
    the LL step with the subtraction deferred into the next sweep, set_multiplicand / mul (forward-only and multiply modes of the row
    kernel), the fused back sweeps (mul_add, square_mul_copy) -- digit vectors against the oracle."""
    o = orc.Oracle(p, 4)
    rng = np.random.default_rng(p)
    w = o.widths().astype(np.uint64)
    d0 = (rng.integers(0, 1 << 62, o.n, dtype=np.uint64) & ((np.uint64(1) << w) - np.uint64(1))) | (w << np.uint64(32))
    with Engine(p, 4) as e:
        assert e.n == n == o.n
        e.set_digits(0, d0); o.set_digits(0, d0)
        for it in range(3):
            e.square_mul(0); o.square_mul(0)
        assert np.array_equal(e.digits(0), o.digits(0))
        e.square_mul(0, 3); o.square_mul(0, 3)
        e.sub(0, 2); o.sub(0, 2)
        e.square_mul(0); o.square_mul(0)
        assert np.array_equal(e.digits(0), o.digits(0))
        e.copy(1, 0); o.copy(1, 0)
        e.square_mul(1); o.square_mul(1)
        e.set_multiplicand(2, 1); o.set_multiplicand(2, 1)
        e.mul(0, 2); o.mul(0, 2)
        assert np.array_equal(e.digits(0), o.digits(0))
        e.mul_add(0, 2, 1); o.mul(0, 2); o.add(0, 1)           # dst = dst * src + add_src (engine.h:65)
        assert np.array_equal(e.digits(0), o.digits(0))
        e.square_mul_copy(0, 3); o.square_mul(0); o.copy(3, 0)   # engine.h:81
        assert np.array_equal(e.digits(0), o.digits(0)) and np.array_equal(e.digits(3), o.digits(3))
        assert e.res64(0) == o.res64(0) and e.is_equal(0, 3)


def test_c3_136279841_full_size():
    """BASELINE config C3 (n = 2^23, both radix-8 kernels in use): squarings, the LL step x^2-2, mul and
    the Gerbicz identity against the oracle's digit vectors."""
    p = 136279841
    o = orc.Oracle(p, 3)
    rng = np.random.default_rng(5)
    w = o.widths().astype(np.uint64)
    d0 = (rng.integers(0, 1 << 62, o.n, dtype=np.uint64) & ((np.uint64(1) << w) - np.uint64(1))) | (w << np.uint64(32))
    with Engine(p, 4) as e:
        assert e.n == 1 << 23
        e.set_digits(0, d0); o.set_digits(0, d0)
        assert np.array_equal(e.digits(0), o.digits(0))
        for it in range(3):
            e.square_mul(0); o.square_mul(0)
        assert np.array_equal(e.digits(0), o.digits(0))
        e.square_mul(0, 3); o.square_mul(0, 3)
        e.sub(0, 2); o.sub(0, 2)                       # LL step with the subtraction deferred into the next sweep
        e.square_mul(0); o.square_mul(0)
        assert np.array_equal(e.digits(0), o.digits(0))
        e.copy(1, 0); o.copy(1, 0)
        e.set_multiplicand(2, 1); o.set_multiplicand(2, 1)
        e.mul(0, 2); o.mul(0, 2)
        assert np.array_equal(e.digits(0), o.digits(0))
        assert e.res64(0) == o.res64(0)


@pytest.mark.parametrize("p,plan,shape", [(30402457, None, "m1=512:m2=2048:c=8"), (30402457, "m2=2048,c=4", "m1=512:m2=2048:c=4"),
                                          (100000007, None, "m1=1280:m2=2048:c=4"), (38000009, "m2=2048,c=2", "m1=512:m2=2048:c=2")])
def test_rows_of_2048_two_to_a_tile(p, plan, shape, monkeypatch):
    """rows of 2048 on the register-resident row kernel (kernels_v2.hip, RL = 1: two rows per 4096-pair tile; the reference's
    forward1024 / sqr512 shapes, kernels/marin.cl:1190,1517): n = 2^21 with register-resident and with generic columns, n = 5 2^20
    (p ~ 100 M) with the radix-5 columns -- squarings, x a, the LL step (subtraction folded into the next sweep), multiplicand and mul
    against the oracle's digits, and against the generic rows (MI355_TUNE bit 6) on the same inputs."""
    o = orc.Oracle(p, 3)
    rng = np.random.default_rng(p)
    w = o.widths().astype(np.uint64)
    d0 = (rng.integers(0, 1 << 62, o.n, dtype=np.uint64) & ((np.uint64(1) << w) - np.uint64(1))) | (w << np.uint64(32))
    with Engine(p, 4, plan=plan) as e:
        assert shape in e.describe(), e.describe()
        monkeypatch.setenv("MI355_TUNE", "64")
        with Engine(p, 4, plan=plan) as g:
            e.set_digits(0, d0); g.set_digits(0, d0); o.set_digits(0, d0)
            for it in range(3):
                e.square_mul(0); g.square_mul(0); o.square_mul(0)
            assert np.array_equal(e.digits(0), o.digits(0)) and np.array_equal(g.digits(0), o.digits(0))
            e.square_mul(0, 3); o.square_mul(0, 3)
            e.sub(0, 2); o.sub(0, 2)
            e.square_mul(0); o.square_mul(0)
            assert np.array_equal(e.digits(0), o.digits(0))
            e.copy(1, 0); o.copy(1, 0)
            e.square_mul(1); o.square_mul(1)
            e.set_multiplicand(2, 1); o.set_multiplicand(2, 1)
            e.mul(0, 2, 3); o.mul(0, 2, 3)
            e.square_mul(0); o.square_mul(0)
            assert np.array_equal(e.digits(0), o.digits(0))
            assert e.res64(0) == o.res64(0)


@pytest.mark.parametrize("p", [30402457, 100000007])
def test_rows_of_2048_one_plane_per_thread_forced(p, monkeypatch):
    """k2_rows2048_planes forced (MI355_TUNE bit 14) where the default keeps two rows to a tile (n = 2^21, 5 2^20): more than one round of
    tiles per CU, radix-8 and radix-5 columns around it -- squarings, x a, LL step, multiplicand / mul against the oracle's digits."""
    o = orc.Oracle(p, 3)
    rng = np.random.default_rng(p + 1)
    w = o.widths().astype(np.uint64)
    d0 = (rng.integers(0, 1 << 62, o.n, dtype=np.uint64) & ((np.uint64(1) << w) - np.uint64(1))) | (w << np.uint64(32))
    monkeypatch.setenv("MI355_TUNE", "16384")
    with Engine(p, 4) as e:
        e.set_digits(0, d0); o.set_digits(0, d0)
        for it in range(3):
            e.square_mul(0); o.square_mul(0)
        assert np.array_equal(e.digits(0), o.digits(0))
        e.square_mul(0, 3); o.square_mul(0, 3)
        e.sub(0, 2); o.sub(0, 2)
        e.square_mul(0); o.square_mul(0)
        assert np.array_equal(e.digits(0), o.digits(0))
        e.copy(1, 0); o.copy(1, 0)
        e.square_mul(1); o.square_mul(1)
        e.set_multiplicand(2, 1); o.set_multiplicand(2, 1)
        e.mul(0, 2, 3); o.mul(0, 2, 3)
        e.square_mul(0); o.square_mul(0)
        assert np.array_equal(e.digits(0), o.digits(0))
        assert e.res64(0) == o.res64(0)


def test_full_size_properties_no_oracle():
    """size-independent identities at n = 2^23: (x*y)^2 == x^2 * y^2 and (x+y)^2 - x^2 - y^2 == 2xy."""
    p = 136279841
    rng = np.random.default_rng(11)
    with Engine(p, 8) as e:
        n = e.n
        dig = lambda: e.digits(0) * 0   # noqa: E731
        base = e.digits(7)              # widths (register 7 is zero)
        w = base >> np.uint64(32)
        for r in (0, 1):
            e.set_digits(r, (rng.integers(0, 1 << 62, n, dtype=np.uint64) & ((np.uint64(1) << w) - np.uint64(1))) | (w << np.uint64(32)))
        # (x*y)^2
        e.copy(2, 0); e.set_multiplicand(3, 1); e.mul(2, 3); e.square_mul(2)
        # x^2 * y^2
        e.copy(4, 0); e.square_mul(4); e.copy(5, 1); e.square_mul(5); e.set_multiplicand(6, 5); e.mul(4, 6)
        assert e.is_equal(2, 4)
        # (x+y)^2 - x^2 - y^2 == 2xy
        e.copy(2, 0); e.add(2, 1); e.square_mul(2)
        e.copy(4, 0); e.square_mul(4); e.sub_reg(2, 4)
        e.copy(4, 1); e.square_mul(4); e.sub_reg(2, 4)
        e.copy(4, 0); e.mul(4, 3, 2)
        assert e.is_equal(2, 4)


def test_c4_205271257_full_size_radix5():
    """BASELINE config C4 (n = 5*2^21, radix-5 first stage, rows of 4096): squarings with a factor, the
    Gerbicz-style mul and the LL subtraction vs the oracle's digit vectors."""
    p = 205271257
    o = orc.Oracle(p, 3)
    assert o.n == 5 << 21
    rng = np.random.default_rng(7)
    w = o.widths().astype(np.uint64)
    d0 = (rng.integers(0, 1 << 62, o.n, dtype=np.uint64) & ((np.uint64(1) << w) - np.uint64(1))) | (w << np.uint64(32))
    with Engine(p, 4) as e:
        assert e.n == o.n
        e.set_digits(0, d0); o.set_digits(0, d0)
        for a in (1, 3):
            e.square_mul(0, a); o.square_mul(0, a)
        assert np.array_equal(e.digits(0), o.digits(0))
        e.copy(1, 0); o.copy(1, 0)
        e.set_multiplicand(2, 1); o.set_multiplicand(2, 1)
        e.sub(0, 2); o.sub(0, 2)
        e.mul(0, 2); o.mul(0, 2)
        assert np.array_equal(e.digits(0), o.digits(0))
        assert e.res64(0) == o.res64(0)


@pytest.mark.parametrize("p,n", [(332000003, 5 << 22), (600000001, 1 << 25), (700000001, 5 << 23)])
def test_largest_supported_transforms(p, n):
    """the largest shapes the plan accepts (rows of 8192 = 128 KiB of LDS, columns of 2048 and 2560):
    squarings with a factor and a mul against the oracle's digit vectors."""
    o = orc.Oracle(p, 3)
    assert o.n == n
    rng = np.random.default_rng(p)
    w = o.widths().astype(np.uint64)
    d0 = (rng.integers(0, 1 << 62, o.n, dtype=np.uint64) & ((np.uint64(1) << w) - np.uint64(1))) | (w << np.uint64(32))
    with Engine(p, 4) as e:
        assert e.n == n
        e.set_digits(0, d0); o.set_digits(0, d0)
        for a in (1, 3):
            e.square_mul(0, a); o.square_mul(0, a)
        assert np.array_equal(e.digits(0), o.digits(0))
        e.copy(1, 0); o.copy(1, 0)
        e.set_multiplicand(2, 1); o.set_multiplicand(2, 1)
        e.sub(0, 2); o.sub(0, 2)
        e.mul(0, 2); o.mul(0, 2)
        assert np.array_equal(e.digits(0), o.digits(0))
        assert e.res64(0) == o.res64(0)


def test_largest_transform_size_of_the_reference_schedule():
    """n = 5 * 2^26 (include/marin/engine_gpu.h:1624), the last of the 46 sizes: columns of 20480 = 5 x 4096 pairs do not fit a CU's LDS, so
    the radix-5 stage of the column transform runs through a second work buffer (kernels.hip k_front_split_* / k_back_split_*).  One seeded
    squaring + LL step against the oracle's digit vector, then 3^(2^34) against the libgmp pins (tests/golden/largest_p_pins.json)."""
    import hashlib
    from prmers_amd import resolve_plan
    p = 4000000007
    assert resolve_plan(p) == "marin-hip:n=335544320:m1=20480:m2=8192:c=1:split5"
    o = orc.Oracle(p, 1)
    assert o.n == 5 << 26
    rng = np.random.default_rng(p)
    w = o.widths().astype(np.uint64)
    d0 = (rng.integers(0, 1 << 62, o.n, dtype=np.uint64) & ((np.uint64(1) << w) - np.uint64(1))) | (w << np.uint64(32))
    with Engine(p, 2) as e:
        e.set_digits(0, d0); o.set_digits(0, d0)
        e.square_mul(0, 3); o.square_mul(0, 3)
        e.sub(0, 2); o.sub(0, 2)
        assert np.array_equal(e.digits(0), o.digits(0))
        assert e.res64(0) == o.res64(0)
        del o
        e.copy(1, 0)
        assert e.is_equal(0, 1)
        pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "largest_p_pins.json")))["pins"][str(p)]
        e.set(0, 3)
        it = 0
        for pin in pins:
            while it < pin["iteration"]:
                e.square_mul(0); it += 1
            wds = e.words(0)
            assert "%016X" % (int(wds[0]) | (int(wds[1]) << 32)) == pin["res64"], (p, it)
            assert hashlib.sha256(wds.astype("<u4").tobytes()).hexdigest() == pin["sha256_words"], (p, it)


@pytest.mark.parametrize("p,n,plan", [(800000011, 1 << 26, "m1=4096:m2=8192:c=2"), (1300000003, 5 << 24, "m1=5120:m2=8192:c=2"),
                                      (1800000011, 5 << 25, "m1=10240:m2=8192:c=1")])
def test_largest_transform_sizes(p, n, plan):
    """The reference's schedule entries above 5*2^23 (include/marin/engine_gpu.h:1598,1622-1623): 2^26, 5*2^24, 5*2^25 --
    columns of 4096 / 5120 / 10240 pairs in LDS (generic set), rows of 8192 on the register-resident kernel."""
    from prmers_amd import resolve_plan
    assert resolve_plan(p).endswith(plan)
    o = orc.Oracle(p, 2)
    assert o.n == n
    rng = np.random.default_rng(p)
    w = o.widths().astype(np.uint64)
    d0 = (rng.integers(0, 1 << 62, o.n, dtype=np.uint64) & ((np.uint64(1) << w) - np.uint64(1))) | (w << np.uint64(32))
    with Engine(p, 2) as e:
        e.set_digits(0, d0); o.set_digits(0, d0)
        e.square_mul(0); o.square_mul(0)
        e.square_mul(0, 3); e.sub(0, 2); o.square_mul(0, 3); o.sub(0, 2)
        e.square_mul(0); o.square_mul(0)                          # consumes the weakly carried digits of the step before
        assert np.array_equal(e.digits(0), o.digits(0))
        assert e.res64(0) == o.res64(0)
        e.copy(1, 0)
        assert e.is_equal(0, 1)
    # x_0 = 3, x_{i+1} = x_i^2: GMP pins (tests/golden/huge_p_pins.json, made by make_huge_p_pins.py), no oracle involved
    import hashlib
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "huge_p_pins.json")))["pins"][str(p)]
    with Engine(p, 2) as e:
        e.set(0, 3)
        it = 0
        for pin in pins:
            while it < pin["iteration"]:
                e.square_mul(0); it += 1
            wds = e.words(0)
            assert "%016X" % (int(wds[0]) | (int(wds[1]) << 32)) == pin["res64"], (p, it)
            assert hashlib.sha256(wds.astype("<u4").tobytes()).hexdigest() == pin["sha256_words"], (p, it)


@pytest.mark.parametrize("p,n,m1", [(57885161, 1 << 22, 512), (250000013, 1 << 24, 2048)])
def test_register_resident_columns_other_shapes_full_size(p, n, m1):
    """n = 2^22 (columns of 512 x 8) and n = 2^24 (columns of 2048 x 2), both with rows of 4096: squarings with
    a factor, the LL step, mul and add against the oracle's digit vectors, then an operation sequence that
    exercises the deferred run carries (copy / set_multiplicand of a register with pending carries)."""
    from prmers_amd import resolve_plan
    assert ":m1=%d:m2=4096:" % m1 in resolve_plan(p)
    o = orc.Oracle(p, 4)
    assert o.n == n
    rng = np.random.default_rng(p)
    w = o.widths().astype(np.uint64)
    d0 = (rng.integers(0, 1 << 62, o.n, dtype=np.uint64) & ((np.uint64(1) << w) - np.uint64(1))) | (w << np.uint64(32))
    with Engine(p, 5) as e:
        e.set_digits(0, d0); o.set_digits(0, d0)
        for a in (1, 1, 3):
            e.square_mul(0, a); o.square_mul(0, a)
        assert np.array_equal(e.digits(0), o.digits(0))
        for _ in range(2):
            e.square_mul(0); e.sub(0, 2); o.square_mul(0); o.sub(0, 2)
        e.copy(1, 0); o.copy(1, 0)
        e.set_multiplicand(2, 1); o.set_multiplicand(2, 1)
        e.square_mul(0); o.square_mul(0)
        e.mul(0, 2, 5); o.mul(0, 2, 5)
        e.add(0, 1); o.add(0, 1)
        e.square_mul(0); o.square_mul(0)
        assert np.array_equal(e.digits(0), o.digits(0))
        assert np.array_equal(e.digits(1), o.digits(1))
        assert e.res64(0) == o.res64(0)


@pytest.mark.parametrize("p", [9815459, 136279841, 205271257])
def test_gmp_pins_at_baseline_exponents(p):
    """C2 / C3 / C4 from x0 = 3: res64, low 2048 bits and the SHA-256 of the canonical words at iterations
    30.. against the GMP-generated fixture (tests/golden/big_p_pins.json) -- no oracle involved."""
    from test_oracle_golden import check_pins
    with Engine(p, 2) as e:
        check_pins(e, p)


def test_device_selftest_of_field_primitives():
    """mi355_engine_selftest: the device code paths of gf.hpp / gfdft.hpp (borrow-reusing sub, P for a negated zero,
    lazy sums, every shift of mul_pow2, the LAZY butterflies) against 128-bit host arithmetic on edge and random
    operands, operands equal to P included."""
    from prmers_amd.engine import load_library
    L = load_library()
    assert L.mi355_engine_selftest(0) == 1, L.mi355_engine_last_error().decode()
