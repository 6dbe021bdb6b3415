"""Device-side strong carry / compare / res64 (prmers_amd/csrc/canon.hip) against the host path of the
same engine (MI355_HOST_CARRY=1: D2H + sequential carry, the reference's way, include/marin/engine_gpu.h:1534-1561),
the CPU oracle and Python integers.  Needs a real MI355X:  python -m pytest tests -m gpu"""
import os

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


def Engine(*a, **k):
    from prmers_amd import Engine as E
    return E(*a, **k)


class host_carry:
    """engines created inside use the host read-back path"""
    def __enter__(self):
        self.old = os.environ.get("MI355_HOST_CARRY")
        os.environ["MI355_HOST_CARRY"] = "1"
    def __exit__(self, *a):
        if self.old is None:
            del os.environ["MI355_HOST_CARRY"]
        else:
            os.environ["MI355_HOST_CARRY"] = self.old


CASES = [(127, None), (521, "m2=4,c=2"), (9941, None), (9941, "m2=16,c=4"), (44497, "m2=32,c=4"), (216091, None), (300007, "m2=4096"),
         (1200007, "m2=64,c=8"), (2976221, None)]


def special_values(p, rng):
    Mp = (1 << p) - 1
    vals = [0, 1, 2, 3, Mp - 1, Mp - 2, (1 << (p - 1)), (1 << (p - 1)) - 1, (1 << 64) - 1, (1 << 64), Mp >> 1, Mp ^ 1, Mp ^ (1 << (p // 2))]
    vals += [int.from_bytes(rng.bytes((p + 7) // 8), "little") % Mp for _ in range(3)]
    return vals


@pytest.mark.parametrize("p,plan", CASES)
def test_device_canonical_form_matches_host_and_bigint(p, plan):
    rng = np.random.default_rng(p)
    Mp = (1 << p) - 1
    with Engine(p, 4, plan=plan) as e:
        with host_carry():
            h = Engine(p, 4, plan=plan)
        try:
            vals = special_values(p, rng)
            if p > 2000000:   # (Python's big-integer squarings of 3 M bits dominate the suite otherwise: 105 s of its 400)
                vals = [vals[i] for i in (0, 3, 4, 6, 9, 12, 13)]
            for v in vals:
                e.set_int(0, v); h.set_int(0, v)
                assert e.get_int(0) == v % Mp == h.get_int(0)
                assert np.array_equal(e.digits(0), h.digits(0))
                assert e.res64(0) == h.res64(0) == (v % Mp) & ((1 << 64) - 1)
                # after arithmetic the digits are only weakly carried: x^2, then the strong carry on the device
                e.square_mul(0); h.square_mul(0)
                w = (v % Mp) ** 2 % Mp
                assert e.get_int(0) == w == h.get_int(0), (p, plan, hex(v)[:40])
                assert np.array_equal(e.digits(0), h.digits(0))
                assert e.res64(0) == w & ((1 << 64) - 1) == h.res64(0)
        finally:
            h.close()


@pytest.mark.parametrize("p,plan", [(127, None), (9941, "m2=16,c=4"), (216091, None), (1200007, "m2=64,c=8")])
def test_all_ones_is_zero_and_long_carry_chains(p, plan):
    """2^p - 1 == 0; digit vectors that are all ones but one carry unit ripple through every digit."""
    Mp = (1 << p) - 1
    with Engine(p, 4, plan=plan) as e:
        o = orc.Oracle(p, 4)
        ones = (np.uint64(1) << o.widths().astype(np.uint64)) - np.uint64(1)
        e.set_digits(0, ones); o.set_digits(0, ones)
        assert e.get_int(0) == 0                      # all ones == Mp == 0
        e.set(1, 0)
        assert e.is_equal(0, 1) and e.is_equal(1, 0)
        assert e.res64(0) == o.res64(0)               # the reference's res64 of the all-ones vector
        # all ones plus one at digit 0: the carry runs through all n digits and wraps: value 1
        d = ones.copy(); d[0] += np.uint64(1)
        e.set_digits(2, d)
        assert e.get_int(2) == 1 and e.res64(2) == 1
        e.set(3, 1)
        assert e.is_equal(2, 3)
        # ... plus one in the middle: value 2^offset
        k = o.n // 2
        d = ones.copy(); d[k] += np.uint64(1)
        e.set_digits(2, d)
        off = int(np.sum(o.widths()[:k].astype(np.int64)))
        assert e.get_int(2) == pow(2, off, Mp)
        if o.n > 20000:
            return
        # over-wide digits everywhere (values up to 2^32 - 1): several local carry passes
        rng = np.random.default_rng(7)
        wide = rng.integers(0, 1 << 32, o.n, dtype=np.uint64)
        e.set_digits(2, wide)
        want = 0
        offs = np.concatenate([[0], np.cumsum(o.widths().astype(np.int64))])
        for j in range(o.n):
            want += int(wide[j]) << int(offs[j])
        assert e.get_int(2) == want % Mp


@pytest.mark.parametrize("p,plan", [(9941, None), (216091, None), (1200007, "m2=64,c=8")])
def test_is_equal_on_the_device(p, plan):
    rng = np.random.default_rng(p + 1)
    Mp = (1 << p) - 1
    with Engine(p, 4, plan=plan) as e:
        x = int.from_bytes(rng.bytes((p + 7) // 8), "little") % Mp
        e.set_int(0, x); e.set_int(1, x)
        assert e.is_equal(0, 1)
        # same value reached through different weakly carried digit vectors: x^2 vs x*x through mul
        e.copy(2, 0); e.set_multiplicand(3, 0); e.mul(2, 3)
        e.square_mul(1)
        assert e.is_equal(1, 2)
        assert e.get_int(1) == x * x % Mp
        for bit in (0, 1, p // 2, p - 1):
            e.set_int(2, (x * x % Mp) ^ (1 << bit))
            assert not e.is_equal(1, 2), bit
        e.set_int(2, 0); e.set_int(3, Mp)
        assert e.is_equal(2, 3)


def test_full_size_check_moves_no_register_over_pcie():
    """C3 size: the Gerbicz-style comparison and res64 give the oracle's answers; timing is reported by bench.py."""
    p = 136279841
    o = orc.Oracle(p, 2)
    with Engine(p, 4) as e:
        d = orc.seeded_digits(p, o.n, 3) if hasattr(orc, "seeded_digits") else None
        if d is None:
            rng = np.random.default_rng(3)
            d = rng.integers(0, 1 << 16, o.n, dtype=np.uint64)
        e.set_digits(0, d); o.set_digits(0, d)
        for _ in range(3):
            e.square_mul(0); o.square_mul(0)
        assert e.res64(0) == o.res64(0)
        e.copy(1, 0)
        assert e.is_equal(0, 1)
        e.sub(1, 1)
        assert not e.is_equal(0, 1)
        assert np.array_equal(e.digits(0), o.digits(0))
