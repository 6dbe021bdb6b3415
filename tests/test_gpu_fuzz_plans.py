"""Seeded sweep over (exponent, forced plan) pairs nobody picked by hand: random exponents from p ~ 2 000 to p ~ 10 000 000 (n = 2^7 .. 2^19 and
5 2^k), every admissible split m = M1 x M2 and run length C the plan accepts for them -- so every kernel set meets shapes at its edges (one
tile, one row, columns of 5 .. 5120, rows of 2 .. 8192) -- each through a short register-machine program against the oracle, bit-exact.
The list is fixed by the seed and by the plan policy (resolve_plan runs without a GPU).  Needs a real MI355X:  python -m pytest tests -m gpu"""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


def _cases(count=200, seed=20261005):
    try:
        from prmers_amd import resolve_plan
        from prmers_amd.engine import EngineError
    except Exception:   # library not built: nothing to collect
        return []
    rng = np.random.default_rng(seed)
    out, seen = [], set()
    while len(out) < count:
        lg = rng.uniform(11.0, 23.3)
        p = int(2 ** lg) | 1
        m2 = 1 << int(rng.integers(1, 14))
        c = 1 << int(rng.integers(0, 4))
        spec = "m2=%d,c=%d" % (m2, c)
        try:
            desc = resolve_plan(p, spec)
        except EngineError:
            continue
        if desc in seen:
            continue
        seen.add(desc)
        out.append((p, spec))
    return out


CASES = _cases()


def Engine(*a, **k):
    from prmers_amd import Engine as E
    return E(*a, **k)


@pytest.mark.parametrize("p,plan", CASES)
def test_random_exponent_and_plan_against_the_oracle(p, plan):
    rng = np.random.default_rng(p)
    x0 = int.from_bytes(rng.bytes((p + 7) // 8), "little") % ((1 << p) - 1)
    o = orc.Oracle(p, 3)
    o.set_value(0, x0)
    with Engine(p, 4, plan=plan) as e:
        assert e.n == o.n
        e.set_int(0, x0)
        for a in (1, 1, 3):
            e.square_mul(0, a); o.square_mul(0, a)
        assert np.array_equal(e.digits(0), o.digits(0)), (p, plan, e.describe())
        e.copy(1, 0); o.copy(1, 0)
        e.set_multiplicand(2, 1); o.set_multiplicand(2, 1)
        e.sub(0, 2); o.sub(0, 2)
        e.square_mul(0); o.square_mul(0)
        e.mul(0, 2, 5); o.mul(0, 2, 5)
        e.square_mul_n(0, 3, 1, 2)
        for _ in range(3): o.square_mul(0); o.sub(0, 2)
        assert np.array_equal(e.digits(0), o.digits(0)), (p, plan, e.describe())
        assert e.res64(0) == o.res64(0)
