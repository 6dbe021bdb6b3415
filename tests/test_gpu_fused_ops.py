"""Fused variants of the engine surface (include/marin/engine.h:65-131, kernels/marin.cl:1856-2365):
add / sub_reg / addsub / addsub_copy as one run-wise sweep on pending-carry digits, mul_add / square_mul_copy /
mul_copy inside the back sweep -- against the oracle running the base-class compositions, on register-resident and
generic kernel shapes.  Needs a real MI355X:  python -m pytest tests -m gpu"""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

# exponent, plan: generic single-row, generic two-level (C = 1, 2, 4, 8), radix-5 columns, register-resident columns / rows
CASES = [(127, None), (127, "m2=2,c=1"), (521, "m2=4,c=2"), (9941, "m2=16,c=4"), (9941, "m2=64,c=8"), (13967, None), (44497, "m2=32,c=4"),
         (400063, "m2=8,c=4"), (800283, "m2=8,c=2"), (300007, "m2=8,c=4"), (300007, "m2=16,c=8"), (300007, "m2=4,c=2"), (216091, None), (600011, "m2=32,c=8"), (1200007, "m2=64,c=8")]


def Engine(*a, **k):
    from prmers_amd import Engine as E
    return E(*a, **k)


def rand_residue(rng, p):
    return int.from_bytes(rng.bytes((p + 7) // 8), "little") % ((1 << p) - 1)


def same(e, o, regs):
    for r in regs:
        assert np.array_equal(e.digits(r), o.digits(r)), r


@pytest.mark.parametrize("p,plan", CASES)
def test_fused_variants_match_the_compositions(p, plan):
    rng = np.random.default_rng(p)
    Mp = (1 << p) - 1
    o = orc.OracleEngine(p, 10)
    with Engine(p, 10, plan=plan) as e:
        x, y, z = (rand_residue(rng, p) for _ in range(3))
        for eng in (e, o):
            eng.set_int(0, x); eng.set_int(1, y); eng.set_int(2, z)
            eng.square_mul(0); eng.square_mul(1, 3)          # pending run carries on the inputs
        # add / sub_reg on pending-carry digits, results used by a squaring before anything is normalised
        for eng in (e, o):
            eng.add(0, 1); eng.sub_reg(2, 1); eng.square_mul(0); eng.square_mul(2)
        same(e, o, (0, 1, 2))
        # addsub: outputs alias the inputs
        e.addsub(0, 1, 0, 1)
        o.copy(8, 0); o.copy(9, 0); o.add(8, 1); o.sub_reg(9, 1); o.copy(0, 8); o.copy(1, 9)
        for eng in (e, o):
            eng.square_mul(0); eng.square_mul(1)
        same(e, o, (0, 1))
        # addsub_copy into four other registers
        e.addsub_copy(3, 4, 5, 6, 0, 2)
        o.copy(3, 0); o.copy(4, 0); o.add(3, 2); o.sub_reg(4, 2); o.copy(5, 3); o.copy(6, 4)
        same(e, o, (3, 4, 5, 6, 0, 2))
        assert e.get_int(3) == (e.get_int(0) + e.get_int(2)) % Mp
        assert e.get_int(4) == (e.get_int(0) - e.get_int(2)) % Mp
        # square_mul_copy / mul_copy / mul_add (the addend with and without pending carries, and equal to dst)
        e.square_mul_copy(3, 7, 3); o.square_mul(3, 3); o.copy(7, 3)
        same(e, o, (3, 7))
        for eng in (e, o):
            eng.set_multiplicand(8, 4)
        e.mul_copy(5, 8, 9, 1); o.mul(5, 8, 1); o.copy(9, 5)
        same(e, o, (5, 9))
        e.mul_add(6, 8, 3, 3); o.mul(6, 8, 3); o.add(6, 3)       # addend with pending carries
        e.mul_add(7, 8, 2, 1); o.mul(7, 8, 1); o.add(7, 2)
        e.mul_add(9, 8, 9, 1); o.copy(1, 9); o.mul(9, 8, 1); o.add(9, 1)   # dst is its own addend
        same(e, o, (6, 7, 9))
        for eng in (e, o):
            eng.square_mul(6); eng.square_mul(7); eng.square_mul(9)
        same(e, o, (6, 7, 9))
        assert e.res64(9) == o.res64(9)


def test_copy_keeps_the_pending_state():
    p = 300007
    o = orc.OracleEngine(p, 4)
    with Engine(p, 4, plan="m2=8,c=4") as e:
        for eng in (e, o):
            eng.set(0, 3)
            for _ in range(30):
                eng.square_mul(0)
            eng.sub(0, 2)            # a deferred small subtraction on top of pending run carries
            eng.copy(1, 0)
            eng.square_mul(1)
            eng.copy(2, 1)
        same(e, o, (0, 1, 2))
        assert e.is_equal(1, 2) and not e.is_equal(0, 1)


def test_fused_ops_reject_bad_arguments():
    from prmers_amd import EngineError
    with Engine(9941, 4) as e:
        e.set(0, 3); e.set(1, 5)
        with pytest.raises(EngineError):
            e.mul_add(0, 1, 1)           # not a multiplicand
        e.set_multiplicand(2, 1)
        with pytest.raises(EngineError):
            e.mul_add(0, 2, 1, 0)        # factor 0
        with pytest.raises(EngineError):
            e.addsub(0, 0, 0, 1)         # outputs must differ
        with pytest.raises(EngineError):
            e.add(0, 2)                  # a multiplicand is not a residue
        with pytest.raises(EngineError):
            e.square_mul_copy(0, 9)      # register out of range
