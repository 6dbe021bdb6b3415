"""Runs of squarings in one engine call (Engine::square_mul_n / mi355_engine_square_mul_n: what a PRP or Lucas-Lehmer loop issues between
two checks, RunPrpOrLlMarin.cpp:338-409) on the product library: the same digits as the loop of square_mul / sub calls, against the oracle,
the reference-held residues and Python integers, on the small generic plans, the register-resident kernels and the split sweeps.
(The one-cooperative-launch form of the small transforms is not in the product library: it was measured slower; its own
checks are tools/exp_coop_check.py, run against libmi355_engine_exp.so.)  Needs a real MI355X."""
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


# exponent, plan: two-level generic plans with runs of at least four digits (C >= 2), power-of-two and radix-5 columns, one tile .. 512 rows
RUN_CASES = [(127, "m2=2,c=2"), (521, "m2=4,c=2"), (1801, "m2=8,c=4"), (3997, "m2=16,c=4"), (9941, "m2=64,c=8"), (9941, "m2=4,c=4"),
             (13967, None), (44497, None), (102701, None), (400063, "m2=64,c=4"), (1001, "m2=2,c=2"), (2976221, None), (9815459, None),
             (19000013, None),
             # columns of 256 x 4 on the radix-4 set, 2 .. 16 tiles (the experimental library runs these as back + front in one launch)
             (86243, "m2=8,c=4"), (132049, "m2=16,c=4"), (756839, "m2=64,c=4")]


@pytest.mark.parametrize("p,plan", RUN_CASES)
def test_runs_of_squarings_match_the_oracle(p, plan):
    rng = np.random.default_rng(p)
    x0 = int.from_bytes(rng.bytes((p + 7) // 8), "little") % ((1 << p) - 1)
    o = orc.Oracle(p, 2)
    o.set_value(0, x0)
    with Engine(p, 2, plan=plan) as e:
        assert ":coop=" not in e.describe()   # the product library has no cooperative kernel
        assert e.n == o.n
        e.set_int(0, x0)
        e.square_mul_n(0, 17)
        for _ in range(17): o.square_mul(0)
        assert np.array_equal(e.digits(0), o.digits(0))
        e.square_mul_n(0, 5, 3)
        for _ in range(5): o.square_mul(0, 3)
        assert np.array_equal(e.digits(0), o.digits(0))
        # the Lucas-Lehmer form (x^2 - 2 folded into the next front sweep where the kernels can)
        e.square_mul_n(0, 9, 1, 2)
        for _ in range(9): o.square_mul(0); o.sub(0, 2)
        assert np.array_equal(e.digits(0), o.digits(0))
        assert e.res64(0) == o.res64(0)
        # pending state survives the other operations: copy, multiplicand, mul, add
        e.sub(0, 5); o.sub(0, 5)
        e.copy(1, 0); o.copy(1, 0)
        e.square_mul_n(1, 3); [o.square_mul(1) for _ in range(3)]
        e.set_multiplicand(1, 1); o.set_multiplicand(1, 1)
        e.mul(0, 1); o.mul(0, 1)
        e.square_mul(0); o.square_mul(0)
        assert np.array_equal(e.digits(0), o.digits(0))


@pytest.mark.parametrize("p,prime", [(521, True), (523, False), (2203, True), (9941, True), (9949, False), (11213, True)])
def test_lucas_lehmer_in_runs_of_squarings(p, prime):
    """s <- s^2 - 2 from 4, p - 2 times, as three calls of square_mul_n (unit_tests.sh:5-14 holds the verdicts)."""
    with Engine(p, 2) as e:
        e.set(0, 4)
        total = p - 2
        for part in (total // 3, total // 3, total - 2 * (total // 3)):
            e.square_mul_n(0, part, 1, 2)
        v = e.get_int(0)
        assert (v == 0) == prime


def test_prp_m11213_residues_of_the_reference_in_runs():
    """unit_tests.sh:166-178: Res64 of 3^(2^k) mod M11213 at k = 1000 .. 11000 -- a thousand squarings per call."""
    p = 11213
    with Engine(p, 2) as e:
        e.set(0, 3)
        for k, want in sorted((int(k), v) for k, v in GOLD["m11213_intermediate_res64"].items() if k != "src"):
            e.square_mul_n(0, 1000)
            assert "%016X" % e.res64(0) == want.upper(), k


def test_c2_exponent_a_gerbicz_block_in_runs():
    """BASELINE configs[1] (p = 9815459): two Gerbicz-Li blocks -- B = 3132 squarings per engine call, d <- d x, and the verification of the
    block (prp.py hands every run of plain iterations to square_mul_n) -- against the same blocks stepped one square_mul at a time."""
    from prmers_amd import prp
    p = 9815459
    B = int(p ** 0.5)
    with Engine(p, prp.REGISTERS) as e, Engine(p, 2) as ref:
        msgs = []
        r = prp.run_prp_or_ll(e, p, "prp", checklevel=1, max_iters=2 * B + 7, log=msgs.append)
        assert r["gerbicz_checks"] >= 1 and r["gerbicz_errors"] == 0 and any("Check passed" in m for m in msgs)
        ref.set(0, 3)
        for _ in range(2 * B + 7): ref.square_mul(0)
        assert np.array_equal(e.digits(0), ref.digits(0))


def test_an_injected_error_is_caught_when_the_blocks_run_as_single_calls():
    from prmers_amd import prp
    p = 86243
    with Engine(p, prp.REGISTERS) as e:
        msgs = []
        r = prp.run_prp_or_ll(e, p, "prp", checklevel=1, erroriter=1500, log=msgs.append)
        assert r["is_prime"] and r["gerbicz_errors"] == 1 and any("Check FAILED" in m for m in msgs)


@pytest.mark.parametrize("p,plan", [(57885161, None), (205271257, None), (1600589, "m2=32,c=4"), (1600589, "m2=16,c=2"), (3200123, "m2=8,split5")])
def test_runs_of_squarings_on_the_register_resident_and_split_paths(p, plan):
    """square_mul_n where it is the loop of launches (every shape the one-launch kernel does not serve): same digits as the loop of
    square_mul / sub calls, PRP and Lucas-Lehmer forms, pending subtraction carried across the calls"""
    rng = np.random.default_rng(p)
    x0 = int.from_bytes(rng.bytes(64), "little")
    with Engine(p, 2, plan=plan) as e:
        e.set_int(0, x0); e.set_int(1, x0)
        e.square_mul_n(0, 6, 3)
        for _ in range(6): e.square_mul(1, 3)
        assert np.array_equal(e.digits(0), e.digits(1))
        e.square_mul_n(0, 5, 1, 2); e.square_mul_n(0, 2, 1, 2)
        for _ in range(7): e.square_mul(1); e.sub(1, 2)
        assert np.array_equal(e.digits(0), e.digits(1)) and e.is_equal(0, 1)
