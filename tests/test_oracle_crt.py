"""Pins of the second oracle (oracle/oracle_crt.c: GF(M61^2) x GF(M31^2) with a prime-factor radix-3 / radix-9 axis, SURVEY.md 8f
N1): Python big integers, the reference's own CPU prototype run here (oracle/_ref/ref_mixed_crt, built in place from
docs/mersenne2_mixed_crt_2d_half_fast/*.cpp), and the libgmp pins at BASELINE configs[3] with the PFA sizes."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import orc_crt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_mixed_crt")


@pytest.mark.parametrize("p,odd", [(127, 1), (521, 1), (521, 3), (1279, 3), (1279, 9), (2203, 9), (4423, 3), (9941, 9), (11213, 9), (11213, 3), (19937, 1)])
def test_squarings_match_python_integers(p, odd):
    o = orc_crt.OracleCrt(p, odd)
    assert o.n % odd == 0 and o.n == orc_crt.lib().orcc_transform_size(p, odd)
    Mp = (1 << p) - 1
    x = 3
    o.set(3)
    for it in range(min(p, 60)):
        a = 3 if it % 7 == 3 else 1
        o.square_mul(a)
        x = x * x * a % Mp
        assert o.value() == x, (p, odd, it)
    o.sub(2)
    assert o.value() == (x - 2) % Mp
    w = o.widths().astype(np.int64)
    assert int(w.sum()) == p and set(np.unique(w)) <= {p // o.n, p // o.n + 1}


@pytest.mark.parametrize("p,odd,prime", [(1279, 9, True), (2203, 3, True), (2281, 9, True), (2293, 9, False), (4253, 3, True), (4423, 9, True), (9941, 9, True), (9949, 3, False)])
def test_lucas_lehmer_verdicts_match_the_reference_prototype_run_here(p, odd, prime):
    """complete LL tests at forced radix 3 / 9: the oracle, the known status of 2^p - 1, and -- when the reference tree is present --
    the reference's own program (m2:1141-1190) on the same exponent and radix"""
    o = orc_crt.OracleCrt(p, odd)
    o.set(4)
    for _ in range(p - 2):
        o.square_mul(); o.sub(2)
    v = o.value()
    assert (v == 0) == prime
    if os.path.exists(REF):
        out = subprocess.run([REF, str(p), str(odd), "--no-progress"], capture_output=True, text=True).stdout
        assert ("transform=%d*2^" % odd) in out and ("= %d," % o.n) in out      # same transform size rule (m2:479-503)
        assert ("%d is prime" % p in out) == prime and ("%d is composite" % p in out) == (not prime)


@pytest.mark.parametrize("n", [9 << 20, 3 << 21])
def test_big_p_pins_at_the_pfa_sizes_of_config_4(n):
    """p = 205271257 with the forced radix-9 size 9*2^20 = 9,437,184 words and the automatic radix-3 size 3*2^21 = 6,291,456 words
    (README.md:907-926): residues of 3^(2^i) against tests/golden/big_p_pins.json (libgmp) at i = 30 and 31"""
    p = 205271257
    pins = {e["iteration"]: e for e in json.load(open(os.path.join(ROOT, "tests", "golden", "big_p_pins.json")))["pins"][str(p)]}
    o = orc_crt.OracleCrt(p, 9 if n % 9 == 0 else 3, n)
    assert o.n == n
    o.set(3)
    for it in range(1, 32):
        o.square_mul()
        if it in (30, 31):
            w = o.words()
            assert "%016X" % (int(w[0]) | (int(w[1]) << 32)) == pins[it]["res64"]
            assert hashlib.sha256(w.astype("<u4").tobytes()).hexdigest() == pins[it]["sha256_words"]


GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")))


def _res64(words):
    return "%016X" % (int(words[0]) | (int(words[1]) << 32))


@pytest.mark.parametrize("odd", [1, 3, 9])
def test_reference_held_m11213_residues_pin_this_oracle(odd):
    """unit_tests.sh:166-178 (Res64 of 3^(2^k) mod M11213 at k = 1000 ... 11000) and :152-153 (type-1 residue of the finished test):
    canonical residues do not depend on the field, so the vectors the reference holds for its Marin path pin this family as they stand"""
    import prmers_amd.prp as prp
    p = 11213
    want = {int(k): v for k, v in GOLD["m11213_intermediate_res64"].items() if k.isdigit()}
    assert sorted(want) == list(range(1000, 11001, 1000))
    o = orc_crt.OracleCrt(p, odd)
    o.set(3)
    for k in range(1, p + 1):
        o.square_mul()
        if k in want:
            assert _res64(o.words()) == want[k], (odd, k)
    assert o.value() == 9                                    # 3^(2^p) = 9: M11213 is prime
    w = prp.prp3_div9(p, o.words())
    assert prp.format_res64(w) == GOLD["m11213_final"]["res64"] and prp.format_res2048(w) == "0" * 511 + "1"


@pytest.mark.parametrize("odd", [9])
def test_reference_held_m100003_residue_pins_this_oracle(odd):
    """unit_tests.sh:140-141: res64 and res2048 of the type-1 PRP residue of the composite M100003 (radix 9 here; the GPU suite runs the
    vector through the HIP engine at radix 3 and 9)"""
    import prmers_amd.prp as prp
    p = 100003
    o = orc_crt.OracleCrt(p, odd)
    o.set(3)
    for _ in range(p):
        o.square_mul()
    w = prp.prp3_div9(p, o.words())
    assert prp.format_res64(w) == GOLD["m100003"]["res64"]
    assert prp.format_res2048(w) == GOLD["m100003"]["res2048"].lower()
