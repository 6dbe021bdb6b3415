"""The resident GF(M61^2) x GF(M31^2) squaring engine with a prime-factor radix-3 / radix-9 axis (prmers_amd/csrc/crt_engine.hip,
SURVEY.md 8f N1; reference third_party/aevum/src/cl/fft-middle.cl:663-720, pfaunpack.cl:12-56, carry.cl:506-588) through the C ABI,
against the CRT oracle (oracle/oracle_crt.c), Python integers and the libgmp pins.  Needs a real MI355X:  python -m pytest tests -m gpu"""
import json
import os

import numpy as np
import pytest

import orc_crt

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def CrtEngine(*a, **k):
    from prmers_amd import CrtEngine as E
    return E(*a, **k)


# (exponent, odd radix, forced words or 0, plan): single-pass rows, two-pass rows (h2 forces the four-step split at small sizes)
CASES = [(521, 1, 0, None), (607, 3, 0, None), (1279, 9, 0, None), (9941, 9, 0, None), (9941, 3, 0, "h2=2"), (11213, 9, 9 << 6, "h2=3"),
         (44497, 1, 0, "h2=4"), (86243, 9, 0, "h2=5"), (216091, 3, 0, None), (400063, 9, 9 << 11, "h2=6"), (1257787, 9, 0, None),
         (3021377, 1, 0, None), (6972593, 3, 3 << 17, "h2=7"),
         # rows of 1024 and columns of 2 .. 256: the radix-8 kernel set (crt_rows.hpp) with every last-step radix
         # (forced sizes keep at least 15 bits per word: the run-wise carry needs them, see the constructor's check)
         (756839, 9, 9 << 12, None), (216091, 1, 1 << 13, None), (1257787, 3, 3 << 14, None), (6972593, 9, 9 << 15, None), (1398269, 1, 1 << 16, None),
         (6972593, 3, 3 << 17, None), (37156667, 9, 9 << 18, None), (13466917, 1, 1 << 19, None)]


@pytest.mark.parametrize("p,odd,n,plan", CASES)
def test_square_mul_matches_the_oracle_digit_for_digit(p, odd, n, plan):
    with CrtEngine(p, odd, n, plan=plan) as e:
        o = orc_crt.OracleCrt(p, odd, e.n)   # the engine's automatic size is the reference's rule with at least 8 words per odd residue class
        assert e.n == o.n and (n == 0 or e.n == n)
        rng = np.random.default_rng(p + odd)
        w = o.widths().astype(np.uint64)
        start = rng.integers(0, 1 << 62, o.n, dtype=np.uint64) & ((np.uint64(1) << w) - np.uint64(1))
        o.set_digits(start); e.set_digits(0, start)
        for it, a in enumerate((1, 3, 1, 1)):
            o.square_mul(a); e.square_mul(0, a)
            assert np.array_equal(e.raw_digits(0), o.digits()), (p, odd, plan, it)
        assert np.array_equal(e.words(0), o.words())
        assert e.res64(0) == o.value() & ((1 << 64) - 1)


@pytest.mark.parametrize("p,odd,n", [(756839, 9, 9 << 12), (1257787, 3, 3 << 14), (37156667, 9, 9 << 18)])
def test_fused_back_and_carry_against_the_two_kernel_form(p, odd, n, monkeypatch):
    """k_back_carry (inverse odd axis + unweight + Garner + carry out of LDS, ranges of 512 digits per work-group) against k_back +
    k_crt_runs_linked (MI355_CRT_TUNE bit 1) and the oracle: the weakly carried digit vectors may differ, the residues may not"""
    rng = np.random.default_rng(p)
    with CrtEngine(p, odd, n) as e:
        monkeypatch.setenv("MI355_CRT_TUNE", "2")
        with CrtEngine(p, odd, n) as two:
            o = orc_crt.OracleCrt(p, odd, n)
            w = o.widths().astype(np.uint64)
            start = rng.integers(0, 1 << 62, n, dtype=np.uint64) & ((np.uint64(1) << w) - np.uint64(1))
            start[::512] = (np.uint64(1) << w[::512]) - np.uint64(1)          # all-ones digits at the range boundaries: carries cross them
            e.set_digits(0, start); two.set_digits(0, start); o.set_digits(start)
            for it, a in enumerate((1, 3, 1, 1, 1)):
                e.square_mul(0, a); two.square_mul(0, a); o.square_mul(a)
                d = e.raw_digits(0)
                assert np.array_equal(d, two.raw_digits(0)) and np.array_equal(d, o.digits()), (p, odd, it)
            assert e.is_equal(0, 0) and e.res64(0) == two.res64(0)


def test_both_kernel_sets_agree(monkeypatch):
    """the same squarings on the radix-8 row kernels and on the generic LDS radix-2 ones (MI355_CRT_KERNELS=generic)"""
    p, odd, n = 13466917, 9, 9 << 16
    rng = np.random.default_rng(5)
    with CrtEngine(p, odd, n) as e:
        assert e.describe().endswith("radix8")
        monkeypatch.setenv("MI355_CRT_KERNELS", "generic")
        with CrtEngine(p, odd, n) as g:
            assert g.describe().endswith("generic")
            start = rng.integers(0, 1 << 20, n, dtype=np.uint64)
            e.set_digits(0, start); g.set_digits(0, start)
            monkeypatch.setenv("MI355_CRT_KERNELS", "split")      # one field per column launch (the default from columns of 1024 on)
            with CrtEngine(p, odd, n) as sp:
                sp.set_digits(0, start)
                for a in (1, 3, 1):
                    e.square_mul(0, a); g.square_mul(0, a); sp.square_mul(0, a)
                    assert np.array_equal(e.raw_digits(0), g.raw_digits(0)) and np.array_equal(e.raw_digits(0), sp.raw_digits(0))


@pytest.mark.parametrize("p,odd", [(127, 1), (521, 3), (1279, 9), (2203, 9), (2281, 3)])
def test_lucas_lehmer_verdicts(p, odd):
    """complete LL tests of small Mersenne primes and of composites next to them, every iterate against Python integers"""
    for q, prime in ((p, True), (p + 2, False)):
        try:
            e = CrtEngine(q, odd)
        except Exception:
            if q == p:
                raise
            continue
        with e:
            Mq = (1 << q) - 1
            s = 4
            e.set(0, 4)
            for i in range(q - 2):
                e.square_mul(0, 1); e.sub(0, 2)
                s = (s * s - 2) % Mq
                if i % 97 == 0 or i == q - 3:
                    assert e.get_int(0) == s, (q, i)
            assert (e.get_int(0) == 0) == prime


@pytest.mark.parametrize("p,odd,n", [(9941, 9, 0), (216091, 3, 3 << 11), (6972593, 9, 9 << 15)])
def test_register_machine_copy_multiplicand_mul_add_sub_equal(p, odd, n):
    """set_multiplicand / mul / copy / add / sub_reg / is_equal / set_words of the same plugin ABI (EngineApi.h:28-59) against Python
    integers, on the generic and on the radix-8 kernel set"""
    Mp = (1 << p) - 1
    def red(v):                      # v mod 2^p - 1 by folding (Python's % is quadratic at seven million bits)
        while v >> p:
            v = (v & Mp) + (v >> p)
        return 0 if v == Mp else v
    rng = np.random.default_rng(p)
    with CrtEngine(p, odd, n, reg_count=6) as e:
        x = red(int.from_bytes(rng.bytes((p + 7) // 8), "little"))
        y = red(int.from_bytes(rng.bytes((p + 7) // 8), "little"))
        e.set_int(0, x); e.set_int(1, y)
        assert e.get_int(0) == x and e.get_int(1) == y
        e.copy(2, 0)
        e.set_multiplicand(3, 1)                       # register 3 <- image of y
        e.mul(2, 3, 3)                                 # x * y * 3
        assert e.get_int(2) == red(x * y * 3)
        e.copy(4, 3); e.copy(5, 0); e.mul(5, 4)        # a copied image multiplies the same
        assert e.get_int(5) == red(x * y)
        e.square_mul(0); e.copy(4, 0)
        assert e.get_int(4) == red(x * x)
        e.add(4, 2)
        assert e.get_int(4) == red(x * x + 3 * x * y)
        e.sub_reg(4, 2)
        assert e.get_int(4) == red(x * x) and e.is_equal(4, 0)
        e.sub(4, 1)
        assert not e.is_equal(4, 0)
        e.set(5, 0); e.set_int(4, Mp)                  # 2^p - 1 == 0
        assert e.is_equal(4, 5)
        with pytest.raises(Exception, match="multiplicand"):
            e.square_mul(3)
        with pytest.raises(Exception, match="multiplicand"):
            e.mul(0, 1)
        # the fused variants (engine.h:65-131)
        e.set_int(0, x); e.set_int(1, y)
        e.addsub(2, 4, 0, 1)
        assert e.get_int(2) == red(x + y) and e.get_int(4) == red(x + Mp - y)
        e.addsub(0, 1, 0, 1)                            # in place: (x, y) <- (x + y, x - y)
        assert e.get_int(0) == red(x + y) and e.get_int(1) == red(x + Mp - y)
        e.set_int(0, x); e.set_int(1, y)
        e.addsub_copy(2, 4, 5, 3, 0, 1)
        assert e.get_int(5) == e.get_int(2) == red(x + y) and e.get_int(3) == e.get_int(4) == red(x + Mp - y)
        e.set_multiplicand(3, 1); e.copy(2, 0); e.mul_add(2, 3, 0, 3)
        assert e.get_int(2) == red(3 * x * y + x)
        e.copy(2, 0); e.square_mul_copy(2, 4, 3)
        assert e.get_int(2) == e.get_int(4) == red(3 * x * x)
        e.copy(2, 0); e.mul_copy(2, 3, 5)
        assert e.get_int(2) == e.get_int(5) == red(x * y)
        # (x y)^2 = x^2 y^2 through two different operation orders
        e.set_int(0, x); e.set_int(1, y)
        e.set_multiplicand(3, 1); e.copy(2, 0); e.mul(2, 3); e.square_mul(2)
        e.square_mul(0); e.square_mul(1); e.set_multiplicand(3, 1); e.mul(0, 3)
        assert e.is_equal(0, 2)


def test_prp_with_gerbicz_li_checks_on_the_crt_engine():
    """the caller loop of the reference (RunPrpOrLlMarin.cpp:338-409 mirrored in prmers_amd/prp.py) runs unchanged on this engine:
    a complete PRP of M9941 and of a composite, Gerbicz-Li checks on, an injected error caught and repaired"""
    from prmers_amd import prp
    with CrtEngine(9941, 9, reg_count=prp.REGISTERS) as e:
        r = prp.run_prp_or_ll(e, 9941, "prp", checklevel=1)
        assert r["is_prime"] and r["gerbicz_checks"] >= 1 and r["gerbicz_errors"] == 0
    with CrtEngine(9949, 3, reg_count=prp.REGISTERS) as e:
        r = prp.run_prp_or_ll(e, 9949, "prp", checklevel=1, erroriter=4000)
        assert not r["is_prime"] and r["gerbicz_errors"] == 1


@pytest.mark.parametrize("odd", [3, 9])
def test_reference_held_residues_on_the_crt_engine(odd):
    """the vectors the reference's own test script holds (unit_tests.sh:166-178: Res64 of 3^(2^k) mod M11213 at k = 1000 ... 11000;
    :152-153 its final type-1 residue; :140-141 res64 + res2048 of M100003) through the HIP engine of this family at radix 3 and 9 --
    canonical residues do not depend on the field the transform runs in"""
    import json
    import os
    from prmers_amd import prp
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))
    want = {int(k): v for k, v in gold["m11213_intermediate_res64"].items() if k.isdigit()}
    with CrtEngine(11213, odd) as e:
        e.set(0, 3)
        for k in range(1, 11213 + 1):
            e.square_mul(0)
            if k in want:
                assert "%016X" % e.res64(0) == want[k], (odd, k)
        assert e.get_int(0) == 9
    with CrtEngine(100003, odd, reg_count=prp.REGISTERS) as e:
        r = prp.run_prp_or_ll(e, 100003, "prp")
        assert not r["is_prime"] and r["res64"] == gold["m100003"]["res64"] and r["res2048"] == gold["m100003"]["res2048"].lower()
    with CrtEngine(11213, odd, reg_count=prp.REGISTERS) as e:
        r = prp.run_prp_or_ll(e, 11213, "prp")
        assert r["is_prime"] and r["res64"] == gold["m11213_final"]["res64"] and r["res2048"] == "0" * 511 + "1"


def test_checkpoints_and_raw_images_on_the_crt_engine(tmp_path):
    """get_data / set_data / checkpoints (engine.h:134-146) with residues and a multiplicand image in the register file, and the caller's
    checkpoint file (prmers_amd/prp.py, version-2 layout) carrying a PRP across two engines"""
    from prmers_amd import prp
    p = 9941
    with CrtEngine(p, 9, reg_count=prp.REGISTERS) as e, CrtEngine(p, 9, reg_count=prp.REGISTERS) as f:
        e.set(0, 3)
        for _ in range(50):
            e.square_mul(0)
        e.copy(1, 0); e.set_multiplicand(2, 1)
        ck = e.get_checkpoint()
        assert ck.size == e.get_checkpoint_size() == prp.REGISTERS * e.get_register_data_size()
        assert f.set_checkpoint(ck)
        assert f.get_int(0) == e.get_int(0) == pow(3, 1 << 50, (1 << p) - 1)
        e.mul(0, 2); f.mul(0, 2)                         # the restored image multiplies the same
        assert f.get_int(0) == e.get_int(0) == pow(3, 1 << 51, (1 << p) - 1)
        assert not f.set_data(0, np.zeros(5, dtype=np.uint8))
    path = prp.checkpoint_name(p, "prp", str(tmp_path))
    with CrtEngine(p, 3, reg_count=prp.REGISTERS) as e:
        part = prp.run_prp_or_ll(e, p, "prp", max_iters=4000, ckpt_path=path, backup_every=1000, checklevel=1)
        assert not part["complete"]
    with CrtEngine(p, 3, reg_count=prp.REGISTERS) as e:
        msgs = []
        r = prp.run_prp_or_ll(e, p, "prp", ckpt_path=path, checklevel=1, log=msgs.append)
        assert "Resuming from a checkpoint." in msgs and r["is_prime"] and r["complete"] and r["gerbicz_errors"] == 0


def _max_exponent(n):
    import math
    lo, hi = n, 60 * n
    while lo < hi:
        mid = (lo + hi + 1) // 2
        if math.log2(n) + 2.0 * (mid / n + 1.0) < 92.0:
            lo = mid
        else:
            hi = mid - 1
    return lo


@pytest.mark.parametrize("odd,n", [(9, 9 << 9), (3, 3 << 12), (1, 1 << 14)])
def test_additions_at_the_top_of_a_size_range_keep_their_headroom(odd, n):
    """digit-wise additions leave digits above their widths; at the largest exponent a size admits the next transform has no
    bit to spare, so the engine must relax such a register first.  (x + 2 y)^2, (x - y) y ... against Python integers."""
    p = _max_exponent(n)
    if p % 2 == 0:
        p -= 1
    M = (1 << p) - 1
    rng = np.random.default_rng(p)
    x, y = (int.from_bytes(rng.bytes((p + 7) // 8), "little") % M for _ in range(2))
    with CrtEngine(p, odd, n, reg_count=6) as e:
        assert e.n == n
        e.set_int(0, x); e.set_int(1, y)
        e.add(0, 1); e.add(0, 1)                  # x + 2 y: two excess bits
        e.copy(2, 0)
        e.square_mul(0)
        assert e.get_int(0) == pow(x + 2 * y, 2, M)
        e.set_multiplicand(3, 2)                  # a multiplicand of a register that went through additions
        e.mul(1, 3, 3)
        assert e.get_int(1) == 3 * y * (x + 2 * y) % M
        e.set_int(4, x); e.set_int(5, y)
        for _ in range(20):                        # repeated additions without a transform in between
            e.add(4, 5)
        e.sub_reg(4, 5)
        e.square_mul(4)
        assert e.get_int(4) == pow(x + 19 * y, 2, M)
        e.set_int(4, x); e.set_int(5, y)
        e.addsub(2, 3, 4, 5)                       # sum -> 2, difference -> 3
        e.square_mul(2); e.square_mul(3)
        assert e.get_int(2) == pow(x + y, 2, M) and e.get_int(3) == pow(x - y, 2, M)


def test_device_side_canonical_form_on_the_crt_family(monkeypatch):
    """SURVEY.md 8f N4 on this family: is_equal / res64 / digits / sub_reg canonicalise on the device (canon.hip with u64 digits);
    the host path of round 2 (MI355_HOST_CARRY=1) and Python integers are the checkers: all ones, carry chains through every digit,
    digits far above their widths, 0 == 2^p - 1."""
    p, odd = 216091, 9
    M = (1 << p) - 1
    from prmers_amd import CrtEngine as E
    with E(p, odd) as e:
        n = e.n
        j = np.arange(n + 1, dtype=np.uint64)
        ceil = (j * np.uint64(p) + np.uint64(n - 1)) // np.uint64(n)
        w = (ceil[1:] - ceil[:-1]).astype(np.uint64)
        ones = (np.uint64(1) << w) - np.uint64(1)
        rng = np.random.default_rng(7)

        def value(d):
            v, sh = 0, 0
            for dj, wj in zip(d.tolist(), w.tolist()):
                v += int(dj) << sh
                sh += int(wj)
            return v % M
        cases = {"all_ones": ones.copy(), "ones_plus_one": ones.copy(), "chain": ones.copy(), "wide": rng.integers(0, 1 << 61, n, dtype=np.uint64),
                 "random": rng.integers(0, 1 << 62, n, dtype=np.uint64) & ones}
        cases["ones_plus_one"][0] += np.uint64(1)              # 2^p: the carry runs through every digit and wraps to 1
        cases["chain"][5] += np.uint64(3)
        for name, d in cases.items():
            e.set_digits(0, d)
            want = value(d)
            assert e.get_int(0) == want, name
            assert e.res64(0) == want & (2**64 - 1), name
            e.set_int(1, want)
            assert e.is_equal(0, 1) and e.is_equal(1, 0), name
            e.set_int(2, (want + 1) % M)
            assert not e.is_equal(0, 2), name
            e.set_int(3, 12345); e.sub_reg(3, 0)
            assert e.get_int(3) == (12345 - want) % M, name
        e.set_digits(0, cases["all_ones"]); e.set(1, 0)
        assert e.is_equal(0, 1) and e.res64(0) == 0            # 2^p - 1 == 0
    monkeypatch.setenv("MI355_HOST_CARRY", "1")
    with E(p, odd) as h:
        for name, d in cases.items():
            h.set_digits(0, d)
            assert h.get_int(0) == value(d) and h.res64(0) == value(d) & (2**64 - 1), name


def test_automatic_radix_behind_the_plain_crt_spec():
    """fft_spec "crt" / "crt:auto": radix by the reference's size-ratio gates (README.md:888-926)"""
    for p, odd, n in [(9941, 1, 256), (86243, 9, 2304), (216091, 3, 6144)]:
        from prmers_amd import resolve_plan
        assert resolve_plan(p, "crt") == "crt-hip:n=%d:odd=%d" % (n, odd), resolve_plan(p, "crt")
        with CrtEngine(p, None) as e:
            assert (e.odd, e.n) == (odd, n)
            e.set(0, 3)
            for _ in range(40):
                e.square_mul(0)
            assert e.get_int(0) == pow(3, 1 << 40, (1 << p) - 1)


def test_sizes_with_too_few_bits_per_word_are_refused():
    from prmers_amd import EngineError
    with pytest.raises(EngineError, match="bits per word"):
        CrtEngine(216091, 9, 9 << 15)


def test_set_sub_and_all_ones():
    p, odd = 9941, 9
    with CrtEngine(p, odd) as e:
        Mp = (1 << p) - 1
        e.set(0, 5); assert e.get_int(0) == 5
        e.sub(0, 7); assert e.get_int(0) == Mp - 2        # borrow through every digit and around
        e.square_mul(0, 1); assert e.get_int(0) == 4
        o = orc_crt.OracleCrt(p, odd)
        ones = (np.uint64(1) << o.widths().astype(np.uint64)) - np.uint64(1)
        e.set_digits(0, ones)
        assert e.get_int(0) == 0 and e.res64(0) == 0      # 2^p - 1 reads as zero
        e.square_mul(0, 3); assert e.get_int(0) == 0


@pytest.mark.parametrize("odd,n", [(9, 9 << 20), (3, 3 << 21)])
def test_pfa_sizes_of_config_4_against_the_gmp_pins_and_the_oracle(odd, n):
    """p = 205271257 at the forced radix-9 size 9 * 2^20 and the automatic radix-3 size 3 * 2^21 (BASELINE configs[3], README.md:907-926):
    3^(2^k) against the libgmp pins, and one seeded squaring against the oracle"""
    p = 205271257
    import hashlib
    pins = {c["iteration"]: c for c in json.load(open(os.path.join(HERE, "golden", "big_p_pins.json")))["pins"][str(p)]}
    with CrtEngine(p, odd, n) as e:
        assert e.n == n
        e.set(0, 3)
        for k in range(1, max(pins) + 1):
            e.square_mul(0, 1)
            if k in pins:
                assert e.res64(0) == int(pins[k]["res64"], 16), k
                assert hashlib.sha256(e.words(0).astype("<u4").tobytes()).hexdigest() == pins[k]["sha256_words"], k
        o = orc_crt.OracleCrt(p, odd, n)
        rng = np.random.default_rng(1)
        w = o.widths().astype(np.uint64)
        start = rng.integers(0, 1 << 62, n, dtype=np.uint64) & ((np.uint64(1) << w) - np.uint64(1))
        o.set_digits(start); e.set_digits(0, start)
        o.square_mul(3); e.square_mul(0, 3)
        assert np.array_equal(e.raw_digits(0), o.digits())
        total, per = e.time_square_mul(0, 20)
        print("crt engine p=%d %s: %.4f ms/iter (event-bracketed stages) %s" % (p, e.describe(), total / 20, {k: round(v, 4) for k, v in per.items()}))


def test_headline_exponent_on_the_crt_family_against_the_gmp_pins():
    """p = 136279841 (BASELINE configs[2]) at 2^22 words of 32.5 bits, columns of 2048: 3^(2^k) against the libgmp pins"""
    import hashlib
    p = 136279841
    pins = {c["iteration"]: c for c in json.load(open(os.path.join(HERE, "golden", "big_p_pins.json")))["pins"][str(p)]}
    with CrtEngine(p, 1, reg_count=1) as e:
        assert e.n == 1 << 22 and "h1=2048" in e.describe()
        e.set(0, 3)
        for k in range(1, max(pins) + 1):
            e.square_mul(0, 1)
            if k in pins:
                assert e.res64(0) == int(pins[k]["res64"], 16), k
                assert hashlib.sha256(e.words(0).astype("<u4").tobytes()).hexdigest() == pins[k]["sha256_words"], k
