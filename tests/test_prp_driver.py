"""The PRP / LL driver (prmers_amd/prp.py), i.e. the caller side of the hot path: Gerbicz-Li check,
fault injection and rollback, checkpoints, worktodo sharding.  CPU tests drive it over the oracle;
the GPU tests drive the very same code over the HIP engine."""
import json
import os

import numpy as np
import pytest

import orc
from prmers_amd import prp

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


def make_engine(kind, p):
    if kind == "gpu":
        from prmers_amd import Engine
        return Engine(p, prp.REGISTERS)
    return orc.OracleEngine(p, prp.REGISTERS)


KINDS = ["oracle", pytest.param("gpu", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("p", [127, 521, 1279])
def test_prp_prime_with_gerbicz(kind, p):
    msgs = []
    with make_engine(kind, p) as e:
        r = prp.run_prp_or_ll(e, p, "prp", log=msgs.append, checklevel=1)
    assert r["is_prime"] and r["complete"] and r["res64"] == "0000000000000001"
    assert r["gerbicz_errors"] == 0 and r["gerbicz_checks"] >= 1
    assert any(m.startswith("[Gerbicz Li] Check passed!") for m in msgs)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("p,prime", [(127, True), (521, True), (1001, False), (607, True), (91, False)])
def test_ll_unsafe(kind, p, prime):
    """x -> x^2 - 2 from 4, p-2 iterations, prime iff 0 or Mp (RunPrpOrLlMarin.cpp:245,321-324,449-450)."""
    with make_engine(kind, p) as e:
        r = prp.run_prp_or_ll(e, p, "ll")
    s, M = 4, (1 << p) - 1
    for _ in range(p - 2):
        s = (s * s - 2) % M
    assert r["is_prime"] == prime == (s == 0)
    if prime:   # 0 may come out as the all-ones vector 2^p-1 (engine.h:286-295; RunPrpOrLlMarin.cpp:449-450)
        assert r["res64"] in ("0" * 16, "F" * 16)
    else:
        assert r["res64"] == "%016X" % (s & (2**64 - 1))


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("erroriter", GOLD["gerbicz_error_injection"]["erroriters"])
def test_gerbicz_error_injection_golden(kind, erroriter):
    """unit_tests.sh:24-50: `prmers 9941 -erroriter N` must print the failed check at iter 9941, restore
    iter=0 (j=9940), and still finish with the right answer."""
    g = GOLD["gerbicz_error_injection"]
    msgs = []
    with make_engine(kind, g["p"]) as e:
        r = prp.run_prp_or_ll(e, g["p"], "prp", erroriter=erroriter, log=msgs.append)
    assert "Injected error at iteration %d" % erroriter in msgs
    assert "[Gerbicz Li] Check FAILED! iter=%d" % g["failed_at_iter"] in msgs
    assert "[Gerbicz Li] Restore iter=%d (j=%d)" % (g["restore_iter"], g["restore_j"]) in msgs
    assert r["gerbicz_errors"] == 1 and r["is_prime"] and r["complete"]


@pytest.mark.parametrize("kind", KINDS)
def test_m11213_type1_residue(kind):
    with make_engine(kind, 11213) as e:
        r = prp.run_prp_or_ll(e, 11213, "prp")
    assert r["is_prime"] and r["res64"] == GOLD["m11213_final"]["res64"] and r["res2048"] == "0" * 511 + "1"


@pytest.mark.parametrize("kind", KINDS)
def test_checkpoint_roundtrip_and_resume(kind, tmp_path):
    p = 1279
    path = prp.checkpoint_name(p, "prp", str(tmp_path))
    with make_engine(kind, p) as e:
        part = prp.run_prp_or_ll(e, p, "prp", max_iters=700, ckpt_path=path, backup_every=100, gerbicz=False)
        assert not part["complete"] and part["iterations"] == 700
    assert os.path.exists(path) and os.path.exists(path + ".old")
    with make_engine(kind, p) as e:
        msgs = []
        r = prp.run_prp_or_ll(e, p, "prp", ckpt_path=path, gerbicz=False, log=msgs.append)
    assert "Resuming from a checkpoint." in msgs and r["is_prime"] and r["complete"]
    # corrupt -> ignored (file.h:104-111), other mode -> ignored (RunPrpOrLlMarin.cpp:166-170)
    raw = bytearray(open(path, "rb").read()); raw[40] ^= 1; open(path, "wb").write(bytes(raw))
    with make_engine(kind, p) as e:
        assert prp.load_checkpoint(path, e, p, "prp") is None
        assert prp.load_checkpoint(path + ".old", e, p, "ll") is None
        assert prp.load_checkpoint(path + ".old", e, p, "prp") is not None


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("side_file", [True, False])
def test_failed_check_after_a_resume_rolls_back_to_the_saved_point(kind, side_file, tmp_path):
    """A checkpoint written in the middle of a Gerbicz-Li block, a resume, then a fault: the rollback must land on the
    (iteration, R4 / R5) pair the checkpoint was saved with (RunPrpOrLlMarin.cpp:251-255 reloads itersave / jsave) --
    or, without the side file, on the resumed state itself -- and the run must still end on the known residue."""
    p = 9941
    path = prp.checkpoint_name(p, "prp", str(tmp_path))
    with make_engine(kind, p) as e:
        part = prp.run_prp_or_ll(e, p, "prp", checklevel=1, max_iters=5050, ckpt_path=path, backup_every=1010)
    assert not part["complete"] and part["gerbicz_errors"] == 0
    assert prp.load_gerbicz_state(path, 5050) is not None
    itersave = prp.load_gerbicz_state(path, 5050)[0]
    assert 0 < itersave < 5050
    if not side_file:
        os.remove(prp.gerbicz_state_name(path))
    msgs = []
    with make_engine(kind, p) as e:
        r = prp.run_prp_or_ll(e, p, "prp", checklevel=1, ckpt_path=path, erroriter=5060, log=msgs.append)
    assert "Resuming from a checkpoint." in msgs and "Injected error at iteration 5060" in msgs
    restored = [m for m in msgs if m.startswith("[Gerbicz Li] Restore")]
    assert len(restored) == 1
    assert restored[0].startswith("[Gerbicz Li] Restore iter=%d " % (itersave if side_file else 5049))
    assert r["gerbicz_errors"] == 1 and r["complete"] and r["is_prime"] and r["res64"] == "0000000000000001"
    # no squaring is done twice beyond the rolled-back block
    assert r["iterations"] == p


def test_resume_from_the_older_generation_keeps_its_rollback_point(tmp_path):
    """the rollback side file rotates with the checkpoint.  Checkpoints at 4040 and 5050; the main file is then torn, so the
    run resumes from <ckpt>.old (iteration 4040) -- and a check that fails right after must roll back to the block that generation was
    verified from (the point stored in <ckpt>.gl.old), not to the resumed mid-block state."""
    p = 9941
    path = prp.checkpoint_name(p, "prp", str(tmp_path))
    e = orc.OracleEngine(p, prp.REGISTERS)
    prp.run_prp_or_ll(e, p, "prp", checklevel=1, max_iters=5050, ckpt_path=path, backup_every=1010)
    assert os.path.exists(prp.gerbicz_state_name(path) + ".old")
    newer, older = prp.load_gerbicz_state(path, 5050), prp.load_gerbicz_state(path, 4040)
    assert newer is not None and older is not None and older[0] < 4040 < newer[0] < 5050
    with open(path, "r+b") as f:      # tear the main checkpoint: its CRC no longer matches
        f.seek(100); f.write(b"\xff" * 8)
    msgs = []
    e = orc.OracleEngine(p, prp.REGISTERS)
    r = prp.run_prp_or_ll(e, p, "prp", checklevel=1, ckpt_path=path, erroriter=4050, log=msgs.append)
    assert "Resuming from a checkpoint." in msgs and "Injected error at iteration 4050" in msgs
    restored = [m for m in msgs if m.startswith("[Gerbicz Li] Restore")]
    assert len(restored) == 1 and restored[0].startswith("[Gerbicz Li] Restore iter=%d " % older[0]), restored
    assert r["gerbicz_errors"] == 1 and r["complete"] and r["is_prime"] and r["res64"] == "0000000000000001"


@pytest.mark.parametrize("kind", KINDS)
def test_interrupt_checkpoints_and_the_resume_reaches_the_golden_residue(kind, tmp_path):
    """SIGINT path of the reference (RunPrpOrLlMarin.cpp:296-309): state saved at the iteration the flag is seen, clean
    return; the resumed run ends on the reference's M11213 residue."""
    p = 11213
    path = prp.checkpoint_name(p, "prp", str(tmp_path))
    polls = [0]

    def stop():   # polled before every run of plain iterations (at most 256 of them: prp.py batches them into one engine call)
        polls[0] += 1
        return polls[0] > 40
    msgs = []
    with make_engine(kind, p) as e:
        part = prp.run_prp_or_ll(e, p, "prp", ckpt_path=path, should_stop=stop, log=msgs.append)
    at = part["iterations"]
    assert part["interrupted"] and not part["complete"] and 40 <= at <= 40 * 256
    assert "Interrupted, state saved at iteration %d j=%d" % (at, p - at - 1) in msgs
    with make_engine(kind, p) as e:
        r = prp.run_prp_or_ll(e, p, "prp", ckpt_path=path)
    assert r["complete"] and r["is_prime"] and r["res64"] == GOLD["m11213_final"]["res64"] and r["gerbicz_errors"] == 0


def test_a_run_that_keeps_failing_its_checks_gives_up():
    p = 127

    class Liar(orc.OracleEngine):
        def is_equal(self, a, b):
            return False
    with Liar(p, prp.REGISTERS) as e:
        with pytest.raises(RuntimeError, match="giving up"):
            prp.run_prp_or_ll(e, p, "prp", checklevel=1)


def test_worktodo_parsing_and_sharding():
    lines = ["# comment", "PRP=1,2,136279841,-1", "PRP=0123456789ABCDEF0123456789ABCDEF,1,2,136279879,-1,76,0",
             "Test=136279901", "DoubleCheck=0123456789abcdef0123456789abcdef,85473391,76,1", "Test=1,2,127,-1",
             "PFactor=1,2,100003,-1,70,2", "garbage", "PRPDC=N/A,1,2,521,-1"]
    got = [prp.parse_worktodo_line(l) for l in lines]
    assert got == [None, ("prp", 136279841), ("prp", 136279879), ("ll", 136279901), ("ll", 85473391), ("ll", 127),
                   None, None, ("prp", 521)]
    assert prp.shard_worktodo(lines, 0, 2) == [("prp", 136279841), ("ll", 136279901), ("ll", 127)]
    assert prp.shard_worktodo(lines, 1, 2) == [("prp", 136279879), ("ll", 85473391), ("prp", 521)]


def test_residue_formatting_matches_oracle():
    p = 9941
    o = orc.Oracle(p, 1)
    o.set_value(0, 3**500)
    d = o.digits(0)
    assert np.array_equal(prp.pack_words(d, p), orc.pack_words(d, p))
    w = prp.prp3_div9(p, prp.pack_words(d, p))
    assert (prp.format_res64(w), prp.format_res2048(w)) == orc.prp_type1_hex(d, p)
    assert prp.digits_equal_to(d, 3**500 % ((1 << p) - 1)) and not prp.digits_equal_to(d, 9)


@pytest.mark.gpu
def test_c2_9815459_partial_run_with_gerbicz_check():
    """BASELINE config C2: PRP p=9815459 (n=2^19) with the Gerbicz-Li check on: the first 2*B+5 iterations
    of the real schedule plus a forced check, compared with the oracle's residue."""
    p = 9815459
    B = int(p ** 0.5)
    msgs = []
    from prmers_amd import Engine
    with Engine(p, prp.REGISTERS) as e:
        # j % B == 0 happens every B iterations; checklevel=1 verifies at each of them
        r = prp.run_prp_or_ll(e, p, "prp", checklevel=1, max_iters=2 * B + 5, log=msgs.append)
        assert r["gerbicz_errors"] == 0 and r["gerbicz_checks"] >= 2, msgs
        o = orc.Oracle(p, 1)
        o.set(0, 3)
        for _ in range(r["iterations"]):
            o.square_mul(0)
        assert np.array_equal(e.digits(prp.R0), o.digits(0))


@pytest.mark.gpu
@pytest.mark.parametrize("p", [57885161, 136279841])
def test_gerbicz_blocks_at_register_resident_shapes(p):
    """n = 2^22 and C3 (n = 2^23), both served end to end by the register-resident kernels: the first two
    Gerbicz-Li blocks of the real PRP schedule (2*B + 5 squarings of a full-size residue plus the two block
    verifications, ~45 000 squarings at C3) must pass their checks, and an injected error must be caught."""
    B = int(p ** 0.5)
    from prmers_amd import Engine
    msgs = []
    with Engine(p, prp.REGISTERS) as e:
        r = prp.run_prp_or_ll(e, p, "prp", checklevel=1, max_iters=2 * B + 5, log=msgs.append)
        assert r["gerbicz_errors"] == 0 and r["gerbicz_checks"] >= 2, msgs[-5:]
    msgs = []
    with Engine(p, prp.REGISTERS) as e:
        r = prp.run_prp_or_ll(e, p, "prp", checklevel=1, max_iters=B + 5, erroriter=100, log=msgs.append)
        assert r["gerbicz_errors"] >= 1 and any("Check FAILED" in m for m in msgs), msgs[-5:]


@pytest.mark.gpu
def test_full_prp_of_m216091_on_gpu():
    """a complete PRP of a known Mersenne prime beyond the reference's unit-test list (n = 10240, radix-5
    shape): 216091 squarings with the Gerbicz-Li check on end on residue 9, type-1 res64 = 1."""
    from prmers_amd import Engine
    with Engine(216091, prp.REGISTERS) as e:
        r = prp.run_prp_or_ll(e, 216091, "prp")
    assert r["complete"] and r["is_prime"] and r["res64"] == "0000000000000001" and r["gerbicz_errors"] == 0


@pytest.mark.gpu
def test_prp_blocks_of_m859433_on_the_radix5_columns_of_2560():
    """the Mersenne prime exponent 859433 (n = 5 2^13) with the transform forced onto the columns of 2560 = 5 x 512 with runs of two pairs
    (kernels_v5.hip, J = 1: the column kernels of n = 5 2^22 and 5 2^23): the first 150 000 squarings of its PRP with the Gerbicz-Li check on,
    an injected error caught and repaired on the way.  (The complete PRP -- 859433 squarings, residue 9, one injected error repaired -- ran
    on the same kernels in 73 s: profiles/r04_soak_c.txt.)"""
    from prmers_amd import Engine
    p = 859433
    with Engine(p, prp.REGISTERS, plan="m2=8,c=2") as e:
        assert e.describe().startswith("marin-hip:n=40960:m1=2560:m2=8:c=2")
        msgs = []
        r = prp.run_prp_or_ll(e, p, "prp", checklevel=1, max_iters=150000, erroriter=60000, log=msgs.append)
    assert r["gerbicz_errors"] == 1 and r["gerbicz_checks"] >= 2 and any("Check FAILED" in m for m in msgs), msgs[-5:]


def test_result_json_shape():
    """keys and order of the reference's PRP / LL result JSON (src/io/JsonBuilder.cpp:396-441)."""
    with orc.OracleEngine(127, prp.REGISTERS) as e:
        r = prp.run_prp_or_ll(e, 127, "prp")
        js = prp.result_json(r, e.get_size())
    assert js.startswith('{"status":"P","exponent":127,"worktype":"PRP-3","res64":"0000000000000001","res2048":"')
    d = json.loads(js)
    assert list(d)[:9] == ["status", "exponent", "worktype", "res64", "res2048", "residue-type", "errors", "shift-count", "fft-length"]
    assert d["errors"] == {"gerbicz": 0} and d["fft-length"] == 8 and d["residue-type"] == 1
    with orc.OracleEngine(1001, prp.REGISTERS) as e:
        r = prp.run_prp_or_ll(e, 1001, "ll")
        d = json.loads(prp.result_json(r, e.get_size()))
    assert d["status"] == "C" and d["worktype"] == "LL" and "res2048" not in d


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("p,prime", [(127, True), (521, True), (607, True), (1001, False), (2203, True)])
def test_ll_safe_block_recomputation(kind, p, prime):
    """LL-safe (RunLlSafeMarin.cpp:96-360): V, U = prod V, block re-computation check; answer vs Python ints."""
    msgs = []
    with make_engine(kind, p) as e:
        r = prp.run_ll_safe(e, p, log=msgs.append)
    s, M = 4, (1 << p) - 1
    for _ in range(p - 2):
        s = (s * s - 2) % M
    assert r["complete"] and r["is_prime"] == prime == (s == 0) and r["errors"] == 0 and r["checks"] >= 1
    assert r["res64"] == "%016X" % (s & (2**64 - 1))
    assert sum(m.startswith("[Error check] Check passed!") for m in msgs) == r["checks"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("erroriter", [1, 50, 300, 1277])
def test_ll_safe_detects_and_repairs_injected_error(kind, erroriter):
    """-erroriter N in LL-safe mode: the block check fails once, the state rolls back to the block start
    and the run still ends on the right answer (M1279 is prime)."""
    p, msgs = 1279, []
    with make_engine(kind, p) as e:
        r = prp.run_ll_safe(e, p, erroriter=erroriter, log=msgs.append)
    B = int(p / (p ** 0.5))
    blk_start = ((erroriter - 1) // B) * B
    assert r["is_prime"] and r["errors"] == 1 and r["res64"] == "0" * 16
    assert "Injected error at iteration %d" % erroriter in msgs
    assert "[Error check] Restore iter=%d" % blk_start in msgs


def make_engine_n(kind, p, regs):
    if kind == "gpu":
        from prmers_amd import Engine
        return Engine(p, regs)
    return orc.OracleEngine(p, regs)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("p,prime", [(127, True), (521, True), (607, True), (1001, False), (1279, True)])
def test_ll_safe2_pairs_with_gerbicz(kind, p, prime):
    """second LL-safe mode (RunLlSafeMarin.cpp:394-728): (2 + sqrt 3)^(2^(p-1)) in pairs with the Gerbicz-Li check;
    primality and the reported LL residue S_(p-2) against Python integers."""
    msgs = []
    with make_engine_n(kind, p, prp.LLSAFE2_REGISTERS) as e:
        r = prp.run_ll_safe2(e, p, checklevel=1, log=msgs.append)
    s, M = 4, (1 << p) - 1
    for _ in range(p - 2):
        s = (s * s - 2) % M
    assert r["complete"] and r["is_prime"] == prime == (s == 0) and r["gerbicz_errors"] == 0 and r["gerbicz_checks"] >= 1
    if prime:   # 0 may come out as the all-ones vector 2^p-1 (engine.h:286-295), as in the LL-unsafe mode
        assert r["res64"] in ("0" * 16, "F" * 16)
    else:
        assert r["res64"] == "%016X" % (s & (2**64 - 1))
    assert any(m.startswith("[Gerbicz-Li] Check OK") for m in msgs)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("erroriter", [1, 40, 700, 1278])
def test_ll_safe2_detects_and_repairs_injected_error(kind, erroriter):
    p, msgs = 1279, []
    with make_engine_n(kind, p, prp.LLSAFE2_REGISTERS) as e:
        r = prp.run_ll_safe2(e, p, erroriter=erroriter, checklevel=1, log=msgs.append)
    assert r["is_prime"] and r["gerbicz_errors"] == 1 and r["res64"] in ("0" * 16, "F" * 16)
    assert any("Check FAILED" in m for m in msgs) and any(m.startswith("[Gerbicz-Li] Restore iter=") for m in msgs)


def test_prp_in_slices_resumes_after_a_passed_check():
    """run_prp_or_ll(resume=..., stop_after_s=...): a PRP of M9941 cut into three slices, each continued on a
    fresh engine from (residue, Gerbicz accumulator, iteration) saved right after a passed check, ends on the
    same answer as an uninterrupted run (tools/long_prp.py uses this for runs longer than one GPU-box call)."""
    p = 9941
    state, saved = None, None
    for slice_no in range(3):
        with orc.OracleEngine(p, prp.REGISTERS) as e:
            if state is not None:
                e.o.set_digits(prp.R0, saved[0]); e.o.set_digits(prp.R1, saved[1])
            r = prp.run_prp_or_ll(e, p, "prp", checklevel=1, resume=state, stop_after_s=None if slice_no == 2 else 0.0)
            if slice_no < 2:
                assert r["state"] is not None and not r["complete"] and r["gerbicz_errors"] == 0
                state, saved = r["state"], (e.digits(prp.R0), e.digits(prp.R1))
    assert r["complete"] and r["is_prime"] and r["res64"] == "0000000000000001" and r["gerbicz_errors"] == 0
    assert state["it"] > 100   # the third slice really started in the middle
