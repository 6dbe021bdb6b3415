"""PRP / Lucas-Lehmer drivers over the `engine` register machine -- the callers of the hot path.

Mirror of the reference's Marin-path driver (src/modes/RunPrpOrLlMarin.cpp:97-777): base-3 PRP with
the Gerbicz-Li check and rollback (:338-412), the `-erroriter` fault injection (:326-336), LL-unsafe
(x^2-2, :321-324), checkpoint files (:156-211 with the CRC of include/marin/file.h:49-111), result
formatting (include/core/AlgoUtils.hpp:165-223) and the worktodo line formats
(src/io/WorktodoParser.cpp:78-400) used to shard exponents one per GPU; plus the LL-safe
driver with block re-computation (src/modes/RunLlSafeMarin.cpp:96-360).

Everything here talks to an engine only through its interface (set / copy / square_mul / sub /
set_multiplicand / mul / get_int / digits / get_checkpoint / set_checkpoint), so the same driver runs
on prmers_amd.Engine (MI355X) and, in the CPU tests, on an oracle-backed stand-in.
"""
import math
import os
import struct
import zlib

import numpy as np

# register roles of the reference driver (RunPrpOrLlMarin.cpp:212)
R0, R1, R2, R3, R4, R5, RBASE, RTMP = range(8)
REGISTERS = 8
CKPT_BACKEND_ID = 3      # 1 = Marin/OpenCL, 2 = Aevum (RunPrpOrLlMarin.cpp:154); images are backend-specific


# ---------------------------------------------------------------------------------------------
# residue formatting (AlgoUtils.hpp:165-223, engine.h:257-295)
# ---------------------------------------------------------------------------------------------
def pack_words(digits, p):
    """little-endian 32-bit words of an encoded digit vector (value | width << 32); the digits are canonical
    (value < 2^width), so their bit fields are disjoint and can be scattered with two vector adds."""
    d = np.asarray(digits, dtype=np.uint64)
    w = d >> np.uint64(32)
    v = d & ((np.uint64(1) << w) - np.uint64(1)) & np.uint64(0xFFFFFFFF)
    off = np.concatenate(([0], np.cumsum(w, dtype=np.uint64)[:-1])).astype(np.uint64)
    wc = (p + 31) // 32
    acc = np.zeros(wc + 2, dtype=np.uint64)
    sh = v << (off & np.uint64(31))                      # < 2^63: a digit spans at most two words
    idx = (off >> np.uint64(5)).astype(np.int64)
    np.add.at(acc, idx, sh & np.uint64(0xFFFFFFFF))
    np.add.at(acc, idx + 1, sh >> np.uint64(32))
    return acc[:wc].astype("<u4")


def _div3(p, x):
    """x / 3 mod 2^p-1 for odd p: (x + r (2^p - 1)) / 3 with r = -x mod 3 (2^p - 1 = 1 mod 3); the reference does
    the same division word by word (AlgoUtils.hpp:180-203)."""
    r = (-x) % 3
    return (x + r * ((1 << p) - 1)) // 3


def prp3_div9(p, W):
    """type-1 residue: divide 3^(2^p) by 9 mod 2^p-1 on the word vector (AlgoUtils.hpp:204-210)."""
    wc = (p + 31) // 32
    x = int.from_bytes(np.ascontiguousarray(W, dtype="<u4").tobytes(), "little")
    x = _div3(p, _div3(p, x))
    return np.frombuffer(x.to_bytes(wc * 4, "little"), dtype="<u4").copy()


def format_res64(W):
    return "%016X" % ((int(W[1]) << 32 if len(W) > 1 else 0) | (int(W[0]) if len(W) else 0))


def format_res2048(W):
    return "".join("%08x" % (int(W[i]) if i < len(W) else 0) for i in range(63, -1, -1))


def digits_equal_to(digits, a):
    """engine::digit::equal_to (engine.h:272-284): the digit vector spells the small constant a."""
    d = np.asarray(digits, dtype=np.uint64)
    w = d >> np.uint64(32)
    expect = np.zeros(d.size, dtype=np.uint64)
    r, k = int(a), 0
    while r and k < d.size:
        wk = int(w[k])
        expect[k] = r & ((1 << wk) - 1)
        r >>= wk
        k += 1
    return r == 0 and bool(np.array_equal(d & np.uint64(0xFFFFFFFF), expect))


def digits_equal_to_Mp(digits):
    d = np.asarray(digits, dtype=np.uint64)
    w = d >> np.uint64(32)
    return bool(np.all((d & np.uint64(0xFFFFFFFF)) == (np.uint64(1) << w) - np.uint64(1)))


# ---------------------------------------------------------------------------------------------
# checkpoint file: version 2 layout of RunPrpOrLlMarin.cpp:190-206, CRC trailer of file.h:98-111
# ---------------------------------------------------------------------------------------------
def checkpoint_name(p, mode, directory="."):
    return os.path.join(directory, ("llunsafe_" if mode == "ll" else "") + "m_%d.ckpt" % p)


def save_checkpoint(path, eng, p, mode, it, elapsed):
    data = np.asarray(eng.get_checkpoint(), dtype=np.uint8).tobytes()
    body = struct.pack("<iIIIId", 2, p, 1 if mode == "prp" else 2, CKPT_BACKEND_ID, it, elapsed) + data
    crc = zlib.crc32(body) & 0xFFFFFFFF
    trailer = struct.pack("<I", (~crc & 0xFFFFFFFF) ^ 0xA23777AC)
    new, old = path + ".new", path + ".old"
    with open(new, "wb") as f:
        f.write(body + trailer)
    if os.path.exists(old):
        os.remove(old)
    if os.path.exists(path):
        os.rename(path, old)
    os.rename(new, path)


def load_checkpoint(path, eng, p, mode):
    """-> (iter, elapsed) or None (missing / other exponent, mode or backend / bad CRC)."""
    try:
        raw = open(path, "rb").read()
    except OSError:
        return None
    head = struct.calcsize("<iIIIId")
    if len(raw) < head + 4:
        return None
    version, rp, smode, sbackend, it, elapsed = struct.unpack("<iIIIId", raw[:head])
    if version != 2 or rp != p or smode != (1 if mode == "prp" else 2) or sbackend != CKPT_BACKEND_ID:
        return None
    body, trailer = raw[:-4], struct.unpack("<I", raw[-4:])[0]
    if trailer != ((~zlib.crc32(body) & 0xFFFFFFFF) ^ 0xA23777AC):
        return None
    data = np.frombuffer(body[head:], dtype=np.uint8)
    if data.size != eng.get_checkpoint_size() or not eng.set_checkpoint(data):
        return None
    return it, elapsed


# Gerbicz-Li rollback point next to the checkpoint (the reference keeps itersave / jsave in its backup manager and reloads
# them on a resume, RunPrpOrLlMarin.cpp:251-255): iteration of the checkpoint they belong to, itersave, jsave, checkpass, CRC.
GL_MAGIC = 0x474C3352   # "GL3R"


def gerbicz_state_name(ckpt_path):
    return ckpt_path + ".gl"


def save_gerbicz_state(ckpt_path, it, itersave, jsave, checkpass):
    body = struct.pack("<IIQQQ", GL_MAGIC, it, itersave, jsave, checkpass)
    name = gerbicz_state_name(ckpt_path)
    tmp = name + ".new"
    with open(tmp, "wb") as f:
        f.write(body + struct.pack("<I", zlib.crc32(body) & 0xFFFFFFFF))
    # the side file rotates with the checkpoint (<ckpt>.gl.old belongs to <ckpt>.old): a resume from the older generation -- a torn main
    # checkpoint, or a crash between the renames -- still finds the rollback point that generation was verified from
    if os.path.exists(name):
        os.replace(name, name + ".old")
    os.replace(tmp, name)


def load_gerbicz_state(ckpt_path, it):
    """-> (itersave, jsave, checkpass) from whichever generation of the side file belongs to the checkpoint of iteration `it`, else None."""
    n = struct.calcsize("<IIQQQ")
    for name in (gerbicz_state_name(ckpt_path), gerbicz_state_name(ckpt_path) + ".old"):
        try:
            raw = open(name, "rb").read()
        except OSError:
            continue
        if len(raw) != n + 4 or struct.unpack("<I", raw[n:])[0] != (zlib.crc32(raw[:n]) & 0xFFFFFFFF):
            continue
        magic, sit, itersave, jsave, checkpass = struct.unpack("<IIQQQ", raw[:n])
        if magic == GL_MAGIC and sit == it:
            return int(itersave), int(jsave), int(checkpass)
    return None


MAX_GERBICZ_ERRORS = 64   # a run that keeps failing its checks is stopped (hardware that is not fit for the job)


# ---------------------------------------------------------------------------------------------
# worktodo (WorktodoParser.cpp:78-400): only what the one-exponent-per-GPU sharder needs
# ---------------------------------------------------------------------------------------------
def parse_worktodo_line(line):
    """-> (mode, exponent) for PRP= / PRPDC= / Test= / DoubleCheck= lines, else None."""
    line = line.strip()
    if not line or line[0] == "#" or "=" not in line:
        return None
    key, rest = line.split("=", 1)
    key = key.strip().upper()
    if key in ("PRP", "PRPDC"):
        mode = "prp"
    elif key in ("TEST", "DOUBLECHECK"):
        mode = "ll"
    else:
        return None
    parts = [x.strip() for x in rest.split(",")]
    if parts and (parts[0] == "" or parts[0] == "N/A"):
        parts = parts[1:]
    if parts and (parts[0] in ("AID", "N/A") or (len(parts[0]) == 32 and all(c in "0123456789abcdefABCDEF" for c in parts[0]))):
        parts = parts[1:]
    if len(parts) >= 4 and parts[0] == "1" and parts[1] == "2" and parts[3] == "-1":
        return mode, int(parts[2])           # k,b,n,c form: 1*2^n-1
    if mode == "ll" and parts and parts[0].isdigit():
        return mode, int(parts[0])           # Test=exponent[,how_far_factored[,pm1]]
    return None


def shard_worktodo(lines, rank, world):
    """entries handled by `rank`: line i -> GPU i mod world."""
    entries = [e for e in (parse_worktodo_line(l) for l in lines) if e]
    return entries[rank::world]


# ---------------------------------------------------------------------------------------------
# the driver
# ---------------------------------------------------------------------------------------------
def _square_n(eng, reg, count, sub=0):
    """count x { reg = reg^2; reg -= sub }: one engine call where the engine has square_mul_n, else the loop it stands for."""
    if count <= 0:
        return
    f = getattr(eng, "square_mul_n", None)
    if f is not None and count > 1:
        f(reg, count, 1, sub)
        return
    for _ in range(count):
        eng.square_mul(reg)
        if sub:
            eng.sub(reg, sub)


def run_prp_or_ll(eng, p, mode="prp", gerbicz=True, erroriter=0, checklevel=0, max_iters=None,
                  log=None, ckpt_path=None, backup_every=0, resume=None, stop_after_s=None, on_check=None, should_stop=None,
                  backup_interval_s=None):
    """One PRP (mode "prp") or LL-unsafe (mode "ll") test of 2^p-1 on `eng` (>= 8 registers).

    Returns a dict: is_prime, res64, res2048, iterations, gerbicz_checks, gerbicz_errors, complete.
    max_iters stops early (complete = False) for partial runs; the result fields then describe the
    residue reached.  log(msg) receives the reference's messages ("[Gerbicz Li] Check passed! iter=N").
    resume = {"it": i, "j": j}: continue a PRP right after the passed Gerbicz-Li check of iteration i (the
    caller has put the residue into R0 and the Gerbicz accumulator into R1); stop_after_s: stop at the first
    passed check after that many seconds and return that state in result["state"] (long runs in slices).
    on_check(passed, iteration) is called after every Gerbicz-Li check (the multi-GPU launcher reduces its status
    word there).
    backup_every (iterations) / backup_interval_s (seconds, the reference's -t, RunPrpOrLlMarin.cpp:311-319): checkpoint cadence.
    should_stop(): polled before every iteration; when it returns true the state is checkpointed (ckpt_path) and the run
    returns with complete = False and interrupted = True -- the reference's SIGINT path (RunPrpOrLlMarin.cpp:296-309).
    A checkpoint carries the Gerbicz-Li rollback point (R4 / R5 inside the register dump, itersave / jsave / checkpass in
    the side file): a check that fails after a resume rolls back to the block the checkpoint itself was verified from.
    """
    import time as _time
    t_start = _time.time()
    log = log or (lambda m: None)
    prp = mode == "prp"
    total = p if prp else p - 2
    ri = 0
    gl = None
    if ckpt_path:
        got = load_checkpoint(ckpt_path, eng, p, mode) or load_checkpoint(ckpt_path + ".old", eng, p, mode)
        if got:
            gl = load_gerbicz_state(ckpt_path, got[0])   # the side-file generation that names this checkpoint's iteration
            ri = got[0]
            log("Resuming from a checkpoint.")
    if ri == 0 and resume is None:
        eng.set(R1, 1)
        eng.set(R0, 3 if prp else 4)
    if gl is None:            # else R4 / R5 of the checkpoint are the state the rollback point names
        eng.copy(R4, R0)      # last state that passed a check
        eng.copy(R5, R1)
    eng.set(RBASE, 3)
    eng.set_multiplicand(RTMP, RBASE)

    B = max(int(math.sqrt(p)), 1)
    auto = int((1000 * 600.0) / B)
    if auto == 0:
        auto = (total // B) // max(int(math.sqrt(B)), 1)
    checkpasslevel = checklevel if checklevel > 0 else max(auto, 1)
    itersave, jsave = 0, total - 1
    checkpass = 0
    errordone = False
    checks = errors = 0
    done = 0
    it, j = ri, total - ri - 1
    if gl is not None:
        itersave, jsave, checkpass = gl
    elif ri > 0:
        # no rollback point on file: R4 / R5 hold the resumed state itself, so a rollback re-enters the loop at ri
        itersave, jsave = ri - 1, total - ri
    if resume is not None:
        itersave, jsave = int(resume["it"]), int(resume["j"])
        it, j = itersave + 1, jsave - 1
    state = None
    interrupted = False
    batched = hasattr(eng, "square_mul_n")

    last_backup = [t_start]

    def checkpoint(at):
        save_checkpoint(ckpt_path, eng, p, mode, at, _time.time() - t_start)
        save_gerbicz_state(ckpt_path, at, itersave, jsave, checkpass)
        last_backup[0] = _time.time()

    while it < total:
        if max_iters is not None and done >= max_iters:
            break
        if state is not None:
            break
        if should_stop is not None and should_stop():
            interrupted = True
            if ckpt_path:
                checkpoint(it)
            log("Interrupted, state saved at iteration %d j=%d" % (it, j))
            break
        # the run of plain iterations up to the next event (Gerbicz-Li boundary, injected error, checkpoint, stop poll) goes to the
        # engine as ONE call where it offers square_mul_n (one cooperative launch on the small transforms); `it` / `j` then name the
        # last iteration of the run, as they would after that many turns of the reference's loop (RunPrpOrLlMarin.cpp:338-409)
        r = 1
        if batched:
            r = (j % B + 1) if (prp and gerbicz) else (total - it)
            r = min(r, total - it, 65536)      # (a bound on one engine call: the loop stays responsive without Gerbicz-Li boundaries)
            if erroriter > 0 and not errordone and erroriter > it:
                r = min(r, erroriter - it)
            if max_iters is not None:
                r = min(r, max_iters - done)
            if ckpt_path and backup_every:
                r = min(r, backup_every - done % backup_every)
            if ckpt_path and backup_interval_s is not None:
                r = min(r, 256 - (done & 255))
            if should_stop is not None or stop_after_s is not None:
                r = min(r, 256)
            r = max(r, 1)
        _square_n(eng, R0, r, 0 if prp else 2)
        it += r - 1
        j -= r - 1
        done += r
        if erroriter > 0 and it + 1 == erroriter and not errordone:
            errordone = True
            eng.sub(R0, 2)
            log("Injected error at iteration %d" % (it + 1))
        if prp and gerbicz and ((j != 0 and j % B == 0) or it == total - 1):
            checkpass += 1
            eng.copy(R3, R1)
            eng.set_multiplicand(R2, R0)
            eng.mul(R1, R2)
            if not (checkpass != checkpasslevel and it != total - 1):
                checkpass = 0
                checks += 1
                modB = B if p % B == 0 else p % B
                _square_n(eng, R3, B - modB - 1 if B > modB else 0)
                if p % B == 0:
                    eng.mul(R3, RTMP)
                else:
                    eng.square_mul(R3, 3)
                _square_n(eng, R3, modB)
                # the reference compares two mpz read-backs (RunPrpOrLlMarin.cpp:363-366); here the engine compares the
                # canonical forms on the device (canon.hip), 16 bytes over PCIe
                same = eng.is_equal(R3, R1) if hasattr(eng, "is_equal") else eng.get_int(R3) == eng.get_int(R1)
                if on_check:
                    on_check(bool(same), it + 1)
                if not same:
                    log("[Gerbicz Li] Mismatch")
                    log("[Gerbicz Li] Check FAILED! iter=%d" % (it + 1))
                    log("[Gerbicz Li] Restore iter=%d (j=%d)" % (itersave, jsave))
                    j, it = jsave, itersave
                    if it == 0:
                        it -= 1
                        j += 1
                    errors += 1
                    if errors > MAX_GERBICZ_ERRORS:
                        raise RuntimeError("Gerbicz-Li check failed %d times: giving up on exponent %d" % (errors, p))
                    eng.copy(R0, R4)
                    eng.copy(R1, R5)
                else:
                    log("[Gerbicz Li] Check passed! iter=%d" % (it + 1))
                    eng.copy(R4, R0)
                    eng.copy(R5, R1)
                    itersave, jsave = it, j
                    if stop_after_s is not None and _time.time() - t_start > stop_after_s and it != total - 1:
                        state = {"it": it, "j": j}
        it += 1
        j -= 1
        if ckpt_path and it < total and ((backup_every and done % backup_every == 0) or
                                         (backup_interval_s is not None and (done & 255) == 0 and _time.time() - last_backup[0] >= backup_interval_s)):
            checkpoint(it)

    d = eng.digits(R0)
    if prp:
        is_prime = digits_equal_to(d, 9)
    else:
        is_prime = digits_equal_to(d, 0) or digits_equal_to_Mp(d)
    words = pack_words(d, p)
    if prp and pow(2, p, 9) != 1:      # 2^p-1 not divisible by 9 (RunPrpOrLlMarin.cpp:288-291,456)
        words = prp3_div9(p, words)
    return {"exponent": p, "mode": mode, "is_prime": bool(is_prime) and it >= total, "res64": format_res64(words),
            "res2048": format_res2048(words), "iterations": it, "gerbicz_checks": checks,
            "gerbicz_errors": errors, "complete": it >= total, "state": state, "interrupted": interrupted,
            "fft_length": int(getattr(eng, "n", 0))}


# register roles of the LL-safe driver (RunLlSafeMarin.cpp:20-28: V, U, their last good copies, the re-run copies)
LS_RV, LS_RU, LS_RVC, LS_RUC, LS_RVCHK, LS_RUCHK, LS_RTMP = range(7)


def run_ll_safe(eng, p, block=0, erroriter=0, max_iters=None, log=None):
    """Lucas-Lehmer with error detection by block re-computation -- mirror of LL-safe mode
    (RunLlSafeMarin.cpp:96-360).  V follows x -> x^2 - 2 from 4; U accumulates the product of the V's
    (set_multiplicand + mul per iteration, :257-260); every B = p / sqrt(p) iterations (or `block`, the
    reference's -llsafe_block) the block is recomputed from the last good (V, U) and both pairs must agree
    (:268-296), otherwise the state rolls back to the block start (:297-318).  erroriter injects V -= 2
    once, like the reference's -erroriter (:245-255).  Needs >= 8 registers."""
    log = log or (lambda m: None)
    total = p - 2 if p >= 2 else 0
    eng.set(LS_RV, 4)
    eng.set(LS_RU, 2)
    eng.copy(LS_RVC, LS_RV)
    eng.copy(LS_RUC, LS_RU)
    eng.copy(LS_RVCHK, LS_RVC)
    eng.copy(LS_RUCHK, LS_RUC)
    B = int(block) if block > 0 else int(p / math.sqrt(float(p)))
    B = min(max(B, 1), max(total, 1))

    def step(rv, ru):
        eng.set_multiplicand(LS_RTMP, rv)
        eng.mul(ru, LS_RTMP)
        eng.square_mul(rv)
        eng.sub(rv, 2)

    errordone = False
    itersave = 0
    checks = errors = done = 0
    it = 0
    while it < total:
        if max_iters is not None and done >= max_iters:
            break
        if erroriter > 0 and it + 1 == erroriter and not errordone:
            errordone = True
            eng.sub(LS_RV, 2)
            log("Injected error at iteration %d" % (it + 1))
        step(LS_RV, LS_RU)
        done += 1
        if (it + 1) % B == 0 or it + 1 == total:
            blk = B if (it + 1) % B == 0 else (it + 1) - itersave
            eng.copy(LS_RVCHK, LS_RVC)
            eng.copy(LS_RUCHK, LS_RUC)
            for _ in range(blk):
                step(LS_RVCHK, LS_RUCHK)
            checks += 1
            ok = eng.get_int(LS_RVCHK) == eng.get_int(LS_RV) and eng.get_int(LS_RUCHK) == eng.get_int(LS_RU)
            if not ok:
                log("[Error check] Mismatch")
                log("[Error check] Check FAILED! iter=%d" % it)
                log("[Error check] Restore iter=%d" % itersave)
                errors += 1
                eng.copy(LS_RV, LS_RVC)
                eng.copy(LS_RU, LS_RUC)
                it = itersave
                continue
            log("[Error check] Check passed! iter=%d" % it)
            eng.copy(LS_RVC, LS_RV)
            eng.copy(LS_RUC, LS_RU)
            itersave = it + 1
        it += 1

    d = eng.digits(LS_RV)
    is_mp = digits_equal_to_Mp(d)
    is_prime = (digits_equal_to(d, 0) or is_mp) and it >= total
    words = pack_words(d, p)
    if is_prime and is_mp:
        words[:] = 0          # the all-ones vector stands for 0 (RunLlSafeMarin.cpp:335-338)
    return {"exponent": p, "mode": "llsafe", "is_prime": bool(is_prime), "res64": format_res64(words),
            "res2048": format_res2048(words), "iterations": it, "checks": checks, "errors": errors,
            "complete": it >= total}


# register roles of the second LL-safe driver (RunLlSafeMarin.cpp:479): 18 registers
(L2_RES_A, L2_RES_B, L2_ACC_A, L2_ACC_B, L2_CHK_A, L2_CHK_B, L2_SAVE_R_A, L2_SAVE_R_B, L2_SAVE_F_A, L2_SAVE_F_B,
 L2_BASE_A, L2_BASE_B, L2_TA, L2_TB, L2_M0, L2_M1, L2_PREV_A, L2_PREV_B) = range(18)
LLSAFE2_REGISTERS = 18


def run_ll_safe2(eng, p, erroriter=0, checklevel=0, max_iters=None, log=None):
    """Lucas-Lehmer in Z[sqrt 3] with the Gerbicz-Li check -- mirror of the reference's second LL-safe mode
    (RunLlSafeMarin.cpp:394-728).  The residue is the pair (A, B) = A + B sqrt 3 = (2 + sqrt 3)^(2^k) mod 2^p-1,
    squared p-1 times from (2, 1) (pair_square :481-490: A <- A^2 + 3 B^2, B <- 2 A B); every B = floor(sqrt p)
    iterations the accumulator pair is multiplied by the residue (pair_mul_by :492-507) and every `checklevel`
    blocks the Gerbicz-Li identity is verified on a copy and rolled back on a mismatch (:611-660).  2^p-1 is prime
    iff the result is (-1, 0) (:664-671); the classic LL residue S_(p-2) = 2 A_(p-2) is reported (:674-680).
    Needs 18 registers (`add` is the only operation beyond the PRP set)."""
    log = log or (lambda m: None)
    total = p - 1 if p > 1 else 0

    def pair_square(A, Bq):
        eng.copy(L2_TA, A); eng.copy(L2_TB, Bq)
        eng.square_mul(A); eng.square_mul(Bq, 3); eng.add(A, Bq)
        eng.copy(Bq, L2_TA)
        eng.set_multiplicand(L2_M0, L2_TB)
        eng.mul(Bq, L2_M0, 2)

    def pair_mul_by(A, Bq, Cq, D):
        eng.set_multiplicand(L2_M0, Cq); eng.set_multiplicand(L2_M1, D)
        eng.copy(L2_TA, A); eng.copy(L2_TB, Bq)
        eng.copy(A, L2_TA); eng.mul(A, L2_M0)
        eng.copy(Bq, L2_TA); eng.mul(Bq, L2_M1)
        eng.copy(L2_TA, L2_TB); eng.mul(L2_TA, L2_M1, 3); eng.add(A, L2_TA)
        eng.copy(L2_TA, L2_TB); eng.mul(L2_TA, L2_M0); eng.add(Bq, L2_TA)

    eng.set(L2_RES_A, 2); eng.set(L2_RES_B, 1)
    eng.set(L2_ACC_A, 1); eng.set(L2_ACC_B, 0)
    eng.copy(L2_SAVE_R_A, L2_RES_A); eng.copy(L2_SAVE_R_B, L2_RES_B)
    eng.copy(L2_SAVE_F_A, L2_ACC_A); eng.copy(L2_SAVE_F_B, L2_ACC_B)
    eng.set(L2_BASE_A, 2); eng.set(L2_BASE_B, 1)
    eng.copy(L2_PREV_A, L2_RES_A); eng.copy(L2_PREV_B, L2_RES_B)

    B = min(max(int(math.sqrt(p)), 1), max(total, 1))
    auto = int((1000.0 * 600.0) / B)
    if auto == 0:
        auto = (total // B) // max(int(math.sqrt(B)), 1)
    checkpasslevel = checklevel if checklevel > 0 else max(auto, 1)
    itersave, jsave = 0, total - 1
    checkpass = checks = errors = done = 0
    errordone = False
    it, j = 0, total - 1
    while it < total:
        if max_iters is not None and done >= max_iters:
            break
        if erroriter > 0 and it + 1 == erroriter and not errordone:
            errordone = True
            eng.sub(L2_RES_A, 2)
            log("Injected error at iteration %d" % (it + 1))
        if it + 1 == total:
            eng.copy(L2_PREV_A, L2_RES_A); eng.copy(L2_PREV_B, L2_RES_B)
        pair_square(L2_RES_A, L2_RES_B)
        done += 1
        if (j != 0 and j % B == 0) or it == total - 1:
            checkpass += 1
            eng.copy(L2_CHK_A, L2_ACC_A); eng.copy(L2_CHK_B, L2_ACC_B)
            pair_mul_by(L2_ACC_A, L2_ACC_B, L2_RES_A, L2_RES_B)
            if checkpass >= checkpasslevel or it == total - 1:
                checks += 1
                modB = B if total % B == 0 else total % B
                for _ in range(B - modB - 1 if B > modB else 0):
                    pair_square(L2_CHK_A, L2_CHK_B)
                if modB != B:
                    pair_square(L2_CHK_A, L2_CHK_B)
                pair_mul_by(L2_CHK_A, L2_CHK_B, L2_BASE_A, L2_BASE_B)
                for _ in range(modB):
                    pair_square(L2_CHK_A, L2_CHK_B)
                ok = eng.get_int(L2_CHK_A) == eng.get_int(L2_ACC_A) and eng.get_int(L2_CHK_B) == eng.get_int(L2_ACC_B)
                if not ok:
                    log("[Gerbicz-Li] Check FAILED at iter=%d block=[%d..%d]" % (it + 1, itersave + 1, it + 1))
                    errors += 1
                    eng.copy(L2_RES_A, L2_SAVE_R_A); eng.copy(L2_RES_B, L2_SAVE_R_B)
                    eng.copy(L2_ACC_A, L2_SAVE_F_A); eng.copy(L2_ACC_B, L2_SAVE_F_B)
                    checkpass = 0
                    if itersave == 0:
                        it, j = 0, jsave
                    else:
                        it, j = itersave + 1, jsave - 1
                    log("[Gerbicz-Li] Restore iter=%d" % it)
                    continue
                log("[Gerbicz-Li] Check OK at iter=%d block=[%d..%d]" % (it + 1, itersave + 1, it + 1))
                eng.copy(L2_SAVE_R_A, L2_RES_A); eng.copy(L2_SAVE_R_B, L2_RES_B)
                eng.copy(L2_SAVE_F_A, L2_ACC_A); eng.copy(L2_SAVE_F_B, L2_ACC_B)
                itersave, jsave = it, j
                checkpass = 0
        it += 1
        j -= 1

    Mp = (1 << p) - 1
    complete = it >= total
    is_prime = complete and eng.get_int(L2_RES_A) == Mp - 1 and eng.get_int(L2_RES_B) == 0
    eng.add(L2_PREV_A, L2_PREV_A)          # S_(p-2) = 2 A_(p-2)
    words = pack_words(eng.digits(L2_PREV_A), p)
    return {"exponent": p, "mode": "llsafe2", "is_prime": bool(is_prime), "res64": format_res64(words),
            "res2048": format_res2048(words), "iterations": it, "gerbicz_checks": checks, "gerbicz_errors": errors,
            "complete": complete}


PROGRAM_VERSION = "mi355-marin-hip 0.3"   # MI355_ENGINE_VERSION of include/mi355_engine.h


def result_json(r, fft_length, program_version=PROGRAM_VERSION, port=8, user="", computer="", aid="", timestamp=""):
    """Result line in the reference's PrimeNet-style JSON (src/io/JsonBuilder.cpp:322-472): same keys, same
    order for the PRP / LL work types ("status" P/C, "worktype" PRP-3 / LL, res64, res2048 + residue-type 1
    for PRP, errors.gerbicz, shift-count 0, fft-length, program)."""
    import json
    prp_mode = r["mode"] == "prp"
    out = [("status", "P" if r["is_prime"] else "C"), ("exponent", r["exponent"]), ("worktype", "PRP-3" if prp_mode else "LL"),
           ("res64", r["res64"])]
    if prp_mode:
        out += [("res2048", r["res2048"]), ("residue-type", 1)]
    out += [("errors", {"gerbicz": r["gerbicz_errors"]}), ("shift-count", 0), ("fft-length", int(fft_length)),
            ("program", {"name": "prmers", "version": program_version, "port": port})]
    for k, v in (("user", user), ("computer", computer), ("aid", aid), ("timestamp", timestamp)):
        if v:
            out.append((k, v))
    return json.dumps(dict(out), separators=(",", ":"))
