#!/usr/bin/env python3
"""Static check of gfx950 ISA text for the one software-managed hazard the inline asm in gf.hpp can create:
a VALU instruction that writes an SGPR / VCC (compare, carry-out, v_mad_u64_u32 carry) followed, with fewer than
two wait states in between, by a VALU instruction that reads that register (as a mask or carry-in).  The
compiler inserts the s_nops for its own code but does not look inside asm statements, so the whole file is
scanned.  Also reports, per kernel, an opcode histogram weighted with the measured issue costs
(profiles/r02_microbench_isa2.txt) when --histogram is given.

usage: check_isa_hazards.py file.s [--histogram] [--kernel SUBSTR]
exit status 1 when a hazard is found.
"""
import re
import sys
from collections import Counter, defaultdict

WAIT_STATES = 2
SREG = re.compile(r"(?<![\w.])(vcc|s\[(\d+):(\d+)\]|s(\d+))(?![\w\[])")
CARRY_OUT_OPS = ("v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32",
                 "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")

# measured cycles per wave-instruction per SIMD at 4 waves / SIMD (profiles/r02_microbench_isa2.txt, r01_microbench_isa.txt)
FULL_RATE = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32",
             "v_ashrrev_i32", "v_cndmask_b32"}


def sregs(tok):
    """set of scalar register numbers named by an operand token ('vcc' -> {'vcc'})"""
    out = set()
    for m in SREG.finditer(tok):
        if m.group(1) == "vcc":
            out.add("vcc")
        elif m.group(2) is not None:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add(int(m.group(4)))
    return out


def issue_cost(mn, ops):
    base = mn[:-4] if mn.endswith(("_e32", "_e64")) else mn
    if not mn.startswith("v_"):
        return 0.0
    if base in ("v_mad_u64_u32", "v_mad_i64_i32"):
        return 5.2
    if base.startswith("v_cmp"):
        return 4.9
    if base in FULL_RATE:
        # VOP3 encodings of the same operation issue at the slow rate: explicit _e64, an SGPR / constant in src1, or a non-VCC mask
        if mn.endswith("_e64"):
            return 4.6
        if base == "v_cndmask_b32":
            o = [x.strip() for x in ops.split(",")]
            if len(o) == 4 and (o[3] != "vcc" or not o[2].startswith("v")):
                return 4.6
        return 2.7
    return 4.6


def scan(path, want_hist=False, kernel_filter=None):
    kernels = []
    cur = None
    for raw in open(path):
        line = raw.split(";")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", line)
        if m:
            name = m.group(1)
            if not name.startswith(".L"):
                cur = {"name": name, "insns": []}
                kernels.append(cur)
            elif cur is not None:
                cur["insns"].append(("label", name, ""))
            continue
        if cur is None or line.lstrip().startswith("."):
            continue
        parts = line.strip().split(None, 1)
        mn = parts[0]
        ops = parts[1] if len(parts) > 1 else ""
        if re.match(r"^[vs]_|^ds_|^global_|^buffer_|^flat_|^scratch_", mn):
            cur["insns"].append((mn, ops, raw.rstrip()))
    bad = []
    for k in kernels:
        if kernel_filter and kernel_filter not in k["name"]:
            continue
        # distance (in wait states) since the last VALU write of each scalar register
        since = defaultdict(lambda: 99)
        at_branch = defaultdict(list)
        hist = Counter()
        cost = 0.0
        for idx, (mn, ops, raw) in enumerate(k["insns"]):
            if mn == "label":
                for snap in at_branch.pop(ops, []):
                    for r, d in snap.items():
                        since[r] = min(since[r], d)
                continue
            hist[mn] += 1
            cost += issue_cost(mn, ops)
            toks = [t.strip() for t in ops.split(",")] if ops else []
            is_valu = mn.startswith("v_")
            base = mn[:-4] if mn.endswith(("_e32", "_e64")) else mn
            writes = set()
            reads = set()
            if is_valu:
                if base.startswith("v_cmp"):
                    if base.startswith("v_cmpx"):
                        pass
                    elif toks and SREG.fullmatch(toks[0]):
                        writes |= sregs(toks[0]); src = toks[1:]
                    else:
                        writes.add("vcc"); src = toks
                    for t in src:
                        reads |= sregs(t)
                elif base.startswith(CARRY_OUT_OPS):
                    writes |= sregs(toks[1]) if len(toks) > 1 else set()
                    for t in toks[2:]:
                        reads |= sregs(t)
                elif base in ("v_readlane_b32", "v_readfirstlane_b32"):
                    pass   # different hazard class (the compiler's own)
                else:
                    for t in toks[1:]:
                        reads |= sregs(t)
                for r in reads:
                    if since[r] < WAIT_STATES:
                        bad.append((k["name"], idx, raw.strip(), r, since[r]))
            # advance
            states = 1
            if mn == "s_nop":
                states = int(ops.strip(), 0) + 1
            for r in list(since):
                since[r] += states
            for r in writes:
                since[r] = 0
            if mn.startswith("s_cbranch") or mn == "s_branch":
                at_branch[ops.strip()].append(dict(since))
        if want_hist:
            print(f"== {k['name']}: {sum(hist.values())} instructions, static VALU issue cost {cost:.0f} cycles")
            for mn, c in hist.most_common(40):
                print(f"   {c:6d}  {mn}")
    return bad


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    kf = None
    if "--kernel" in sys.argv:
        kf = sys.argv[sys.argv.index("--kernel") + 1]
        args = [a for a in args if a != kf]
    bad = []
    for path in args:
        bad += scan(path, "--histogram" in sys.argv, kf)
    for name, idx, raw, reg, d in bad:
        print(f"HAZARD in {name} at instruction {idx}: '{raw}' reads {reg} {d} wait state(s) after a VALU wrote it")
    print(f"{len(bad)} hazard(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
