#!/usr/bin/env python3
"""Dynamic VALU issue cost of one kernel from its gfx950 ISA text, per wave index.

The squaring kernels are straight-line code except for (a) one scalar switch per shift seam on the wave index
(eight constant-shift copies, each wave runs its own), (b) the rare-correction slow paths of gf.hpp behind
`s_cbranch_scc0 .Lgf_fast*` (taken with probability ~2^-32 per lane) and (c) the structurizer's exec tests, which a
full wave never takes.  This walks the kernel once per wave index with those three rules, sums the measured issue cost
of every instruction on the path (tools/check_isa_hazards.py: issue_cost, profiles/r02_microbench_isa2.txt) and prints
the opcode histogram averaged over the eight waves, so that

    predicted time = waves_per_simd x cycles_per_wave / clock

can be put next to the rocprof duration of the same kernel (profiles/r02_valu_histogram.md).

usage: isa_dynamic_cost.py file.s --kernel SUBSTR [--waves 8] [--top 30] [--json out.json]
A conditional branch that is none of (a)-(c) is assumed NOT taken (the guarded block is counted) and listed.
"""
import json
import re
import sys
from collections import Counter

sys.path.insert(0, __import__("os").path.dirname(__file__))
from check_isa_hazards import issue_cost  # noqa: E402


def load_kernel(path, substr):
    insns, labels, name = [], {}, None
    grab = False
    for raw in open(path):
        line = raw.split(";")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", line)
        if m:
            lab = m.group(1)
            if not lab.startswith(".L"):
                if grab:
                    break
                if substr in lab:
                    grab, name = True, lab
                continue
            if grab:
                labels[lab] = len(insns)
            continue
        if not grab or line.lstrip().startswith("."):
            continue
        parts = line.strip().split(None, 1)
        mn = parts[0]
        if re.match(r"^[vs]_|^ds_|^global_|^buffer_|^flat_|^scratch_", mn):
            insns.append((mn, parts[1] if len(parts) > 1 else ""))
    if name is None:
        raise SystemExit("no kernel matching %r in %s" % (substr, path))
    return name, insns, labels


def wave_register(insns):
    """the SGPR compared against small constants most often: the scalar switch variable"""
    c = Counter()
    for mn, ops in insns:
        if mn.startswith("s_cmp_"):
            o = [x.strip() for x in ops.split(",")]
            if len(o) == 2 and re.fullmatch(r"s\d+", o[0]) and re.fullmatch(r"-?\d+", o[1]) and 0 <= int(o[1]) <= 8:
                c[o[0]] += 1
    return c.most_common(1)[0][0] if c else None


CMP = {"eq": lambda a, b: a == b, "lg": lambda a, b: a != b, "lt": lambda a, b: a < b, "gt": lambda a, b: a > b,
       "le": lambda a, b: a <= b, "ge": lambda a, b: a >= b}


def walk(insns, labels, wreg, w):
    hist, cost, pc, scc, unknown = Counter(), 0.0, 0, None, []
    steps = 0
    while pc < len(insns):
        steps += 1
        if steps > 10 * len(insns):
            raise SystemExit("walk does not terminate (a loop the rules above do not resolve)")
        mn, ops = insns[pc]
        hist[mn] += 1
        cost += issue_cost(mn, ops)
        if mn.startswith("s_cmp_"):
            o = [x.strip() for x in ops.split(",")]
            m = re.fullmatch(r"s_cmp_(eq|lg|lt|gt|le|ge)_[iu]32", mn)
            if m and len(o) == 2 and o[0] == wreg and re.fullmatch(r"-?\d+", o[1]):
                scc = CMP[m.group(1)](w, int(o[1]))
            else:
                scc = None
        elif mn.startswith("s_") and not mn.startswith(("s_cbranch", "s_branch", "s_nop", "s_waitcnt", "s_barrier", "s_mov", "s_load", "s_setprio",
                                                         "s_endpgm", "s_sleep")):
            scc = None   # any other SALU instruction may write SCC
        if mn == "s_endpgm":
            break
        if mn == "s_branch":
            pc = labels[ops.strip()]
            continue
        if mn.startswith("s_cbranch_"):
            kind, tgt = mn[len("s_cbranch_"):], ops.strip()
            if kind == "execz":
                take = False
            elif kind == "execnz":
                take = True
            elif tgt.startswith(".Lgf_fast"):
                take = True
            elif kind in ("scc0", "scc1") and scc is not None:
                take = scc == (kind == "scc1")
            else:
                take = False
                unknown.append((pc, mn, tgt))
            if take:
                pc = labels[tgt]
                continue
        pc += 1
    return hist, cost, unknown


def main():
    a = sys.argv[1:]
    path = a[0]
    kern = a[a.index("--kernel") + 1]
    nw = int(a[a.index("--waves") + 1]) if "--waves" in a else 8
    top = int(a[a.index("--top") + 1]) if "--top" in a else 30
    name, insns, labels = load_kernel(path, kern)
    wreg = wave_register(insns)
    tot, costs, unk = Counter(), [], set()
    for w in range(nw):
        h, c, u = walk(insns, labels, wreg, w)
        tot.update(h)
        costs.append(c)
        unk.update(u)
    valu = {k: v / nw for k, v in tot.items() if k.startswith("v_")}
    other = {k: v / nw for k, v in tot.items() if not k.startswith("v_")}
    n_valu = sum(valu.values())
    avg = sum(costs) / nw
    print("kernel %s" % name)
    print("static instructions %d; switch variable %s; per-wave VALU cost (cycles) by wave index: %s" % (len(insns), wreg, " ".join("%.0f" % c for c in costs)))
    print("dynamic per wave (mean over %d wave indices): %.0f VALU instructions, %.0f issue cycles (%.2f cycles / instruction)" % (nw, n_valu, avg, avg / n_valu))
    print("other per wave: " + ", ".join("%s %.0f" % (k, v) for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:12]))
    rows = []
    for k, v in sorted(valu.items(), key=lambda kv: -kv[1] * issue_cost(kv[0], "v0, v0, v0, vcc" if "cndmask" in kv[0] and kv[0].endswith("e32") else "")):
        rows.append((k, v))
    # cost share needs the operands (cndmask forms): recompute per opcode on the path of wave 1
    share = Counter()
    for w in range(nw):
        pc_hist = Counter()
        # second walk collecting cost per opcode
        pc, scc, steps = 0, None, 0
        while pc < len(insns):
            mn, ops = insns[pc]
            share[mn] += issue_cost(mn, ops) / nw
            if mn.startswith("s_cmp_"):
                o = [x.strip() for x in ops.split(",")]
                m = re.fullmatch(r"s_cmp_(eq|lg|lt|gt|le|ge)_[iu]32", mn)
                scc = CMP[m.group(1)](w, int(o[1])) if (m and len(o) == 2 and o[0] == wreg and re.fullmatch(r"-?\d+", o[1])) else None
            if mn == "s_endpgm":
                break
            if mn == "s_branch":
                pc = labels[ops.strip()]
                continue
            if mn.startswith("s_cbranch_"):
                kind, tgt = mn[len("s_cbranch_"):], ops.strip()
                take = (kind == "execnz") or tgt.startswith(".Lgf_fast") or (kind in ("scc0", "scc1") and scc is not None and scc == (kind == "scc1"))
                if kind == "execz":
                    take = False
                if take:
                    pc = labels[tgt]
                    continue
            pc += 1
    print("%-28s %10s %12s %7s" % ("opcode", "per wave", "issue cycles", "share"))
    for k, c in share.most_common(top):
        if c <= 0:
            continue
        print("%-28s %10.1f %12.0f %6.1f%%" % (k, valu.get(k, 0), c, 100 * c / avg))
    if unk:
        print("conditional branches assumed not taken: " + ", ".join("%s %s @%d" % (m, t, p) for p, m, t in sorted(unk)))
    if "--json" in a:
        json.dump({"kernel": name, "valu_per_wave": n_valu, "issue_cycles_per_wave": avg, "by_wave": costs,
                   "cycles_by_opcode": dict(share.most_common()), "count_by_opcode": valu}, open(a[a.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
