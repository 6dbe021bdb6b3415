#!/usr/bin/env python3
"""Timeline / grid-scaling probe of the three C3 sweeps (diagnostics; needs libmi355_engine_probe.so:
make -C prmers_amd/csrc -j8 OBJDIR=obj_probe OUT=../libmi355_engine_probe.so EXTRA=-DMI355_PROBE).

For each sweep (front, rows, back): average launch time over 1, 2 and 4 rounds of its grid (fixed ramp/tail against
per-round cost), with one work-group per CU (padding LDS), with and without the last-half-round priority boost, and one
launch with the per-work-group timeline on: realtime (100 MHz) and shader-clock stamps at entry / exit, HW_ID and XCC_ID
of the wave that recorded them.  Output: a text summary on stdout and gpurun_out/probe_<tag>.npz.
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355_ENGINE_LIB", os.path.join(ROOT, "prmers_amd", "libmi355_engine_probe.so"))
import numpy as np  # noqa: E402

from prmers_amd import Engine  # noqa: E402
from prmers_amd.engine import load_library  # noqa: E402


def probe(e, L, kind, mult, extra_lds, boost, iters, timeline=True, base=1024):
    grid = base * mult
    tl = np.zeros(grid * 8, dtype=np.uint64) if timeline else None
    ms = C.c_double(0)
    rc = L.mi355_probe(C.c_void_p(e.h), kind, mult, extra_lds, boost, iters, C.byref(ms),
                       tl.ctypes.data_as(C.c_void_p) if timeline else None, grid * 8 if timeline else 0)
    if not rc:
        raise RuntimeError(L.mi355_engine_last_error().decode())
    return ms.value, (tl.reshape(grid, 8) if timeline else None)


def analyse(tl):
    """per-launch figures from a timeline: span, shader clock, per-CU busy intervals"""
    r0, r1, c0, c1, hw, xcc = (tl[:, i].astype(np.int64) for i in range(6))
    t0 = r0.min()
    span_us = (r1.max() - t0) / 100.0
    dur_us = (r1 - r0) / 100.0
    with np.errstate(divide="ignore", invalid="ignore"):
        ghz = np.where(r1 > r0, (c1 - c0) / ((r1 - r0) * 10.0), 0)   # shader cycles per ns
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    x = xcc & 0xf
    cuid = ((x * 8 + se) * 2 + sh) * 16 + cu
    start_us = (r0 - t0) / 100.0
    end_us = (r1 - t0) / 100.0
    # per-CU: time with 2, 1, 0 groups resident between launch start and launch end
    occ = {0: 0.0, 1: 0.0, 2: 0.0, 3: 0.0}
    ncu = 0
    last_end = []
    for c in np.unique(cuid):
        m = cuid == c
        ev = sorted([(s, 1) for s in start_us[m]] + [(t, -1) for t in end_us[m]])
        cur, prev = 0, 0.0
        for tt, d in ev:
            occ[min(cur, 3)] += tt - prev
            prev = tt
            cur += d
        occ[0] += span_us - prev
        ncu += 1
        last_end.append(end_us[m].max())
    last_end = np.array(last_end)
    return {"span_us": round(span_us, 2), "groups": int(len(r0)), "cus_seen": ncu,
            "group_dur_us": {"mean": round(float(dur_us.mean()), 2), "min": round(float(dur_us.min()), 2), "max": round(float(dur_us.max()), 2)},
            "shader_ghz": {"mean": round(float(ghz[ghz > 0].mean()), 3), "min": round(float(ghz[ghz > 0].min()), 3), "max": round(float(ghz.max()), 3)},
            "cu_time_frac_with_k_groups": {k: round(v / (span_us * ncu), 3) for k, v in occ.items()},
            "first_start_spread_us": round(float(np.sort(start_us)[min(511, len(start_us) - 1)]), 2),
            "cu_last_end_us": {"min": round(float(last_end.min()), 2), "mean": round(float(last_end.mean()), 2), "max": round(float(last_end.max()), 2)}}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    L = load_library()
    vp = C.c_void_p
    L.mi355_probe.restype = C.c_int
    L.mi355_probe.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_double), vp, C.c_size_t]
    p = int(sys.argv[3]) if len(sys.argv) > 3 else 136279841
    out = {}
    arrays = {}
    with Engine(p, 3) as e:
        sys.path.insert(0, ROOT)
        from bench import seeded_digits
        e.set_digits(0, seeded_digits(p, e.n, 1))
        for _ in range(5):
            e.square_mul(0)
        e.copy(1, 0)
        e.copy(2, 0)
        e.sync()
        # clock warm-up
        t = e.time_square_mul(2, 3000)
        out["squaring_ms_warm"] = t[0] / 3000
        names = {0: "front", 1: "rows", 2: "back"}
        quick = len(sys.argv) > 2 and sys.argv[2] == "quick"
        cfgs = [(1, 0, 50, "x1"), (1, 0, 0, "x1_noboost"), (2, 0, 50, "x2"), (4, 0, 50, "x4"), (4, 0, 0, "x4_noboost"),
                (1, 24 * 1024, 0, "x1_one_group_per_cu"), (2, 24 * 1024, 0, "x2_one_group_per_cu")]
        if quick:
            cfgs = [(1, 0, 50, "x1"), (1, 0, -50, "x1_after_another_kernel"), (2, 0, 50, "x2"), (4, 0, 50, "x4")]
        if len(sys.argv) > 4:   # extra padding LDS values to try (bytes)
            cfgs = [(1, int(x), 50, "x1_lds+%s" % x) for x in sys.argv[4].split(",")] + [(1, int(x), -50, "x1_after_another_kernel_lds+%s" % x) for x in sys.argv[4].split(",")]
        for kind in (1, 0, 2):
            for (mult, lds, boost, label) in cfgs:
                base = (e.n // 2 // 4096) if kind == 1 else 1024        # rows: M1 (rows of 4096), columns: M2 / C tiles
                ms, tl = probe(e, L, kind, mult, lds, boost, 200, base=base)
                key = "%s_%s" % (names[kind], label)
                out[key] = {"avg_us": round(ms * 1e3, 2)}
                out[key].update(analyse(tl))
                arrays[key] = tl
                print(key, json.dumps(out[key]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "probe_%s.npz" % tag), **arrays)
    with open(os.path.join(ROOT, "gpurun_out", "probe_%s.json" % tag), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
