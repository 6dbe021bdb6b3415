#!/usr/bin/env python3
"""Checks of the experiments that are NOT in the product library and only built into libmi355_engine_exp.so (`make -C prmers_amd/csrc exp`):
the one-cooperative-launch squaring kernel (kernels.hip k_coop: measured slower than three launches per squaring, DESIGN history 5.2c) and the
back + front sweep in one launch for runs of squarings on the small shapes (kernels_v3.hip k31_cols256_planes: -2 % at C2 for inter-group
waits, profiles/r04_ab_chain_backfront.txt), and the same fusion on the radix-8 column shapes (kernels_v2.hip k31_cols, MI355_CHAIN=1: +1.4 % at
C3, slower everywhere).

    MI355_ENGINE_LIB=prmers_amd/libmi355_engine_exp.so MI355_COOP=1 python tools/exp_coop_check.py          # on an MI355X box

Digits against the oracle and against the three-launch chain of the same library (MI355_COOP unset), a run of squarings per launch, and
the barrier time-out path (MI355_COOP_FAULT=1)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from prmers_amd import Engine  # noqa: E402
from prmers_amd.engine import EngineError  # noqa: E402

CASES = [(127, "m2=2,c=2"), (521, "m2=4,c=2"), (1801, "m2=8,c=4"), (9941, "m2=64,c=8"), (44497, None), (400063, "m2=64,c=4"), (2976221, None),
         (9815459, None)]


def main():
    if "exp" not in os.environ.get("MI355_ENGINE_LIB", ""):
        raise SystemExit("set MI355_ENGINE_LIB to libmi355_engine_exp.so (make -C prmers_amd/csrc exp)")
    os.environ["MI355_COOP"] = "1"
    for p, plan in CASES:
        rng = np.random.default_rng(p)
        x0 = int.from_bytes(rng.bytes((p + 7) // 8), "little") % ((1 << p) - 1)
        o = orc.Oracle(p, 2); o.set_value(0, x0)
        with Engine(p, 2, plan=plan) as e:
            if ":coop=" not in e.describe():   # (shapes the register-resident kernel sets serve take no cooperative launch)
                assert p > 4000000, e.describe()
                print("skipped", p, e.describe()); continue
            e.set_int(0, x0)
            for it in range(6):
                e.square_mul(0, 3 if it == 4 else 1); o.square_mul(0, 3 if it == 4 else 1)
            assert np.array_equal(e.digits(0), o.digits(0)), (p, plan)
            e.square_mul_n(0, 17); [o.square_mul(0) for _ in range(17)]
            e.square_mul_n(0, 9, 1, 2)
            for _ in range(9): o.square_mul(0); o.sub(0, 2)
            assert np.array_equal(e.digits(0), o.digits(0)), (p, plan)
        print("ok", p, plan)
    # the other experiment of the library: back + front in one launch for runs of squarings on the radix-4 column shapes
    # (kernels_v3.hip k31_cols256_planes, profiles/r04_ab_chain_backfront.txt): square_mul_n against the oracle
    for p, plan in [(86243, "m2=8,c=4"), (132049, "m2=16,c=4"), (756839, "m2=64,c=4"), (9815459, None)]:
        rng = np.random.default_rng(p)
        x0 = int.from_bytes(rng.bytes((p + 7) // 8), "little") % ((1 << p) - 1)
        o = orc.Oracle(p, 2); o.set_value(0, x0)
        with Engine(p, 2, plan=plan) as e:
            e.set_int(0, x0)
            e.square_mul_n(0, 17); [o.square_mul(0) for _ in range(17)]
            assert np.array_equal(e.digits(0), o.digits(0)), (p, plan)
            e.square_mul_n(0, 9, 3, 2)
            for _ in range(9): o.square_mul(0, 3); o.sub(0, 2)
            assert np.array_equal(e.digits(0), o.digits(0)), (p, plan)
        print("ok chained", p, plan)
    # the third experiment: back + front in one launch on the radix-8 column shapes (kernels_v2.hip k31_cols, MI355_CHAIN=1; round 4: +1.4 % at C3,
    # profiles/r04_ab_chain_backfront.txt): n = 2^17 on the three column shapes with 16 tiles each, several tiles per XCD, then the headline size
    os.environ["MI355_CHAIN"] = "1"
    for p, plan in [(2976221, "m2=128,c=8"), (2976221, "m2=64,c=4"), (2976221, "m2=32,c=2"), (9815459, "m2=256,c=4"), (19000013, "m2=1024,c=8")]:
        rng = np.random.default_rng(p)
        x0 = int.from_bytes(rng.bytes((p + 7) // 8), "little") % ((1 << p) - 1)
        o = orc.Oracle(p, 2); o.set_value(0, x0)
        with Engine(p, 2, plan=plan) as e:
            e.set_int(0, x0)
            e.square_mul_n(0, 7); [o.square_mul(0) for _ in range(7)]
            assert np.array_equal(e.digits(0), o.digits(0)), (p, plan)
            e.square_mul_n(0, 5, 1, 2); e.square_mul_n(0, 2, 1, 2)
            for _ in range(7): o.square_mul(0); o.sub(0, 2)
            assert np.array_equal(e.digits(0), o.digits(0)), (p, plan)
            e.square_mul_n(0, 300); [o.square_mul(0) for _ in range(300)]
            assert np.array_equal(e.digits(0), o.digits(0)), (p, plan)
        print("ok chained (radix-8 columns)", p, plan)
    p = 136279841
    o = orc.Oracle(p, 1)
    w = o.widths().astype(np.uint64)
    d = ((np.uint64(1) << w) - np.uint64(1)) | (w << np.uint64(32))
    d[0] -= np.uint64(3)
    with Engine(p, 3) as e:
        e.set_digits(0, d); e.set_digits(1, d); o.set_digits(0, d)
        e.square_mul_n(0, 4, 1, 2)
        for _ in range(4): e.square_mul(1); e.sub(1, 2); o.square_mul(0); o.sub(0, 2)
        assert np.array_equal(e.digits(0), e.digits(1)) and np.array_equal(e.digits(0), o.digits(0))
        e.square_mul_n(0, 2000, 1, 2)
        for _ in range(2000): e.square_mul(1); e.sub(1, 2)
        assert e.is_equal(0, 1)
    print("ok chained (radix-8 columns)", p)
    del os.environ["MI355_CHAIN"]
    os.environ["MI355_COOP_FAULT"] = "1"
    with Engine(9941, 2, plan="m2=16,c=4") as e:
        e.set(0, 3); e.square_mul(0)
        try:
            e.digits(0)
            raise SystemExit("the barrier time-out was not reported")
        except EngineError as exc:
            assert "grid barrier timed out" in str(exc)
    print("ok barrier time-out reported")


if __name__ == "__main__":
    main()
