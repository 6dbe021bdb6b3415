"""The environment switches of the engine (README.md, "Switches": MI355_KERNELS, MI355_TUNE bits, MI355_BOOST, MI355_HOST_CARRY; MI355_THREADS
is read once per process and is not varied here) select kernels, tile orders and thread mappings, never results: every setting -- alone and in the combinations
that can meet in one engine -- gives the oracle's digit vectors on shapes of every kernel set ("combinations untested").
The switches are read when an engine is created.  Needs a real MI355X:  python -m pytest tests -m gpu"""
import itertools

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

# exponent, plan: radix-8 columns + rows of 4096 (R = 2: the one shape whose front order bit 5 changes), rows of 2048 (two to a tile /
# planes: bits 6, 13, 14), radix-4 columns and rows (bits 7-11), radix-5 columns of 1280 x 4 and 2560 x 2 (bit 5), generic small tiles
# (bits 3, 4), runs of two digits (C = 1: the carry-fix path)
SHAPES = [(300007, "m2=8,c=4"), (300007, "m2=4096"), (1200007, "m2=2048,c=4"), (600011, "m2=2048"), (300007, "m2=1024"), (132049, "m2=16,c=4"),
          (800283, "m2=16,c=4"), (800283, "m2=8,c=2"), (9941, "m2=16,c=4"), (9941, "m2=128,c=2"), (521, "m2=8,c=2"), (127, "m2=2,c=1")]

KERNELS = [None, "generic", "v2rows", "v2cols"]
TUNE_SINGLE = [0, 1, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 8192, 16384]
# bits that can act on one engine together: tile orders x boost x thread mappings x forced forms
TUNE_COMBOS = [1 | 4 | 32, 1 | 32 | 64, 8 | 16, 4 | 512 | 2048, 256 | 1024, 1 | 4 | 8192, 32 | 16384, 1 | 4 | 8 | 16 | 32 | 64 | 128,
               1 | 4 | 32 | 512 | 2048 | 16384]


def Engine(*a, **k):
    from prmers_amd import Engine as E
    return E(*a, **k)


def reference_run(p):
    """the oracle's digit vectors after a short chain that exercises every sweep variant: squarings with and without a factor, the LL
    step folded into the next load, a multiplicand image and a mul"""
    o = orc.Oracle(p, 3)
    rng = np.random.default_rng(p)
    x0 = int.from_bytes(rng.bytes((p + 7) // 8), "little") % ((1 << p) - 1)
    o.set_value(0, x0)
    out = []
    for a in (1, 3):
        o.square_mul(0, a)
    out.append(o.digits(0).copy())
    o.copy(1, 0); o.set_multiplicand(2, 1)
    o.sub(0, 2); o.square_mul(0); o.mul(0, 2)
    out.append(o.digits(0).copy())
    return x0, out, o.res64(0)


def engine_run(p, plan, x0):
    with Engine(p, 4, plan=plan) as e:
        e.set_int(0, x0)
        out = []
        for a in (1, 3):
            e.square_mul(0, a)
        out.append(e.digits(0).copy())
        e.copy(1, 0); e.set_multiplicand(2, 1)
        e.sub(0, 2); e.square_mul(0); e.mul(0, 2)
        out.append(e.digits(0).copy())
        return out, e.res64(0), e.describe()


@pytest.fixture(scope="module")
def refs():
    return {p: reference_run(p) for p in sorted({p for p, _ in SHAPES})}


def check(refs, p, plan, label):
    x0, want, r64 = refs[p]
    got, g64, desc = engine_run(p, plan, x0)
    for i, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), (p, plan, label, desc, i)
    assert g64 == r64, (p, plan, label)


@pytest.mark.parametrize("p,plan", SHAPES)
def test_every_switch_alone_gives_the_same_digits(p, plan, refs, monkeypatch):
    for ks in KERNELS:
        for tune in TUNE_SINGLE:
            if ks is None: monkeypatch.delenv("MI355_KERNELS", raising=False)
            else: monkeypatch.setenv("MI355_KERNELS", ks)
            monkeypatch.setenv("MI355_TUNE", str(tune))
            check(refs, p, plan, "kernels=%s tune=%d" % (ks, tune))


@pytest.mark.parametrize("p,plan", SHAPES)
def test_switch_combinations_give_the_same_digits(p, plan, refs, monkeypatch):
    for ks, tune, boost, host in itertools.product(KERNELS[:2] + KERNELS[3:], TUNE_COMBOS, ("25", "100"), (None, "1")):
        if ks is None: monkeypatch.delenv("MI355_KERNELS", raising=False)
        else: monkeypatch.setenv("MI355_KERNELS", ks)
        monkeypatch.setenv("MI355_TUNE", str(tune))
        monkeypatch.setenv("MI355_BOOST", boost)
        if host is None: monkeypatch.delenv("MI355_HOST_CARRY", raising=False)
        else: monkeypatch.setenv("MI355_HOST_CARRY", host)
        check(refs, p, plan, "kernels=%s tune=%d boost=%s host_carry=%s" % (ks, tune, boost, host))
