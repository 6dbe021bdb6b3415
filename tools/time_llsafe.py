"""Per-iteration cost of the two LL-safe loops next to a plain squaring (add / copy / mul on weakly carried digits).
usage (GPU box): python tools/time_llsafe.py [exponent] [iterations]"""
import sys, time
sys.path.insert(0, '.')
from prmers_amd import Engine, prp

p = int(sys.argv[1]) if len(sys.argv) > 1 else 136279841
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2000


def timed(fn):
    fn(200)
    t = time.perf_counter(); fn(iters); return 1e3 * (time.perf_counter() - t) / iters


with Engine(p, prp.LLSAFE2_REGISTERS) as e:
    def plain(k):
        e.set(0, 3)
        for _ in range(k):
            e.square_mul(0)
        e.sync()
    def ll(k):
        e.set(0, 4)
        for _ in range(k):
            e.square_mul(0); e.sub(0, 2)
        e.sync()
    t_plain, t_ll = timed(plain), timed(ll)
    t0 = time.perf_counter(); r1 = prp.run_ll_safe(e, p, max_iters=iters); e.sync(); t_safe1 = 1e3 * (time.perf_counter() - t0) / r1["iterations"]
    t0 = time.perf_counter(); r2 = prp.run_ll_safe2(e, p, max_iters=iters); e.sync(); t_safe2 = 1e3 * (time.perf_counter() - t0) / r2["iterations"]
    print("p=%d n=%d: square_mul %.4f ms, LL step (square, -2) %.4f ms, LL-safe (block recomputation: mul + square per step) %.4f ms, "
          "LL-safe2 (Z[sqrt 3] pair: 2 squares, 1 multiplicand, 1 mul, 1 add, 3 copies per step) %.4f ms" % (p, e.n, t_plain, t_ll, t_safe1, t_safe2))
