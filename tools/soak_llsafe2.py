import sys, time
sys.path.insert(0, '.')
from prmers_amd import Engine, prp
p = int(sys.argv[1]); iters = int(sys.argv[2])
t = time.time()
with Engine(p, prp.LLSAFE2_REGISTERS) as e:
    r = prp.run_ll_safe2(e, p, checklevel=1, max_iters=iters, log=lambda m: print(m, flush=True))
print(p, {k: r[k] for k in ("iterations", "gerbicz_checks", "gerbicz_errors", "res64")}, "%.1f s" % (time.time() - t), flush=True)
assert r["gerbicz_errors"] == 0 and r["gerbicz_checks"] >= 1
