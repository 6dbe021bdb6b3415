import sys, time
sys.path.insert(0, '.')
from prmers_amd import Engine, prp
p = int(sys.argv[1]); iters = int(sys.argv[2]); lvl = int(sys.argv[3])
msgs = []
t = time.time()
with Engine(p, prp.REGISTERS) as e:
    r = prp.run_prp_or_ll(e, p, "prp", checklevel=lvl, max_iters=iters, log=lambda m: (msgs.append(m), print(m, flush=True)))
print(p, {k: r[k] for k in ("iterations", "gerbicz_checks", "gerbicz_errors", "res64")}, "%.1f s" % (time.time() - t), flush=True)
assert r["gerbicz_errors"] == 0 and r["gerbicz_checks"] >= 1
