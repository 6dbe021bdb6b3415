"""Complete PRP of 2^p-1 on the engine with the Gerbicz-Li check on (one-off validation runs on the GPU box).
Usage: python tools/full_prp.py <p>"""
import sys, time
sys.path.insert(0, '.')
from prmers_amd import Engine, prp, resolve_plan
p = int(sys.argv[1])
print(p, resolve_plan(p), flush=True)
t = time.time()
last = [t]
def log(m):
    if time.time() - last[0] > 30 or "FAILED" in m or "Restore" in m:
        print(m, "%.0f s" % (time.time() - t), flush=True); last[0] = time.time()
with Engine(p, prp.REGISTERS) as e:
    r = prp.run_prp_or_ll(e, p, "prp", log=log)
print({k: r[k] for k in ("exponent", "is_prime", "res64", "iterations", "gerbicz_checks", "gerbicz_errors")}, "%.1f s" % (time.time() - t), flush=True)
