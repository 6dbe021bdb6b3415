"""Complete PRP of a Mersenne exponent on the GF(M61^2) x GF(M31^2) engine with Gerbicz-Li checks (usage: soak_crt.py p odd [checklevel])."""
import sys, time
sys.path.insert(0, '.')
from prmers_amd import CrtEngine, prp
p = int(sys.argv[1]); odd = int(sys.argv[2]); lvl = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t = time.time()
with CrtEngine(p, odd, reg_count=prp.REGISTERS) as e:
    print(e.describe(), flush=True)
    r = prp.run_prp_or_ll(e, p, "prp", checklevel=lvl, log=lambda m: print(m, flush=True) if "Check" in m else None)
print(p, {k: r[k] for k in ("is_prime", "iterations", "gerbicz_checks", "gerbicz_errors", "res64")}, "%.1f s" % (time.time() - t), flush=True)
assert r["gerbicz_errors"] == 0 and r["gerbicz_checks"] >= 1 and r["complete"]
