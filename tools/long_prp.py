"""A complete PRP of 2^p-1 in slices that fit one GPU-box call: each call runs for about --seconds, stops right
after a passed Gerbicz-Li check and saves (residue, Gerbicz accumulator, iteration) to --state; the next call
continues from there.  Usage: python tools/long_prp.py <p> --state state/long_prp.npz --out gpurun_out/long_prp.npz"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, '.')
from prmers_amd import Engine, prp, resolve_plan

ap = argparse.ArgumentParser()
ap.add_argument("p", type=int)
ap.add_argument("--state", default="state/long_prp.npz")
ap.add_argument("--out", default="gpurun_out/long_prp.npz")
ap.add_argument("--seconds", type=float, default=900.0)
a = ap.parse_args()
p = a.p
print(p, resolve_plan(p), flush=True)
t0 = time.time()
last = [t0]
def log(m):
    if time.time() - last[0] > 60 or "FAILED" in m or "Restore" in m:
        print(m, "%.0f s" % (time.time() - t0), flush=True); last[0] = time.time()
with Engine(p, prp.REGISTERS) as e:
    resume = None
    if os.path.exists(a.state):
        z = np.load(a.state)
        assert int(z["p"]) == p
        e.set_digits(prp.R0, z["r0"].astype(np.uint64)); e.set_digits(prp.R1, z["r1"].astype(np.uint64))
        resume = {"it": int(z["it"]), "j": int(z["j"])}
        print("resuming after iteration", resume["it"] + 1, "errors so far", int(z["errors"]), flush=True)
    r = prp.run_prp_or_ll(e, p, "prp", log=log, resume=resume, stop_after_s=a.seconds)
    prev_err = int(np.load(a.state)["errors"]) if os.path.exists(a.state) else 0
    if r["state"] is not None:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        np.savez(a.out, p=p, it=r["state"]["it"], j=r["state"]["j"], errors=prev_err + r["gerbicz_errors"],
                 r0=(e.digits(prp.R0) & np.uint64(0xFFFFFFFF)).astype(np.uint32), r1=(e.digits(prp.R1) & np.uint64(0xFFFFFFFF)).astype(np.uint32))
        print("SLICE DONE: stopped after iteration %d of %d (%.2f %%), checks this slice %d, errors %d, res64 of the residue %016X, %.0f s"
              % (r["state"]["it"] + 1, p, 100.0 * (r["state"]["it"] + 1) / p, r["gerbicz_checks"], r["gerbicz_errors"], e.res64(prp.R0), time.time() - t0), flush=True)
    else:
        print("COMPLETE:", {k: r[k] for k in ("exponent", "is_prime", "res64", "iterations", "gerbicz_checks", "gerbicz_errors")},
              "errors in earlier slices", prev_err, "%.0f s" % (time.time() - t0), flush=True)
