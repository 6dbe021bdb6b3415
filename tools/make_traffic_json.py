#!/usr/bin/env python3
"""profiles/<name>_profile_summary.json (tools/summarize_profile.py) -> the per-launch HBM traffic file bench.py reads:
   tools/make_traffic_json.py profiles/r03_final_profile_summary.json profiles/traffic_latest.json          (Goldilocks, C3)
   tools/make_traffic_json.py profiles/r03_crt_profile_summary.json profiles/traffic_crt_latest.json crt    (second field family)
Goldilocks: bytes per launch of the three sweeps by engine slot name.  crt: bytes per kernel and their sum per squaring."""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
crt = len(sys.argv) > 3 and sys.argv[3] == "crt"
d = json.load(open(src))
tr = d["traffic"]
if not crt:
    slot = {"k_front": "k1_cols", "k_middle": "k2_rows4096", "k_back": "k3_cols"}
    out = {}
    for s, sub in slot.items():
        hits = [v["hbm_bytes_per_launch"] for k, v in tr.items() if sub in k]
        if hits:
            out[s] = max(hits)
    out["unit"] = "HBM bytes per launch (FETCH_SIZE/WRITE_SIZE PMC passes, corrected as noted in the source file)"
    out["plan"] = d["bench"]["config"]["plan"]
else:
    ks = {k.split("::")[-1]: v["hbm_bytes_per_launch"] for k, v in tr.items() if "crt" in k}
    calls = {r["name"].split("::")[-1]: r["calls"] for r in d.get("kernel_stats", []) if "crt" in r["name"]}
    base = max(calls.values()) if calls else 1
    out = {"kernels": ks, "launches_per_squaring": {k: round(c / base, 3) for k, c in calls.items()},
           "per_squaring": sum(v * round(calls.get(k, base) / base) for k, v in ks.items()),
           "unit": "HBM bytes (FETCH_SIZE/WRITE_SIZE PMC passes, corrected as noted in the source file)",
           "plan": d["bench"]["config"]["plan"]}
out["source"] = "%s (tools/profile.sh)" % src
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
