#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory into a small JSON (copied to profiles/).

HBM traffic per launch from FETCH_SIZE / WRITE_SIZE as MI355X_MICROARCH.md prescribes: the counters
are in KiB-like units of 1024 bytes... (rocprofv3 reports them in kilobytes), and on gfx950 FETCH_SIZE
reports one half of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is exact."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def avg_counter(path_glob):
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(path_glob):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            out[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in out.items()}


def main(d):
    res = {"dir": os.path.basename(d.rstrip("/"))}
    stats = glob.glob(os.path.join(d, "trace", "*", "*kernel_stats.csv"))
    if stats:
        res["kernel_stats"] = [{"name": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                "pct": float(r["Percentage"])} for r in csv.DictReader(open(stats[0])) if "mi355" in r["Name"]]
    fetch = avg_counter(os.path.join(d, "pmc_fetch", "*", "*counter_collection.csv"))
    write = avg_counter(os.path.join(d, "pmc_write", "*", "*counter_collection.csv"))
    traffic = {}
    for k in fetch:
        if "mi355" not in k:
            continue
        f_kb = fetch[k].get("FETCH_SIZE", 0.0)
        w_kb = write.get(k, {}).get("WRITE_SIZE", 0.0)
        traffic[k] = {"FETCH_SIZE_raw_KB": f_kb, "WRITE_SIZE_raw_KB": w_kb,
                      "read_bytes_corrected_x2": 2 * f_kb * 1024, "write_bytes": w_kb * 1024,
                      "hbm_bytes_per_launch": 2 * f_kb * 1024 + w_kb * 1024}
    res["traffic"] = traffic
    sq = avg_counter(os.path.join(d, "pmc_sq", "*", "*counter_collection.csv"))
    res["sq"] = {k: v for k, v in sq.items() if "mi355" in k}
    sq2 = avg_counter(os.path.join(d, "pmc_sq2", "*", "*counter_collection.csv"))
    for k, v in sq2.items():
        if "mi355" in k:
            res["sq"].setdefault(k, {}).update(v)
    for name in ("bench_trace.log",):
        p = os.path.join(d, name)
        if os.path.exists(p):
            for line in open(p):
                if line.startswith("{"):
                    res["bench"] = json.loads(line)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
