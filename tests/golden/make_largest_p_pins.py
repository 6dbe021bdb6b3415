#!/usr/bin/env python3
"""Generate tests/golden/largest_p_pins.json: pins for the reference's largest transform size, n = 5*2^26 (include/marin/engine_gpu.h:1624),
same scheme and same libgmp arithmetic as make_big_p_pins.py (x_0 = 3, x_{i+1} = x_i^2 mod 2^p-1; res64, low 2048 bits, SHA-256 of the
canonical words).  Run: python tests/golden/make_largest_p_pins.py   (4-gigabit squarings: ~15 minutes)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_big_p_pins import pins   # noqa: E402

if __name__ == "__main__":
    doc = {"_source": "libgmp via ctypes (tests/golden/make_largest_p_pins.py); x_0 = 3, x_{i+1} = x_i^2 mod 2^p-1; "
                      "words = canonical little-endian 32-bit words, low2048 = hex of the low 2048 bits (most significant first)",
           "pins": {}}
    for p, start, count in ((4000000007, 33, 2),):
        doc["pins"][str(p)] = pins(p, start, count - 1)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "largest_p_pins.json")
    json.dump(doc, open(path, "w"), indent=1)
    print("wrote", path)
