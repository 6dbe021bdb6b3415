#!/usr/bin/env python3
"""Generate tests/golden/huge_p_pins.json: pins for the reference's transform sizes above 5*2^23
(include/marin/engine_gpu.h:1598,1622-1623: 2^26, 5*2^24, 5*2^25), same scheme and same libgmp arithmetic as
make_big_p_pins.py (x_0 = 3, x_{i+1} = x_i^2 mod 2^p-1; res64, low 2048 bits, SHA-256 of the canonical words).
Run: python tests/golden/make_huge_p_pins.py   (about ten minutes: 1.8-gigabit squarings)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_big_p_pins import pins   # noqa: E402

if __name__ == "__main__":
    doc = {"_source": "libgmp via ctypes (tests/golden/make_huge_p_pins.py); x_0 = 3, x_{i+1} = x_i^2 mod 2^p-1; "
                      "words = canonical little-endian 32-bit words, low2048 = hex of the low 2048 bits (most significant first)",
           "pins": {}}
    for p, start, count in ((800000011, 31, 2), (1300000003, 32, 2), (1800000011, 32, 2)):
        doc["pins"][str(p)] = pins(p, start, count - 1)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "huge_p_pins.json")
    json.dump(doc, open(path, "w"), indent=1)
    print("wrote", path)
