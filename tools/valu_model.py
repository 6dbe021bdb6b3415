#!/usr/bin/env python3
"""VALU issue model of the three C3 kernels -> profiles/valu_model_latest.json (read by bench.py: roofline.valu).

Compiles prmers_amd/csrc/kernels_v2.hip to gfx950 ISA (device only, no GPU needed), walks every kernel once per wave index
with tools/isa_dynamic_cost.py (measured issue cost per opcode: profiles/r02_microbench_isa2.txt) and records VALU
instructions and issue cycles per wave.  predicted time of a launch = waves per SIMD x cycles per wave / clock, with
tiles x 8 waves spread over the chip's 1024 SIMDs.

usage: tools/valu_model.py [--out profiles/valu_model_latest.json]
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V2_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"]   # the flags prmers_amd/csrc/Makefile builds kernels_v2.hip with (V2FLAGS)
KERNELS = {   # slot of Engine::kernel_name -> (mangled-name substring, what it is)
    "k_front": ("7k1_colsILi2E", "v2::k1_cols<2>"),
    "k_middle": ("11k2_rows4096ILi0ELi1E", "v2::k2_rows4096<0,1>"),
    "k_back": ("7k3_colsILi2E", "v2::k3_cols<2>"),
}


def main():
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "profiles", "valu_model_latest.json")
    src = os.path.join(ROOT, "prmers_amd", "csrc", "kernels_v2.hip")
    res = {"plan": "marin-hip:n=8388608:m1=1024:m2=4096:c=4", "clock_ghz": 2.4, "simds": 1024, "waves_per_launch": 8192,
           "source": "tools/valu_model.py: ISA walk of prmers_amd/csrc/kernels_v2.hip (hipcc -O3 -S --cuda-device-only, the Makefile's V2FLAGS), issue costs of profiles/r02_microbench_isa2.txt",
           "kernels": {}}
    with tempfile.TemporaryDirectory() as td:
        s = os.path.join(td, "kernels_v2.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-Wno-pass-failed", *V2_FLAGS,
                               "-o", s, src], stderr=subprocess.DEVNULL)
        for slot, (sub, pretty) in KERNELS.items():
            j = os.path.join(td, slot + ".json")
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "isa_dynamic_cost.py"), s, "--kernel", sub, "--json", j],
                                  stdout=subprocess.DEVNULL)
            d = json.load(open(j))
            nops = 0
            res["kernels"][slot] = {"kernel": pretty, "valu_insts_per_wave": round(d["valu_per_wave"], 1),
                                    "issue_cycles_per_wave": round(d["issue_cycles_per_wave"], 1),
                                    "predicted_us": round(8 * d["issue_cycles_per_wave"] / 2.4e3, 2)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
