#!/usr/bin/env python3
"""Register / scratch / occupancy figures of the kernels of one HIP source (no GPU needed): hipcc -Rpass-analysis=kernel-resource-usage.
usage: tools/kernel_resources.py prmers_amd/csrc/kernels_v2.hip [substring ...] [-- extra hipcc flags]"""
import re
import subprocess
import sys


def main():
    a = sys.argv[1:]
    extra = []
    if "--" in a:
        extra = a[a.index("--") + 1:]
        a = a[:a.index("--")]
    src, subs = a[0], a[1:]
    flags = ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"] if "kernels_v2.hip" in src else []
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", "--cuda-device-only", "-Wno-pass-failed",
                          "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", src, *flags, *extra], capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    for name, r in rows.items():
        if subs and not any(s in name for s in subs):
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        print("%-60s VGPRs %3d AGPRs %3d SGPRs %3d scratch %4d spills v/s %d/%d occupancy %d" % (
            dem[-60:], r.get("VGPRs", -1), r.get("AGPRs", 0), r.get("TotalSGPRs", r.get("SGPRs", -1)), r.get("ScratchSize", 0), r.get("VGPRs Spill", 0), r.get("SGPRs Spill", 0),
            r.get("Occupancy", -1)))


if __name__ == "__main__":
    main()
