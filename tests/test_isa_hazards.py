"""gfx950 needs two wait states between a VALU instruction that writes an SGPR / VCC and a VALU instruction that reads it
(the compiler pads its own code but does not look inside the asm statements of prmers_amd/csrc/gf.hpp).
tools/check_isa_hazards.py scans the generated ISA of the shipped kernels for that pattern.  No GPU needed: hipcc
cross-compiles to assembly text."""
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
CHECK = os.path.join(ROOT, "tools", "check_isa_hazards.py")


def _isa(src, td, extra=()):
    out = os.path.join(td, os.path.basename(src) + ".s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-x", "hip", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "prmers_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "include"), *extra, src, "-o", out], stderr=subprocess.DEVNULL)
    return out


def test_checker_sees_a_planted_hazard_and_accepts_padded_code(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("k:\n\tv_add_co_u32_e32 v0, vcc, v1, v2\n\tv_addc_co_u32_e32 v3, vcc, v4, v5, vcc\n\ts_endpgm\n")
    ok = tmp_path / "ok.s"
    ok.write_text("k:\n\tv_mad_u64_u32 v[0:1], s[4:5], v2, v3, v[0:1]\n\ts_nop 1\n\tv_cndmask_b32_e64 v6, 0, -1, s[4:5]\n"
                  "\tv_cmp_lt_u64_e64 s[6:7], s[0:1], v[0:1]\n\ts_or_b64 vcc, vcc, s[6:7]\n\tv_mov_b32_e32 v9, 0\n\tv_cndmask_b32_e32 v7, 0, v8, vcc\n\ts_endpgm\n")
    assert subprocess.run([sys.executable, CHECK, str(bad)], capture_output=True).returncode == 1
    assert subprocess.run([sys.executable, CHECK, str(ok)], capture_output=True).returncode == 0


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", ["prmers_amd/csrc/kernels_v2.hip", "prmers_amd/csrc/kernels_v3.hip", "prmers_amd/csrc/kernels_v5.hip", "prmers_amd/csrc/kernels.hip", "prmers_amd/csrc/selftest.hip"])
def test_shipped_kernels_have_no_unpadded_sgpr_hazard(src):
    extra = _v2_flags() if src.endswith("kernels_v2.hip") else ()
    with tempfile.TemporaryDirectory() as td:
        out = subprocess.run([sys.executable, CHECK, _isa(os.path.join(ROOT, src), td, extra)], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout[-2000:]


def _v2_flags():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import valu_model
    return tuple(valu_model.V2_FLAGS)


def test_the_model_and_the_hazard_scan_use_the_flags_of_the_makefile():
    """kernels_v2.hip is built with its own scheduler strategy (Makefile V2FLAGS); the ISA walks must look at the same code"""
    mk = open(os.path.join(ROOT, "prmers_amd", "csrc", "Makefile")).read()
    assert "V2FLAGS  = " + " ".join(_v2_flags()) in mk


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_committed_valu_model_matches_the_shipped_kernels(tmp_path):
    """profiles/valu_model_latest.json (read by bench.py for roofline.valu) must describe the kernels in the tree: the ISA walk of
    tools/valu_model.py is repeated here and compared with the committed figures (instructions and issue cycles per wave, 1 %)."""
    import json
    out = tmp_path / "model.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "valu_model.py"), "--out", str(out)], stdout=subprocess.DEVNULL)
    fresh = json.load(open(out))
    committed = json.load(open(os.path.join(ROOT, "profiles", "valu_model_latest.json")))
    assert fresh["plan"] == committed["plan"]
    for slot, k in fresh["kernels"].items():
        c = committed["kernels"][slot]
        assert abs(k["valu_insts_per_wave"] - c["valu_insts_per_wave"]) <= 0.01 * c["valu_insts_per_wave"], (slot, k, c)
        assert abs(k["issue_cycles_per_wave"] - c["issue_cycles_per_wave"]) <= 0.01 * c["issue_cycles_per_wave"], (slot, k, c)
