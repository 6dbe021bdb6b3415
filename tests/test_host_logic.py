"""Host-side logic of the product (no GPU): field primitives, plan policy, C-ABI exports."""
import ctypes
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "prmers_amd", "csrc")


def _build_and_run(src, args=()):
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "t")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + CSRC, "-o", exe, os.path.join(ROOT, "tests", "host", src)])
        return subprocess.check_output([exe, *args]).decode()


def test_field_primitives_and_dft8():
    """gf::mul / mul_u32 / mul_pow2 (all 192 shifts) / dft8 against 128-bit reference arithmetic."""
    out = _build_and_run("test_primitives.cpp")
    assert out.strip().startswith("OK"), out


def test_plan_policy_matches_reference_sizes():
    out = _build_and_run("plan_dump.cpp")
    sizes = dict(re.findall(r"p=(\d+) marin-hip:n=(\d+)", out))
    # values printed by the reference's ibdwt::transform_size
    assert sizes["127"] == "8" and sizes["9815459"] == "524288"
    assert sizes["136279841"] == "8388608" and sizes["205271257"] == "10485760"
    for line in out.splitlines():
        m = re.search(r"m1=(\d+):m2=(\d+):c=(\d+).*lds_front=(\d+) lds_mid=(\d+)", line)
        n = int(re.search(r"n=(\d+)", line).group(1))
        m1, m2, c, lf, lm = map(int, m.groups())
        assert m1 * m2 * 2 == n and m2 % c == 0 and lf <= 160 * 1024 and lm <= 160 * 1024


def test_plan_policy_of_the_sizes_beyond_rows_of_4096():
    """columns of 2560 = 5 x 512 keep the rows of n = 5 2^22 at 4096 (round 4); from 2^25 on the rows are 8192 wide; n = 5 2^26 takes the split sweeps"""
    from prmers_amd import resolve_plan
    want = {205271257: "n=10485760:m1=1280:m2=4096:c=4", 250000013: "n=16777216:m1=2048:m2=4096:c=2", 332000003: "n=20971520:m1=2560:m2=4096:c=2",
            600000001: "n=33554432:m1=2048:m2=8192:c=2", 700000001: "n=41943040:m1=2560:m2=8192:c=2", 1300000003: "n=83886080:m1=5120:m2=8192:c=2",
            4000000007: "n=335544320:m1=20480:m2=8192:c=1:split5"}
    for p, plan in want.items():
        assert resolve_plan(p) == "marin-hip:" + plan, (p, resolve_plan(p))
    assert resolve_plan(332000003, "m2=8192") == "marin-hip:n=20971520:m1=1280:m2=8192:c=4"   # the round-3 plan stays selectable


def test_prime_factor_maps_of_the_radix5_columns():
    """kernels_v5.hip Shape<J>::PU / PV (read from the source): the Good-Thomas maps of the 5 x L column split -- input i1 = (L d0 + 5 r) mod M1 and
    output frequency (PU k0 + PV kr) mod M1 are bijections, PU / PV are the CRT idempotents, and with them the length-M1 DFT factors into a
    DFT-5 and a DFT-L with no twiddle in between (checked numerically over a small prime field for L = 8 with the same construction)."""
    src = open(os.path.join(ROOT, "prmers_amd", "csrc", "kernels_v5.hip")).read()
    m = re.search(r"PU = J \? (\d+)u : (\d+)u, PV = (\d+)u", src)
    pu1, pu0, pv = map(int, m.groups())
    for L, pu in ((256, pu0), (512, pu1)):
        M1 = 5 * L
        assert pu % 5 == 1 and pu % L == 0 and pv % L == 1 and pv % 5 == 0
        assert sorted((L * d0 + 5 * r) % M1 for d0 in range(5) for r in range(L)) == list(range(M1))
        assert sorted((pu * k0 + pv * kr) % M1 for k0 in range(5) for kr in range(L)) == list(range(M1))
        assert pu * 4 + pv * (L - 1) < (1 << 20)          # the unreduced labels stay below the reach of the root table (plan.hpp TWhi)
    # the factorisation itself, on a toy field: q = 41, N = 40 = 5 x 8, omega of order 40
    q, N, L = 41, 40, 8
    w = next(g for g in range(2, q) if pow(g, N, q) == 1 and all(pow(g, N // f, q) != 1 for f in (2, 5)))
    u, v = pow(L, -1, 5), pow(5, -1, L)
    PU, PV = L * u, 5 * v
    x = [(7 * i * i + 3 * i + 1) % q for i in range(N)]
    X = [sum(x[n] * pow(w, n * k, q) for n in range(N)) % q for k in range(N)]
    w5, wL = pow(w, N // 5, q), pow(w, 5, q)
    for k0 in range(5):
        for kr in range(L):
            acc = 0
            for d0 in range(5):
                for r in range(L):
                    acc += x[(L * d0 + 5 * r) % N] * pow(w5, d0 * k0, q) * pow(wL, r * kr, q)
            assert acc % q == X[(PU * k0 + PV * kr) % N]
    # the in-place form of the generic kernels (kernels.hip lds_radix5): output k0 of group t goes to slot L k0 + (5 t mod L), the block transform
    # then leaves the frequency (lab_u k0 + 5 k) mod N in slot (k0, k), lab_u = L (L^-1 mod 5) -- engine.hip sets exactly that
    slots = [0] * N
    for t in range(L):
        for k0 in range(5):
            slots[L * k0 + (5 * t) % L] = sum(x[(L * d0 + 5 * t) % N] * pow(w5, d0 * k0, q) for d0 in range(5)) % q
    lab_u = L * u
    for k0 in range(5):
        for k in range(L):
            z = sum(slots[L * k0 + rho] * pow(wL, rho * k, q) for rho in range(L)) % q
            assert z == X[(lab_u * k0 + 5 * k) % N]


def test_plan_weight_tables_and_digit_info_words():
    """plan.hpp: digit widths, both factorisations of the IBDWT weights (incl. the second half of SA/TA used
    for odd digits), the inverse tables and the 2-bit-per-digit DI words, digit by digit against the defining
    formulas (ibdwt.h:111-147: width = ceil(p(j+1)/n) - ceil(pj/n), w_j = 2^((n - pj mod n)/n))."""
    out = _build_and_run("test_plan_tables.cpp")
    assert out.strip().endswith("OK"), out
    assert "di_checked=16384" in out and "di_checked=32768" in out   # the register-resident column shapes were covered
    assert "m1=256:m2=16:c=4 digits=8192 di_checked=8192" in out       # radix-4 columns
    assert "m1=1280:m2=8:c=4 digits=20480 di_checked=20480" in out and "m1=2560:m2=8:c=2 digits=40960 di_checked=40960" in out   # radix-5 columns


def test_product_library_has_no_experimental_code_and_the_experiments_still_build():
    """The experiments that did not make the product (cooperative one-launch squaring, back + front in one launch) live behind
    -DMI355_EXPERIMENTAL: the product library exports and contains nothing of them, and their sources still compile for gfx950 (host and device passes, -fsyntax-only; `make -C prmers_amd/csrc exp` builds the library)
    (their parity checks run on a GPU box: tools/exp_coop_check.py)."""
    from prmers_amd import engine as E
    if not os.path.exists(E.LIB_PATH):
        pytest.skip("libmi355_engine.so not built")
    syms = subprocess.run(["nm", "-D", "--defined-only", E.LIB_PATH], capture_output=True, text=True).stdout.lower()
    assert "coop" not in syms and "chain" not in syms and "k31" not in syms
    blob = open(E.LIB_PATH, "rb").read()
    for name in (b"k_coop", b"k31_cols", b"MI355_COOP", b"MI355_CHAIN"):
        assert name not in blob, name
    csrc = os.path.join(ROOT, "prmers_amd", "csrc")
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-DMI355_EXPERIMENTAL", "-fsyntax-only", "-Wno-unused-command-line-argument",
                               os.path.join(csrc, f)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for f in ("kernels.hip", "kernels_v2.hip", "kernels_v3.hip", "engine.hip")]
    for pr in procs:
        out, _ = pr.communicate()
        assert pr.returncode == 0, out[-2000:]


def test_c_abi_exports_and_no_gpu_behaviour():
    """the library loads, exports every symbol the header declares, resolves plans without a GPU and
    refuses to create an engine without one (no CPU fallback)."""
    from prmers_amd import engine as E
    if not os.path.exists(E.LIB_PATH):
        pytest.skip("libmi355_engine.so not built")
    lib = ctypes.CDLL(E.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "mi355_engine.h")).read()
    declared = sorted(set(re.findall(r"\b(mi355_(?:engine|crt)_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations found"
    for name in declared:
        getattr(lib, name)
    assert sorted(E.EXPORTS) == declared
    assert E.resolve_plan(136279841) == "marin-hip:n=8388608:m1=1024:m2=4096:c=4"
    assert "n=8" in E.resolve_plan(127)
    with pytest.raises(E.EngineError):
        E.resolve_plan(136279841, "m2=3")
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(E.EngineError, match="no CPU fallback|HIP"):
            E.Engine(127, 2)
        with pytest.raises(E.EngineError, match="no CPU fallback|HIP"):
            E.CrtEngine(1279, 9)
    assert "crt-hip:n=9437184:odd=9" == E.resolve_plan(205271257, "crt:9") and "n=6291456" in E.resolve_plan(205271257, "crt:3")


def _build_adapter(td):
    exe = os.path.join(td, "t_adapter")
    gmp = "/usr/lib/x86_64-linux-gnu/libgmp.so.10"
    inc = [i for i in ("/opt/conda/include", "/usr/include") if os.path.exists(os.path.join(i, "gmp.h"))]
    if not inc or not os.path.exists(gmp):
        pytest.skip("gmp headers/library not available")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + inc[0], "-o", exe,
                           os.path.join(ROOT, "tests", "host", "test_engine_adapter.cpp"), "-ldl", gmp])
    return exe


def test_selftest_needs_a_device():
    """without a GPU the device self-test reports failure through last_error instead of crashing"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from prmers_amd.engine import load_library
    L = load_library()
    assert L.mi355_engine_selftest(0) == 0 and L.mi355_engine_last_error()


def test_cpp_adapter_builds_and_binds_every_symbol():
    """include/mi355/engine_hip.h compiles against the engine interface and dlsym-binds the C ABI."""
    from prmers_amd import engine as E
    if not os.path.exists(E.LIB_PATH):
        pytest.skip("libmi355_engine.so not built")
    with tempfile.TemporaryDirectory() as td:
        exe = _build_adapter(td)
        out = subprocess.run([exe, "load", E.LIB_PATH], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_adapter_reg_contract_on_gpu():
    """tests/test_aevum_reg_adapter.cpp:32-93 through engine_hip on the MI355X."""
    from prmers_amd import engine as E
    with tempfile.TemporaryDirectory() as td:
        exe = _build_adapter(td)
        out = subprocess.run([exe, "run", E.LIB_PATH], capture_output=True, text=True)
        assert out.returncode == 0 and "passed" in out.stdout, out.stdout + out.stderr


def _build_cli(td):
    exe = os.path.join(td, "mi355_prp")
    gmp = "/usr/lib/x86_64-linux-gnu/libgmp.so.10"
    inc = [i for i in ("/opt/conda/include", "/usr/include") if os.path.exists(os.path.join(i, "gmp.h"))]
    if not inc or not os.path.exists(gmp):
        pytest.skip("gmp headers/library not available")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), "-I" + inc[0], "-o", exe,
                           os.path.join(ROOT, "examples", "prp_cli.cpp"), "-ldl", gmp])
    return exe


def _build_oracle_shim(td):
    """tests/host/oracle_abi_shim.c + oracle/oracle.c -> a library with the engine's C ABI served by the CPU oracle (tests only)"""
    so = os.path.join(td, "liboracle_abi_shim.so")
    subprocess.check_call(["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-fvisibility=hidden", "-o", so,
                           os.path.join(ROOT, "tests", "host", "oracle_abi_shim.c"), os.path.join(ROOT, "oracle", "oracle.c"), "-lm"])
    return so


def test_cpp_driver_resume_rollback_and_sigint_on_the_oracle(tmp_path):
    """examples/prp_cli.cpp on CPU (engine_hip loading the oracle-backed ABI shim): (1) a partial run, a resume and an
    injected error right after it roll back to the rollback point saved WITH the checkpoint (RunPrpOrLlMarin.cpp:251-255) and
    end on the right residue; (2) SIGINT ends a running test with exit code 0 and a checkpoint (:296-309), and the resumed
    run prints the reference's golden M100003 result line (unit_tests.sh:140-141)."""
    import json
    import signal
    import time
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")))
    exe, shim = _build_cli(str(tmp_path)), _build_oracle_shim(str(tmp_path))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    run = lambda *a: subprocess.run([exe, *a, "-lib", shim], capture_output=True, text=True, cwd=str(tmp_path), env=env)   # noqa: E731
    # (1) p = 9941, checks at every block boundary; checkpoint at 5050 (inside a block), fault at 5060
    o1 = run("9941", "-checklevel", "1", "-ckpt", str(tmp_path), "-maxiters", "5050")
    assert o1.returncode == 0 and "partial run" in o1.stdout and (tmp_path / "m_9941.ckpt.gl").exists(), o1.stdout + o1.stderr
    o2 = run("9941", "-checklevel", "1", "-ckpt", str(tmp_path), "-erroriter", "5060")
    assert o2.returncode == 0, o2.stdout + o2.stderr
    assert "Resuming from a checkpoint at iteration 5050" in o2.stdout and "Injected error at iteration 5060" in o2.stdout
    restores = [l for l in o2.stdout.splitlines() if l.startswith("[Gerbicz Li] Restore")]
    assert len(restores) == 1 and restores[0] == "[Gerbicz Li] Restore iter=4990 (j=4950)"   # last boundary before 5050: j = 9940 - iter = 50 * 99
    assert "probably prime" in o2.stdout and "gerbicz_errors=1" in o2.stdout
    # the same without the side file: the rollback lands on the resumed state itself
    o1 = run("9941", "-checklevel", "1", "-ckpt", str(tmp_path), "-maxiters", "5050")
    os.remove(tmp_path / "m_9941.ckpt.gl")
    o3 = run("9941", "-checklevel", "1", "-ckpt", str(tmp_path), "-erroriter", "5060")
    assert "[Gerbicz Li] Restore iter=5049 (j=4891)" in o3.stdout and "probably prime" in o3.stdout and "gerbicz_errors=1" in o3.stdout, o3.stdout
    # (2) SIGINT while M100003 runs, then the resume
    proc = subprocess.Popen([exe, "100003", "-ckpt", str(tmp_path), "-json", "results.json.txt", "-lib", shim], stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True, cwd=str(tmp_path), env=env)
    time.sleep(3.0)
    proc.send_signal(signal.SIGINT)
    out, err = proc.communicate(timeout=120)
    assert proc.returncode == 0 and "Interrupted by user, state saved at iteration" in out, out + err
    assert (tmp_path / "m_100003.ckpt").exists() and not (tmp_path / "results.json.txt").exists()
    o4 = run("100003", "-ckpt", str(tmp_path), "-json", "results.json.txt")
    assert o4.returncode == 0 and "Resuming from a checkpoint" in o4.stdout, o4.stdout + o4.stderr
    line = json.loads((tmp_path / "results.json.txt").read_text().splitlines()[-1])
    m = gold["m100003"]
    assert line["res64"] == m["res64"] == "1CF45E9503C71FD6" and line["res2048"] == m["res2048"].lower() and line["status"] == "C"
    assert line["errors"] == {"gerbicz": 0}


def test_cpp_driver_ll_safe_on_the_oracle(tmp_path):
    """examples/prp_cli.cpp -llsafe on CPU (the oracle-backed ABI shim): Lucas-Lehmer with block re-computation
    (src/modes/RunLlSafeMarin.cpp:95-392) -- verdicts of unit_tests.sh:5-14 exponents, an injected error caught at the block boundary and
    rolled back to the block start, the same residue as the Python twin (prmers_amd/prp.py run_ll_safe)."""
    import json
    exe, shim = _build_cli(str(tmp_path)), _build_oracle_shim(str(tmp_path))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    run = lambda *a: subprocess.run([exe, *a, "-lib", shim], capture_output=True, text=True, cwd=str(tmp_path), env=env)   # noqa: E731
    o = run("521", "-llsafe", "-llsafe_block", "50", "-erroriter", "120", "-json", "r.json")
    assert o.returncode == 0, o.stdout + o.stderr
    assert "Injected error at iteration 120" in o.stdout and "[Error check] Check FAILED! iter=149" in o.stdout and "[Error check] Restore iter=100" in o.stdout
    assert "LL-safe: prime" in o.stdout and "errors=1" in o.stdout
    line = json.loads((tmp_path / "r.json").read_text().splitlines()[-1])
    assert line["status"] == "P" and line["res64"] == "0000000000000000"
    o = run("523", "-llsafe")
    assert "LL-safe: composite" in o.stdout and "errors=0" in o.stdout, o.stdout + o.stderr
    import orc
    from prmers_amd import prp
    ref = prp.run_ll_safe(orc.OracleEngine(523, 8), 523)
    assert ("res64=%s" % ref["res64"]) in o.stdout, (ref["res64"], o.stdout)
    # an error in the very first block rolls back to iteration 0
    o = run("127", "-llsafe", "-llsafe_block", "40", "-erroriter", "3")
    assert "[Error check] Restore iter=0" in o.stdout and "LL-safe: prime" in o.stdout, o.stdout


def test_cpp_driver_fails_loudly_without_gpu():
    from prmers_amd import engine as E
    import torch
    if torch.cuda.is_available() or not os.path.exists(E.LIB_PATH):
        pytest.skip("needs the built library and no GPU")
    with tempfile.TemporaryDirectory() as td:
        out = subprocess.run([_build_cli(td), "127", "-lib", E.LIB_PATH], capture_output=True, text=True)
        assert out.returncode == 2 and "no CPU fallback" in out.stderr


@pytest.mark.gpu
def test_cpp_driver_prp_ll_and_fault_injection_on_gpu():
    """examples/prp_cli.cpp (engine_hip + the reference's driver loop in C++): M9941 with -erroriter reproduces the
    golden messages of unit_tests.sh:24-50; LL and composite cases."""
    from prmers_amd import engine as E
    with tempfile.TemporaryDirectory() as td:
        exe = _build_cli(td)
        run = lambda *a: subprocess.run([exe, *a, "-lib", E.LIB_PATH], capture_output=True, text=True)   # noqa: E731
        o = run("9941", "-erroriter", "55")
        assert o.returncode == 0, o.stderr
        assert "Injected error at iteration 55" in o.stdout and "[Gerbicz Li] Check FAILED! iter=9941" in o.stdout
        assert "[Gerbicz Li] Restore iter=0 (j=9940)" in o.stdout and "probably prime" in o.stdout
        assert "probably prime" in run("607", "-ll").stdout
        assert "composite" in run("1001").stdout
        # -d: the reference's device selector (src/io/CliParser.cpp:198); a device that does not exist fails loudly (exit 2, src/main.cpp:159-164)
        assert "probably prime" in run("521", "-d", "0").stdout
        bad = run("521", "-d", "99")
        assert bad.returncode == 2 and "device" in bad.stderr.lower(), bad.stdout + bad.stderr
        # LL-safe (src/modes/RunLlSafeMarin.cpp:95-392): block re-computation, an injected error is caught and rolled back
        o = run("2203", "-llsafe", "-llsafe_block", "100", "-erroriter", "150")
        assert o.returncode == 0 and "Injected error at iteration 150" in o.stdout and "[Error check] Check FAILED! iter=199" in o.stdout, o.stdout + o.stderr
        assert "[Error check] Restore iter=100" in o.stdout and "LL-safe: prime" in o.stdout and "errors=1" in o.stdout
        assert "LL-safe: composite" in run("2207", "-llsafe").stdout


REF_INCLUDE = "/root/reference/include"
AEVUM_NAMES = ["version", "last_error", "resolve_fft", "create", "destroy", "transform_size", "word_count", "sync", "set_u32", "set_words",
               "get_words", "copy", "prepare", "square_mul", "mul", "add", "sub_reg", "sub_u32", "equal"]


def test_cpp_adapter_overrides_the_reference_engine_header():
    """The adapter compiled against the reference's REAL abstract class (include/marin/engine.h:16-303, where it lies,
    nothing copied): every pure virtual is overridden, and the test program of the stand-alone build links unchanged.
    Build container only (the reference tree does not travel to the GPU box)."""
    if not os.path.isdir(os.path.join(REF_INCLUDE, "marin")):
        pytest.skip("reference tree not present")
    gmp = "/usr/lib/x86_64-linux-gnu/libgmp.so.10"
    inc = [i for i in ("/opt/conda/include", "/usr/include") if os.path.exists(os.path.join(i, "gmp.h"))]
    if not inc or not os.path.exists(gmp):
        pytest.skip("gmp headers/library not available")
    from prmers_amd import engine as E
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "t_adapter_ref")
        subprocess.check_call(["g++", "-std=c++20", "-O1", "-DMI355_USE_REFERENCE_ENGINE_H", "-I" + os.path.join(ROOT, "include"), "-I" + REF_INCLUDE,
                               "-I" + os.path.join(REF_INCLUDE, "marin"), "-I" + inc[0], "-o", exe,
                               os.path.join(ROOT, "tests", "host", "test_engine_adapter.cpp"), "-ldl", gmp])
        if os.path.exists(E.LIB_PATH):
            out = subprocess.run([exe, "load", E.LIB_PATH], capture_output=True, text=True)
            assert out.returncode == 0, out.stdout + out.stderr
        # engine_hip must be concrete against the real header: instantiating it compiles only if no pure virtual is left
        src = os.path.join(td, "concrete.cpp")
        with open(src, "w") as f:
            f.write('#define MI355_USE_REFERENCE_ENGINE_H\n#include "mi355/engine_hip.h"\n#include <type_traits>\n'
                    'static_assert(!std::is_abstract<engine_hip>::value, "engine_hip leaves a pure virtual of marin/engine.h unimplemented");\n'
                    'static_assert(std::is_base_of<engine, engine_hip>::value, "engine_hip must derive from the reference engine");\nint main() { return 0; }\n')
        subprocess.check_call(["g++", "-std=c++20", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), "-I" + REF_INCLUDE,
                               "-I" + os.path.join(REF_INCLUDE, "marin"), "-I" + inc[0], src])


def test_alias_shim_exports_the_reference_plugin_abi():
    """examples/mi355_as_aevum.c on top of libmi355_engine.so exports the 19 aevum_engine_* symbols the reference's adapter
    binds (third_party/aevum/src/EngineApi.h:28-59; checked the way third_party/aevum/tests/engine_api_load_test.cpp:13-60
    does: dlopen RTLD_NOW|RTLD_LOCAL + dlsym of every name, then version / resolve_fft through the aliases)."""
    from prmers_amd import engine as E
    if not os.path.exists(E.LIB_PATH):
        pytest.skip("libmi355_engine.so not built")
    import ctypes
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "libaevum_engine.so")
        libdir = os.path.dirname(E.LIB_PATH)
        subprocess.check_call(["gcc", "-shared", "-fPIC", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "examples", "mi355_as_aevum.c"), "-o", so, "-L" + libdir,
                               "-l:" + os.path.basename(E.LIB_PATH), "-Wl,-rpath," + libdir])
        lib = ctypes.CDLL(so, mode=os.RTLD_NOW | os.RTLD_LOCAL)
        for name in AEVUM_NAMES:
            getattr(lib, "aevum_engine_" + name)
        lib.aevum_engine_version.restype = ctypes.c_char_p
        assert b"mi355" in lib.aevum_engine_version()
        buf = ctypes.create_string_buffer(96)
        assert lib.aevum_engine_resolve_fft(ctypes.c_uint32(136279841), None, buf, ctypes.c_size_t(96)) == 1
        assert b"n=8388608" in buf.value
        if os.path.isdir("/root/reference/third_party/aevum/src"):
            # same prototypes as the reference's header: compile a translation unit that includes BOTH declarations
            src = os.path.join(td, "proto.c")
            with open(src, "w") as f:
                f.write('#include "EngineApi.h"\n#define MI355_ENGINE_H_SHIM\n#include "mi355_engine.h"\n' +
                        "".join("static __typeof__(aevum_engine_%s)* p_%s = (__typeof__(aevum_engine_%s)*)mi355_engine_%s;\n" % (n, n, n, n)
                                for n in AEVUM_NAMES if n not in ("create",)) + "int main(void) { return 0; }\n")
            subprocess.check_call(["gcc", "-fsyntax-only", "-Werror=incompatible-pointer-types", "-I/root/reference/third_party/aevum/src",
                                   "-I" + os.path.join(ROOT, "include"), src])


def test_cpp_caller_formats_match_the_python_mirror_and_known_answers(tmp_path):
    """include/mi355/caller_formats.h (C++ side of SURVEY.md 8f N3: checkpoint file v2 + CRC, worktodo entries and rotation,
    result JSON, proof checkpoints, residue words) against prmers_amd/prp.py -- itself pinned by the reference's golden
    vectors (tests/test_prp_driver.py) -- and against known answers."""
    import json
    import numpy as np
    from prmers_amd import prp
    exe = str(tmp_path / "t_formats")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "host", "test_caller_formats.cpp")])
    work = tmp_path / "w"
    work.mkdir()
    out = subprocess.run([exe, str(work)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout)
    assert r["crc_check"] == "CBF43926"                      # CRC-32 of "123456789"
    # words / type-1 residue / hex: the same digit vector through the Python mirror
    w = [16] * 7 + [15]
    d = np.array([(w[i] << 32) | ((0x9e37 * (i + 3) + 77) & ((1 << w[i]) - 1)) for i in range(8)], dtype=np.uint64)
    W = prp.pack_words(d, 127)
    assert r["res64_raw"] == prp.format_res64(W)
    W9 = prp.prp3_div9(127, W)
    assert r["res64_div9"] == prp.format_res64(W9) and r["res2048_div9"] == prp.format_res2048(W9)
    x = sum(int(v & 0xFFFFFFFF) << (16 * i) for i, v in enumerate(d)) % ((1 << 127) - 1)
    assert int(r["res64_div9"], 16) == (x * pow(9, -1, (1 << 127) - 1)) % ((1 << 127) - 1) & ((1 << 64) - 1)
    # checkpoint file
    c = r["ckpt"]
    assert c == {"saved": 1, "load": 0, "iteration": 2345, "elapsed": 6.5, "same": 1, "old": 0, "old_iteration": 1234, "wrong_mode": -3,
                 "wrong_exponent": -2, "corrupt": -2, "missing": -1, "name_ll": "llunsafe_m_607.ckpt"}
    # worktodo lines: same verdicts as the Python parser
    lines = ["PRP=1,2,136279841,-1", "PRP=N/A,1,2,9941,-1,75,0", "PRPDC=0123456789ABCDEF0123456789abcdef,1,2,521,-1", "Test=607",
             "DoubleCheck=AID,1279,70,1", "Pfactor=1,2,999,-1,70,2", "# PRP=1,2,127,-1", "", "PRP=1,3,127,-1", "Test=N/A,2203,75,1"]
    for line, got in zip(lines, r["worktodo"]):
        want = prp.parse_worktodo_line(line)
        assert bool(got[0]) == (want is not None), line
        if want:
            assert (("ll" if got[1] else "prp"), got[2]) == want, line
    assert r["worktodo"][2][3] == "0123456789ABCDEF0123456789abcdef"
    assert r["rotate"] == {"removed": 1, "more1": 1, "next": 607, "more2": 1, "more3": 0, "saved": "PRP=1,2,127,-1|# note|Test=607|"}
    # result lines: key for key what the Python mirror prints
    res = {"exponent": 100003, "mode": "prp", "is_prime": False, "res64": "1CF45E9503C71FD6", "res2048": "ab", "gerbicz_errors": 1}
    want = json.loads(prp.result_json(res, 8192, program_version="v", user="u", computer="c", aid="a", timestamp="t"))
    got = json.loads(r["json_prp"])
    assert list(got.keys())[:9] == ["status", "exponent", "worktype", "res64", "res2048", "residue-type", "errors", "shift-count", "fft-length"]
    for k in ("status", "exponent", "worktype", "res64", "res2048", "residue-type", "errors", "shift-count", "fft-length", "user", "computer", "aid", "timestamp"):
        assert got[k] == want[k], k
    ll = json.loads(r["json_ll"])
    assert ll["status"] == "P" and ll["worktype"] == "LL" and "res2048" not in ll and "user" not in ll
    # proof points: ProofSetMarin.cpp:64-84 restated independently
    E, power = 9941, 3
    pts, span = [0], (E + 1) // 2
    for _ in range(power):
        pts += [q + span for q in pts]
        span = (span + 1) // 2
    pts[0] = E
    assert r["proof_points"] == sorted(pts) and len(pts) == 1 << power and r["proof_points"][-1] == E
    assert r["proof"] == {"saved": 1, "other_iteration_ignored": 1, "round_trip": 1, "corruption_caught": 1, "valid_to_before": 1,
                          "file": "9941/proof/%d" % sorted(pts)[2]}


@pytest.mark.gpu
def test_cpp_driver_on_the_crt_family_through_the_same_adapter(tmp_path):
    """the C++ caller (examples/prp_cli.cpp over include/mi355/engine_hip.h) with -fft crt:9 / crt:3: the same adapter, the same loop,
    the GF(M61^2) x GF(M31^2) engine behind it -- PRP of M9941 (prime) with Gerbicz-Li checks, LL of M9689, an injected error repaired"""
    from prmers_amd import engine as E
    exe = _build_cli(str(tmp_path))
    run = lambda *a: subprocess.run([exe, *a, "-lib", E.LIB_PATH], capture_output=True, text=True, cwd=str(tmp_path))   # noqa: E731
    o = run("9941", "-fft", "crt:9:words=576", "-checklevel", "1")
    assert o.returncode == 0 and "probably prime" in o.stdout and "Check passed" in o.stdout, o.stdout + o.stderr
    o = run("9689", "-ll", "-fft", "crt:3:words=384")
    assert o.returncode == 0 and "probably prime" in o.stdout, o.stdout + o.stderr
    o = run("9949", "-fft", "crt:9:words=576", "-checklevel", "1", "-erroriter", "5000")
    assert o.returncode == 0 and "composite" in o.stdout and "Check FAILED" in o.stdout and "gerbicz_errors=1" in o.stdout, o.stdout + o.stderr
    o = run("9941", "-fft", "crt:7")
    assert o.returncode != 0 and "odd radix" in (o.stdout + o.stderr)


@pytest.mark.gpu
def test_cpp_driver_worktodo_checkpoint_proof_and_json_on_gpu(tmp_path):
    """examples/prp_cli.cpp with the caller-side formats (include/mi355/caller_formats.h): a worktodo entry is run in two
    slices through a version-2 checkpoint, leaves the proof residues, the reference's golden result line of M100003
    (unit_tests.sh:140-141) and a rotated worktodo."""
    import json
    from prmers_amd import engine as E
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")))
    exe = _build_cli(str(tmp_path))
    wt = tmp_path / "worktodo.txt"
    wt.write_text("PRP=0123456789ABCDEF0123456789ABCDEF,1,2,100003,-1,75,0\nTest=607\n")
    run = lambda *a: subprocess.run([exe, *a, "-lib", E.LIB_PATH], capture_output=True, text=True, cwd=str(tmp_path))   # noqa: E731
    o1 = run("-worktodo", str(wt), "-ckpt", str(tmp_path), "-proof", "2", "-maxiters", "60000", "-json", "results.json.txt")
    assert o1.returncode == 0 and "partial run" in o1.stdout, o1.stdout + o1.stderr
    assert (tmp_path / "m_100003.ckpt").exists() and wt.read_text().startswith("PRP=")          # not finished: nothing rotated
    o2 = run("-worktodo", str(wt), "-ckpt", str(tmp_path), "-proof", "2", "-json", "results.json.txt")
    assert o2.returncode == 0 and "Resuming from a checkpoint" in o2.stdout, o2.stdout + o2.stderr
    line = json.loads((tmp_path / "results.json.txt").read_text().splitlines()[-1])
    m = gold["m100003"]
    assert line["exponent"] == 100003 and line["worktype"] == "PRP-3" and line["status"] == "C"
    assert line["res64"] == "1CF45E9503C71FD6" and line["residue-type"] == 1 and line["errors"] == {"gerbicz": 0}
    assert line["aid"] == "0123456789ABCDEF0123456789ABCDEF" and line["fft-length"] == 4096
    assert line["res64"] == m["res64"] and line["res2048"] == m["res2048"].lower()
    assert wt.read_text() == "Test=607\n" and "100003" in (tmp_path / "worktodo_save.txt").read_text()
    pts = sorted(int(f.name) for f in (tmp_path / "100003" / "proof").iterdir())
    assert pts == [25001, 50002, 75003, 100003]            # the 2^2 points of ProofSetMarin.cpp:64-84 for E = 100003
    assert (tmp_path / "100003" / "proof" / "25001").stat().st_size == 4 + 4 * ((100003 + 31) // 32)
    o3 = run("-worktodo", str(wt))
    assert "probably prime" in o3.stdout and wt.read_text() == ""


def test_automatic_pfa_radix_follows_the_reference_policy_at_the_exact_boundaries():
    """fft_spec "crt" picks radix 9 / 3 / none by the reference's stock-to-PFA size-ratio gates (1.60 / 1.30, README.md:888-926).  The
    capacity of a size is this engine's own worst-case rule (log2 n + 2 (p/n + 1) < 92; the reference's boundaries come from its measured
    bits-per-word tables and lie ~15 % higher), so the table is checked row by row at OUR exact boundaries: the largest exponent a size
    admits selects exactly the (radix, words) pair of the reference's row, the next exponent moves on to the next larger size."""
    import math
    from prmers_amd import resolve_plan

    def admits(n, p):
        return n <= p and math.log2(n) + 2.0 * (p / n + 1.0) < 92.0

    def max_exponent(n):
        lo, hi = n, 60 * n
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if admits(n, mid):
                lo = mid
            else:
                hi = mid - 1
        return lo

    def auto(p):
        m = re.fullmatch(r"crt-hip:n=(\d+):odd=(\d+)", resolve_plan(p, "crt"))
        return int(m.group(2)), int(m.group(1))

    # (radix, words) of every row of README.md:907-922 that fits 32-bit exponents, in the table's order
    rows = [(3, 393216), (3, 786432), (9, 1179648), (3, 1572864), (9, 2359296), (3, 3145728), (9, 4718592), (3, 6291456), (9, 9437184),
            (3, 12582912), (9, 18874368), (3, 25165824), (9, 37748736), (9, 75497472)]
    sizes = sorted(set([1 << k for k in range(10, 28)] + [3 << k for k in range(8, 26)] + [9 << k for k in range(7, 25)]))
    for odd, n in rows:
        top = max_exponent(n)
        if top >= 2**32:
            continue
        assert auto(top) == (odd, n), (odd, n, top)
        bigger = min(x for x in sizes if x > n)
        if top + 1 < 2**32 and max_exponent(bigger) < 2**32:
            assert auto(top + 1)[1] == bigger, (n, top + 1)
    # in between the gates keep the stock plan: a power of two is chosen exactly when it is the smallest admissible size of the three families
    for k in (17, 20, 23):
        n = 1 << k
        assert auto(max_exponent(n)) == (1, n)
        below = max(x for x in sizes if x < n)          # 3 * 2^(k-2), the next smaller size
        assert auto(max_exponent(below) + 1) == (1, n) and below == 3 * n // 4
    # BASELINE configs[3]: the automatic plan of p = 205271257 is the radix-3 size of the reference's own table row (README.md:916)
    assert auto(205271257) == (3, 6291456)
    assert resolve_plan(205271257, "crt:9") == "crt-hip:n=9437184:odd=9" and resolve_plan(205271257, "crt:auto") == "crt-hip:n=6291456:odd=3"


def test_automatic_pfa_plan_against_the_published_ranges_of_the_reference():
    """the gap between this engine's automatic plan and the reference's published `pfa:auto` ranges (README.md:907-922), row by
    row.  The gates are the reference's (1.30 / 1.60), but a size admits fewer bits per word here (worst-case rule log2 n + 2 (p/n + 1) < 92,
    ~34 bits; the reference's measured tables admit ~39), so: at the START of every radix-3 row the same plan is chosen; at the END of a
    radix-3 row this engine has already moved to the next stock size; inside the radix-9 rows it takes the radix-3 size 4/3 as large (the
    radix-9 size no longer admits the exponent).  Residues are identical either way -- only the transform length differs."""
    from prmers_amd import resolve_plan

    def auto(p):
        m = re.fullmatch(r"crt-hip:n=(\d+):odd=(\d+)", resolve_plan(p, "crt"))
        return int(m.group(2)), int(m.group(1))

    # (radix, first p, last p, words) of the reference's table
    rows = [(3, 10627319, 15724707, 393216), (3, 21071135, 31284264, 786432), (9, 41922069, 46560704, 1179648), (3, 46560705, 62080936, 1572864),
            (9, 83194017, 92625960, 2359296), (3, 92625961, 123343992, 3145728), (9, 165507233, 183789168, 4718592),
            (3, 183789169, 244737648, 6291456), (9, 328414017, 365879616, 9437184), (3, 365879617, 487210368, 12582912),
            (9, 653808129, 725153152, 18874368), (3, 725153153, 965612672, 25165824), (9, 1295872129, 1440869120, 37748736),
            (9, 2574967041, 2862863872, 75497472)]
    same = 0
    for odd, lo, hi, n in rows:
        if odd == 3:
            assert auto(lo) == (3, n), (lo, auto(lo))                       # same plan as the reference where the size admits p
            assert auto(hi) == (1, n // 3 * 4), (hi, auto(hi))              # beyond our capacity of that size: the next stock size
            same += 1
        else:
            assert auto(lo) == auto(hi) == (3, n // 9 * 12), (lo, hi, auto(lo))   # radix-9 row: the radix-3 size above it
    assert same == 7
    assert auto(175000001) == (3, 6291456)   # the reference takes pfa9 at 4,718,592 words here (ADVICE r03's example)
