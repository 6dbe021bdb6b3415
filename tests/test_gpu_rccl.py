"""One RCCL call on hardware: the launcher's status-word path over backend "nccl" (= RCCL) at world size 1, in a fresh
child process in which the process group is the first thing that touches the GPU -- so the collectives of the 8-GPU run of BASELINE
configs[4] (all_reduce of {ok MIN, errors SUM, iterations SUM} at every Gerbicz-Li boundary and at exit, the gather of the
per-exponent results) have run over RCCL on an MI355X before an 8-GPU node runs them.  Needs a real MI355X."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_status_word_over_rccl_in_a_one_rank_group():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_entry.py"), str(port)], capture_output=True, text=True,
                         env=env, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    if "rccl_unavailable" in line:
        pytest.skip("RCCL refused a one-rank communicator on this box: " + line["rccl_unavailable"])
    assert line["backend"] == "nccl"
    st = line["status"]
    # M9941 is prime, M9949 is not (unit_tests.sh:5-14); 100 + 100 boundary checks with checklevel 1, none failing
    assert st["all_ok"] == 1 and st["gerbicz_errors"] == 0 and st["iterations"] == 9941 + 9949, st
    assert st["check_boundary_reductions"] is True and line["reductions"] > 100, line
    by_p = {r["exponent"]: r for r in line["results"]}
    assert by_p[9941]["is_prime"] and by_p[9941]["res64"] == "0000000000000001" and by_p[9941]["complete"]
    assert not by_p[9949]["is_prime"] and by_p[9949]["complete"]
