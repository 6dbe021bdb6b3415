#!/bin/bash
# round 4 (second session): radix-5 columns in prime-factor form: parity of every 5 2^k case, then same-box A/B against the mixed-radix build
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_ops.py tests/test_gpu_runs.py tests/test_gpu_switches.py tests/test_gpu_fuzz_plans.py tests/test_gpu_canon.py tests/test_prp_driver.py -m gpu -x -q -k "400063 or 800283 or 1600589 or 3200123 or 205271257 or 100000007 or 50000017 or 332000003 or 700000001 or random_exponent or m859433 or 216091 or largest" > $O/job26_pytest.log 2>&1; rc=$?; tail -6 $O/job26_pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 700 bash tools/ab_many.sh "205271257 100000007 332000003" prmers_amd/libmi355_engine_base.so prmers_amd/libmi355_engine.so 2>&1 | grep -v amdgpu.ids > $O/job26_ab_pfa.txt
cat $O/job26_ab_pfa.txt
