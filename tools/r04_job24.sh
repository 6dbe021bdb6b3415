#!/bin/bash
# round 4 (second session): seeded (exponent, plan) sweep and the complete PRP of M859433 on the columns of 2560
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests/test_gpu_fuzz_plans.py tests/test_prp_driver.py -m gpu -q -k "random_exponent or m859433" --durations=5 > $O/job24_pytest.log 2>&1; rc=$?; tail -40 $O/job24_pytest.log; exit $rc
