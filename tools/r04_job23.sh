#!/bin/bash
# round 4 (second session): the experimental library's checks (cooperative launch, both chain kernels) on the final tree
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
MI355_ENGINE_LIB=prmers_amd/libmi355_engine_exp.so timeout -k 10 600 python tools/exp_coop_check.py > $O/job23_exp_check.txt 2>&1; rc=$?; tail -25 $O/job23_exp_check.txt; exit $rc
