#!/bin/bash
# round 4 (second session): the whole GPU suite on the tree with the 2560 x 2 columns and the switches test
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=15 > $O/job19_pytest.log 2>&1; rc=$?; tail -25 $O/job19_pytest.log; exit $rc
