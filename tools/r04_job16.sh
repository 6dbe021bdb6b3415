#!/bin/bash
# round 4 (second session): rocprofv3 profile (kernel trace + PMC passes) of n = 5 2^22 on the 2560 x 4096 plan
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 800 bash tools/profile.sh r04b_n5_22 --exponent 332000003 > $O/job16_profile.log 2>&1; echo "profile rc=$?"
cat $GRAFT_REPO_ROOT/gpurun_out/r04b_n5_22/summary.json | head -150
