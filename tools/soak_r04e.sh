#!/bin/bash
# round 4 (second session), final tree: long soak at the headline size, PRP with Gerbicz-Li checks (7 M squarings)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 1120 python tools/soak.py 136279841 7000000 4 2>&1 | grep -v "Check passed" | tail -3 | tee $O/soak_r04e.txt
