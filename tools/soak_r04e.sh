#!/bin/bash
# round 4 (second session), final tree: long soak at the headline size, PRP with Gerbicz-Li checks (5 M squarings); the raw log keeps the run visibly alive
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tools/soak.py 136279841 5000000 4 2>&1 | tee $O/soak_r04e_raw.log | grep -v "Check passed" | tail -3 | tee $O/soak_r04e.txt
