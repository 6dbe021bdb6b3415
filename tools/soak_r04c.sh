#!/bin/bash
# round 4 (second session) soak of the 2560 x 2 radix-5 columns: PRP with Gerbicz-Li checks at n = 5 2^22 (rows of 4096) and 5 2^23 (rows of 8192), LL-safe at 5 2^22
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
{ timeout -k 10 400 python tools/soak.py 332000003 500000 4 2>&1 | grep -v "Check passed" | tail -3
  timeout -k 10 300 python tools/soak.py 700000001 150000 4 2>&1 | grep -v "Check passed" | tail -3
  timeout -k 10 300 python tools/soak_llsafe2.py 332000003 120000 2>&1 | tail -3; } | tee $O/soak_r04c.txt
