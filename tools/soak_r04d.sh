#!/bin/bash
# round 4 (second session) soak of the radix-5 columns in prime-factor form: PRP with Gerbicz-Li checks at n = 5 2^19 .. 5 2^22, LL-safe at 5 2^20, complete PRP of M13466917 is
# on generic kernels (not repeated)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
{ timeout -k 10 200 python tools/soak.py 205271257 500000 4 2>&1 | grep -v "Check passed" | tail -2
  timeout -k 10 200 python tools/soak.py 100000007 800000 4 2>&1 | grep -v "Check passed" | tail -2
  timeout -k 10 200 python tools/soak.py 50000017 1000000 4 2>&1 | grep -v "Check passed" | tail -2
  timeout -k 10 200 python tools/soak.py 332000003 200000 4 2>&1 | grep -v "Check passed" | tail -2
  timeout -k 10 250 python tools/soak_llsafe2.py 100000007 400000 2>&1 | tail -2; } | tee $O/soak_r04d.txt
