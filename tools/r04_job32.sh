#!/bin/bash
# round 4 (second session): complete PRPs of Mersenne primes on the generic radix-5 stage in prime-factor form: M3021377 (n = 5 2^15), M6972593 (n = 5 2^16)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
{ timeout -k 10 300 python tools/full_prp.py 3021377 2>&1 | tail -2; timeout -k 10 600 python tools/full_prp.py 6972593 2>&1 | tail -2; } | tee $O/job32_full_prp.txt
