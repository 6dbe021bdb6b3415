#!/bin/bash
# round 4: complete PRP of the Mersenne prime M20996011 (n = 2^20) on the kernels of this round (radix-4 columns of 256 x 4 with one plane
# per thread + rows of 2048 with one plane per thread), Gerbicz-Li check on
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 1150 python tools/full_prp.py 20996011 2>&1 | tee $O/job18_m20996011_prp.txt | tail -4
