#!/bin/bash
# round-4 soak, second part (final tree): the shapes of the plane-per-thread rows of 2048 -- n = 5 2^20 PRP with Gerbicz-Li checks, LL-safe
# (multiplicand images and multiplications through the same row kernel) at n = 2^20 and 5 2^20 -- and the complete PRP of the Mersenne prime
# M13466917 (n = 2^20)
set -e
O=gpurun_out/r04; mkdir -p $O
python tools/soak.py 100000007 1000000 4 2>&1 | tee -a $O/soakb_raw.log | grep -v "Check passed" | tail -3
python tools/soak_llsafe2.py 19000013 600000 2>&1 | tee -a $O/soakb_raw.log | grep -v "passed" | tail -3
python tools/soak_llsafe2.py 100000007 200000 2>&1 | tee -a $O/soakb_raw.log | grep -v "passed" | tail -3
python tools/full_prp.py 13466917 2>&1 | tee -a $O/soakb_raw.log | tail -2
