#!/bin/bash
# round-4 soak of the final tree (one GPU box call, ~15 min): Gerbicz-Li checked squarings at the shapes whose kernels changed this round --
# the radix-4 set (C2 complete, n = 2^20, rows of 1024 under radix-5 columns), the new LDS slot maps of the radix-8 and radix-5 sets (C3, C4,
# 2^22, 2^24, 5 2^22), lazy sums.  Progress lines every check keep the call alive.
set -e
O=gpurun_out/r04; mkdir -p $O
python tools/full_prp.py 9815459 2>&1 | tee -a $O/soak_raw.log | tail -3                                         # the COMPLETE PRP of the C2 exponent (9.8 M squarings, radix-4 columns)
python tools/soak.py 19000013 1500000 8 2>&1 | tee -a $O/soak_raw.log | grep -v "Check passed" | tail -3        # n = 2^20: radix-4 columns with C = 4 (new default plan)
python tools/soak.py 50000017 600000 4 2>&1 | tee -a $O/soak_raw.log | grep -v "Check passed" | tail -3         # n = 5 2^19: radix-4 rows under radix-5 columns
python tools/soak.py 136279841 400000 4 2>&1 | tee -a $O/soak_raw.log | grep -v "Check passed" | tail -3        # C3
python tools/soak.py 205271257 300000 4 2>&1 | tee -a $O/soak_raw.log | grep -v "Check passed" | tail -3        # C4
python tools/soak.py 57885161 400000 4 2>&1 | tee -a $O/soak_raw.log | grep -v "Check passed" | tail -3         # n = 2^22 (columns 512 x 8)
python tools/soak.py 250000013 150000 2 2>&1 | tee -a $O/soak_raw.log | grep -v "Check passed" | tail -3        # n = 2^24 (columns 2048 x 2)
python tools/soak.py 332000003 60000 1 2>&1 | tee -a $O/soak_raw.log | grep -v "Check passed" | tail -3         # 5 2^22: radix-5 columns + rows of 8192
python tools/soak.py 30402457 500000 4 2>&1 | tee -a $O/soak_raw.log | grep -v "Check passed" | tail -3         # n = 2^21: two rows of 2048 to a tile
