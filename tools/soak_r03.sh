#!/bin/bash
# round-3 soak of the final tree (one GPU box call): Gerbicz-Li checked squarings at the shapes whose kernels changed this round
# (radix-5 columns launched with twelve waves, LDS-only exchange barriers, the split sweeps, the crt family's device-side compare)
set -e
O=gpurun_out/r03; mkdir -p $O
python tools/soak.py 205271257 300000 4 2>&1 | grep -v "Check passed" | tail -3        # C4: 5 * 2^21, radix-5 register-resident columns
python tools/soak.py 136279841 250000 4 2>&1 | grep -v "Check passed" | tail -3        # C3
python tools/soak.py 332000003 60000 1 2>&1 | grep -v "Check passed" | tail -3         # 5 * 2^22: radix-5 columns + rows of 8192
python tools/soak.py 9815459 400000 8 2>&1 | grep -v "Check passed" | tail -3          # C2
python tools/soak_crt.py 1257787 9 4 2>&1 | grep -v "Check passed" | tail -3      # complete PRP of M1257787 on the crt family, radix 9
python tools/full_prp.py 6972593 2>&1 | tail -3                                                 # complete PRP of M6972593
MI355_ENGINE_LIB=prmers_amd/libmi355_engine_exp.so MI355_COOP=1 python tools/soak.py 9815459 200000 8 2>&1 | grep -v "Check passed" | tail -3   # C2 through the one-launch kernel (opt-in path): runs of squarings per cooperative launch
python tools/soak_crt.py 756839 3 4 2>&1 | grep -v "Check passed" | tail -3          # complete PRP of M756839 on the crt family, radix 3 (fused back + carry)
