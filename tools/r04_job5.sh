#!/bin/bash
# round 4: grouped products (one rare-case test per group of eight) + three more lazy sums per radix-8, against the round-3 library (HEAD), same box
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
python -c "
import sys; sys.path.insert(0,'.')
from prmers_amd.engine import load_library
L=load_library(); print('selftest', L.mi355_engine_selftest(0), L.mi355_engine_last_error())" 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_ops.py -x -q > $O/job5_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/job5_tests.log
tools/ab_many.sh "136279841" prmers_amd/libmi355_engine_head.so prmers_amd/libmi355_engine.so 2>&1 | grep -v amdgpu.ids > $O/job5_ab_grouped_c3.txt; cat $O/job5_ab_grouped_c3.txt
for L in prmers_amd/libmi355_engine_head.so prmers_amd/libmi355_engine.so prmers_amd/libmi355_engine_head.so prmers_amd/libmi355_engine.so; do for p in 205271257 57885161 250000013 30402457; do
  MI355_ENGINE_LIB=$L python bench.py --exponent $p --no-cpu-baseline --steps 1000 --warmup 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('$L'.split('/')[-1], $p, d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"
done; done > $O/job5_ab_grouped_sizes.txt 2>&1; cat $O/job5_ab_grouped_sizes.txt
