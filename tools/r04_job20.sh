#!/bin/bash
# round 4: rows of 8192 with the two halves of a work-group on barriers of their own (half_barrier) -- parity, then same-box A/B against
# s_barrier (MI355_TUNE=32768) at 5 2^22, 2^25 and 5 2^23
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "8192 or largest or prp_iterations or ops_random" > $O/job20_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/job20_tests.log
[ $rc -eq 0 ] || exit 1
run() { MI355_TUNE=$1 python bench.py --exponent $2 --no-cpu-baseline --steps 600 --warmup 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('tune=$1', $2, d['config']['plan'], d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"; }
for rep in 1 2 3; do run 0 332000003; run 32768 332000003; run 0 600000001; run 32768 600000001; done > $O/job20_ab_half_barrier.txt 2>&1; cat $O/job20_ab_half_barrier.txt
run 0 700000001; run 32768 700000001
