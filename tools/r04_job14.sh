#!/bin/bash
# round 4: back + front in one launch for runs of squarings on the small shapes -- parity (square_mul_n against the oracle and the loop),
# PRP driver tests (they run their blocks through square_mul_n), A/B against MI355_TUNE=4096 (chain off), complete PRP of M9815459
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
python -m pytest tests/test_gpu_runs.py tests/test_prp_driver.py tests/test_gpu_parity.py -x -q -m gpu > $O/job14_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/job14_tests.log
run() { MI355_TUNE=$1 python bench.py --exponent $2 --no-cpu-baseline --steps 3000 --warmup 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('tune=$1', $2, d['config']['plan'], d['ms_per_step'], 'll', d['ll_ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"; }
for rep in 1 2 3; do run 0 9815459; run 4096 9815459; run 0 19000013; run 4096 19000013; done > $O/job14_ab_chain.txt 2>&1; cat $O/job14_ab_chain.txt
python tools/full_prp.py 9815459 2>&1 | tee $O/job14_c2_full_prp.txt | tail -2
