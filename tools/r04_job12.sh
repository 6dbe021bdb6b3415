#!/bin/bash
# round 4: plane-per-thread column kernels of the radix-4 set -- parity, A/B against the pair form (MI355_TUNE=1024) at C2 and n = 2^20
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_runs.py tests/test_gpu_fused_ops.py tests/test_prp_driver.py -x -q -m gpu > $O/job12_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/job12_tests.log
run() { MI355_TUNE=$1 python bench.py --exponent $2 --no-cpu-baseline --steps 3000 --warmup 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('tune=$1', $2, d['config']['plan'], d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"; }
for rep in 1 2 3; do run 0 9815459; run 1024 9815459; run 128 9815459; run 0 19000013; run 1024 19000013; run 2048 19000013; done > $O/job12_ab_col_planes.txt 2>&1; cat $O/job12_ab_col_planes.txt
