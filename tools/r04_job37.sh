#!/bin/bash
# round 4 (second session): the radix-5 column kernels keep their per-thread half / double (the table form put the back sweep of C4 into its slow placement mode: 58 -> 74 us);
# parity of the 5 2^k shapes on the mixed build, C4 and C3 bench lines
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz_plans.py tests/test_gpu_fused_ops.py -m gpu -x -q -k "400063 or 800283 or 1600589 or 3200123 or 205271257 or 100000007 or random_exponent or fused" > $O/job37_pytest.log 2>&1; rc=$?; tail -3 $O/job37_pytest.log; [ $rc -ne 0 ] && exit $rc
for p in 205271257 136279841; do python bench.py --exponent $p --no-cpu-baseline --steps 1500 --warmup 100 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print($p, d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"; done | tee $O/job37_bench.txt
