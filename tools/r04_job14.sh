#!/bin/bash
# round 4 (second session): columns of 2560 = 5 x 512 with runs of two pairs on the radix-5 kernels -- parity cases, then same-box A/B at
# n = 5 2^22 (new 2560 x 4096 against the old 1280 x 8192) and n = 5 2^23 (register-resident against generic columns)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_ops.py tests/test_gpu_runs.py -m gpu -x -q -k "800283 or 1600589 or 3200123 or largest_supported or 400063" > $O/job14_pytest.log 2>&1; rc=$?; tail -5 $O/job14_pytest.log; [ $rc -ne 0 ] && exit $rc
b() { timeout -k 10 300 python bench.py --exponent $1 ${2:+--plan $2} --no-cpu-baseline --steps 300 --warmup 20 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print($1, d['config']['plan'], d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if b>0}, 'frac', d['roofline']['iteration']['frac'])"; }
{ for r in 1 2; do b 332000003; b 332000003 m2=8192; done; for r in 1 2; do b 700000001; MI355_KERNELS=v2rows b 700000001; done; b 205271257; } > $O/job14_ab.txt 2>&1
cat $O/job14_ab.txt
