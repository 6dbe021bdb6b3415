#!/bin/bash
# round 4 (second session): the padding waves of the radix-5 column kernels touch the next tile's lines (MI355_TUNE bit 15), same box
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "800283 or 1600589 or 400063" > $O/job21_pytest.log 2>&1; rc=$?; tail -3 $O/job21_pytest.log; [ $rc -ne 0 ] && exit $rc
b() { timeout -k 10 300 python bench.py --exponent $1 ${2:+--plan $2} --no-cpu-baseline --steps 300 --warmup 20 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('tune=%s' % os.environ.get('MI355_TUNE','0'), $1, d['config']['plan'], d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if b>0})"; }
{ for r in 1 2 3; do b 332000003; MI355_TUNE=32768 b 332000003; done; for r in 1 2; do b 700000001; MI355_TUNE=32768 b 700000001; b 205271257; MI355_TUNE=32768 b 205271257; done; } > $O/job21_ab.txt 2>&1
cat $O/job21_ab.txt
