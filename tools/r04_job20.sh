#!/bin/bash
# round 4 (second session): back + front in one launch on the radix-8 column shapes (MI355_CHAIN=1): parity, then same-box A/B at C3, n = 2^22, 2^24
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_chain.py -m gpu -x -q > $O/job20_pytest.log 2>&1; rc=$?; tail -15 $O/job20_pytest.log; [ $rc -ne 0 ] && exit $rc
b() { timeout -k 10 300 python bench.py --exponent $1 ${2:+--plan $2} --no-cpu-baseline --steps 2000 --warmup 100 --preheat-seconds 2 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('chain=%s' % os.environ.get('MI355_CHAIN','0'), $1, d['config']['plan'], d['ms_per_step'], 'll', d.get('ll_ms_per_step'), {a:round(b*1e3,1) for a,b in k.items() if b>0})"; }
{ for r in 1 2 3; do b 136279841; MI355_CHAIN=1 b 136279841; done; for p in 57885161 250000013 30402457; do b $p; MI355_CHAIN=1 b $p; done; } > $O/job20_ab.txt 2>&1
cat $O/job20_ab.txt
