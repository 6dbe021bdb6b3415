#!/bin/bash
# round 4 (second session): tile order of the 2560 x 2 front sweep (MI355_TUNE bit 5 = XCD-contiguous), same box
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
b() { timeout -k 10 300 python bench.py --exponent $1 ${2:+--plan $2} --no-cpu-baseline --steps 300 --warmup 20 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('tune=%s' % os.environ.get('MI355_TUNE','0'), $1, d['config']['plan'], d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if b>0}, 'frac', d['roofline']['iteration']['frac'])"; }
{ for r in 1 2; do b 332000003; MI355_TUNE=32 b 332000003; MI355_TUNE=1 b 332000003; done; b 700000001; MI355_TUNE=32 b 700000001; b 250000013; MI355_TUNE=32 b 250000013; } > $O/job15_ab.txt 2>&1
cat $O/job15_ab.txt
