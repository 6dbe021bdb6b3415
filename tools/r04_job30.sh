#!/bin/bash
# round 4 (second session): plan alternatives at n = 2^22 (p ~ 58-75 M) and n = 2^21
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
b() { timeout -k 10 300 python bench.py --exponent $1 ${2:+--plan $2} --no-cpu-baseline --steps 1500 --warmup 100 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('tune=%s' % os.environ.get('MI355_TUNE','0'), $1, d['config']['plan'], d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if b>0})"; }
{ for r in 1 2; do b 57885161; b 57885161 m2=2048; MI355_TUNE=16384 b 57885161 m2=2048; b 57885161 m2=2048,c=8; b 57885161 m2=8192; done; b 30402457; b 30402457 m2=1024; b 30402457 m2=4096; } > $O/job30_plans.txt 2>&1
cat $O/job30_plans.txt
