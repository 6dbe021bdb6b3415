#!/bin/bash
# round 4 (second session): tile order of the 1280 x 4 front sweep again on the prime-factor columns (MI355_TUNE=32: XCD-contiguous), boost share at C4
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
b() { timeout -k 10 200 python bench.py --exponent $1 --no-cpu-baseline --steps 1500 --warmup 100 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('tune=%s boost=%s' % (os.environ.get('MI355_TUNE','0'), os.environ.get('MI355_BOOST','50')), $1, d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if b>0})"; }
{ for r in 1 2; do b 205271257; MI355_TUNE=32 b 205271257; MI355_BOOST=75 b 205271257; MI355_BOOST=25 b 205271257; done; } > $O/job35_ab.txt 2>&1
cat $O/job35_ab.txt
