#!/bin/bash
# round 4 (second session): work-buffer layout microbenchmark (HBM-resident column / row access shapes) and the 2560 x 2 column kernels at a
# size that fits the Infinity Cache (n = 5 2^21 forced to 2560 x 2048) next to the 1280 x 4 ones
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 300 tools/microbench_wlayout > $O/job17_wlayout.txt 2>&1; echo "wlayout rc=$?"; cat $O/job17_wlayout.txt
b() { timeout -k 10 300 python bench.py --exponent $1 ${2:+--plan $2} --no-cpu-baseline --steps 300 --warmup 20 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('tune=%s' % os.environ.get('MI355_TUNE','0'), $1, d['config']['plan'], d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if b>0}, 'frac', d['roofline']['iteration']['frac'])"; }
{ b 205271257; b 205271257 m2=2048,c=2; b 205271257 m2=2048,c=4; b 100000007; b 100000007 m2=1024,c=2; b 332000003; } > $O/job17_ab.txt 2>&1
cat $O/job17_ab.txt
