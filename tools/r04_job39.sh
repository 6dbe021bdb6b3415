#!/bin/bash
# round 4 (second session): boosted share of the last round re-checked at C3 on the final kernels (MI355_BOOST in percent; default 50)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
b() { timeout -k 10 100 python bench.py --no-cpu-baseline --steps 2000 --warmup 100 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('boost=%s' % os.environ.get('MI355_BOOST','50'), d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if b>0})"; }
{ for r in 1 2; do b; MI355_BOOST=40 b; MI355_BOOST=60 b; MI355_BOOST=35 b; done; } > $O/job39_boost.txt 2>&1
cat $O/job39_boost.txt
