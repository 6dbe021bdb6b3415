#!/bin/bash
# round 4: radix-4 set policy -- plans at n = 2^20 and 5 2^19, the rows forced (MI355_TUNE=256) against the default
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
run() { MI355_TUNE=$1 python bench.py --exponent $2 --no-cpu-baseline --steps 3000 --warmup 300 ${3:+--plan $3} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('tune=$1', $2, d['config']['plan'], d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"; }
for rep in 1 2; do
  run 0 9815459; run 128 9815459; run 256 9815459
  run 0 19000013; run 0 19000013 "m2=1024,c=4"; run 128 19000013 "m2=1024,c=4"; run 0 19000013 "m2=2048,c=4"; run 128 19000013 "m2=2048,c=4"; run 0 19000013 "m2=4096,c=4"
  run 0 50000017; run 128 50000017
  run 0 4800007; run 256 4800007
done > $O/job8_v3_policy.txt 2>&1; cat $O/job8_v3_policy.txt
