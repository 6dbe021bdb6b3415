#!/bin/bash
# round 4: with rows of 2048 one plane per thread -- other plans at n = 2^21 / 2^22 (1024 / 2048 rows of 2048), then the whole GPU suite
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
run() { MI355_TUNE=$1 python bench.py --exponent $2 ${3:+--plan $3} --no-cpu-baseline --steps 3000 --warmup 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('tune=$1', $2, d['config']['plan'], d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"; }
for rep in 1 2; do run 0 30402457; run 0 30402457 m2=2048,c=4; run 16384 30402457 m2=2048,c=4; run 0 57885161; run 0 57885161 m2=2048,c=2; run 16384 57885161 m2=2048,c=2; run 0 100000007; run 0 19000013; done > $O/job17_plans.txt 2>&1; cat $O/job17_plans.txt
python -m pytest tests -m gpu -x -q > $O/job17_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/job17_pytest.txt
