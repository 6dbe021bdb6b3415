#!/bin/bash
# round 4: other plans now that rows of 1024 / 2048 have plane-per-thread kernels (n = 2^21, 2^22, C3), and the rocprofv3 profile
# (kernel trace + PMC passes) of n = 2^20 on the final kernels
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { MI355_TUNE=$1 python bench.py --exponent $2 ${3:+--plan $3} --no-cpu-baseline --steps 3000 --warmup 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('tune=$1', $2, d['config']['plan'], d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"; }
for rep in 1 2; do run 0 30402457; run 0 30402457 m2=1024,c=4; run 0 57885161; run 0 57885161 m2=2048,c=4; run 16384 57885161 m2=2048,c=4; run 0 136279841; run 16384 136279841 m2=2048,c=2; done > $O/job19_plans.txt 2>&1; cat $O/job19_plans.txt
bash tools/profile.sh r04_n20 --exponent 19000013 > $O/job19_profile_n20.log 2>&1; echo "profile n20 rc=$?"
