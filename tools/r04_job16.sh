#!/bin/bash
# round 4: rows of 2048 with one plane per thread (kernels_v2.hip k2_rows2048_planes) -- parity, then same-box A/B: n = 2^20 against the
# generic rows (MI355_TUNE=8192), n = 2^21 and 5 2^20 forced (16384) against two rows to a tile
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "2048 or radix4_set or prp_iterations or ops_random" > $O/job16_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/job16_tests.log
[ $rc -eq 0 ] || exit 1
run() { MI355_TUNE=$1 python bench.py --exponent $2 --no-cpu-baseline --steps 3000 --warmup 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('tune=$1', $2, d['config']['plan'], d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"; }
for rep in 1 2 3; do run 0 19000013; run 8192 19000013; run 0 30402457; run 16384 30402457; run 0 100000007; run 16384 100000007; done > $O/job16_ab_rows2048_planes.txt 2>&1; cat $O/job16_ab_rows2048_planes.txt
