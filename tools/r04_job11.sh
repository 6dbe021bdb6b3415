#!/bin/bash
# round 4: plane-per-thread rows of 1024 -- parity, A/B against the pair form (MI355_TUNE=256) and the generic rows (128); then the round's soak
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_runs.py -x -q -k "radix4 or 86243 or 132049 or 300007 or 756839 or 600011 or 1200007 or 2976221 or c2_9815459 or 9815459 or 4800007" > $O/job11_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/job11_tests.log
run() { MI355_TUNE=$1 python bench.py --exponent $2 --no-cpu-baseline --steps 3000 --warmup 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('tune=$1', $2, d['config']['plan'], d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"; }
for rep in 1 2 3; do run 0 9815459; run 256 9815459; run 128 9815459; run 0 4800007; run 128 4800007; run 0 50000017; run 512 50000017; done > $O/job11_ab_planes.txt 2>&1; cat $O/job11_ab_planes.txt
grep -q "rc=0" <<< "$(tail -1 $O/job11_tests.log; echo rc=$?)" || true
bash tools/soak_r04.sh > $O/soak.txt 2>&1; echo "soak rc=$?"; cat $O/soak.txt
