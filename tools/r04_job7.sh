#!/bin/bash
# round 4: the radix-4 kernel set (kernels_v3.hip) -- parity at its shapes, then same-box A/B against the generic set (MI355_TUNE=128)
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "radix4 or 86243 or 132049 or 300007 or 756839 or 600011 or 1200007 or 2976221 or c2_9815459" > $O/job7_tests_v3.log 2>&1; echo "v3 tests rc=$?"; tail -5 $O/job7_tests_v3.log
for rep in 1 2 3; do for tune in 128 0; do for p in 9815459 4800007 50000017 2976221; do
  MI355_TUNE=$tune python bench.py --exponent $p --no-cpu-baseline --steps 3000 --warmup 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('tune=$tune', $p, d['config']['plan'], d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"
done; done; done > $O/job7_ab_v3.txt 2>&1; cat $O/job7_ab_v3.txt
