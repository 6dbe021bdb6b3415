#!/bin/bash
# same-box A/B of two engine libraries at C4 (p = 205271257, n = 5 * 2^21) and C3: tools/ab_c4.sh libA.so libB.so
A=$1; B=$2
for rep in 1 2; do for L in "$A" "$B"; do for p in 205271257 136279841; do
  MI355_ENGINE_LIB=$L python bench.py --exponent $p --no-cpu-baseline --steps 1500 --warmup 150 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('$L'.split('/')[-1], $p, d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"
done; done; done
