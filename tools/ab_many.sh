#!/bin/bash
# same-box A/B of several engine libraries: tools/ab_many.sh "<exponents>" libA.so libB.so ...   (see tools/ab.sh)
PS=$1; shift
for rep in 1 2 3; do for L in "$@"; do for p in $PS; do
  MI355_ENGINE_LIB=$L python bench.py --exponent $p --no-cpu-baseline --steps 1500 --warmup 150 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('$L'.split('/')[-1], $p, d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"
done; done; done
