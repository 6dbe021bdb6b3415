#!/bin/bash
# same-box A/B of the small-tile paths of the generic kernels: MI355_TUNE=8 is the round-1 form (a radix-4 butterfly of both planes per thread)
for p in 2976221 9815459 20996011 30402457; do for t in 8 0; do
  MI355_TUNE=$t python bench.py --exponent $p --steps 2000 --warmup 200 --preheat-seconds 0.5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('tune=$t', d['config']['plan'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
