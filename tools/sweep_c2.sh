#!/bin/bash
# plan sweep for a small exponent on the GPU box: tools/sweep_c2.sh <exponent>
P=${1:-9815459}
for m2 in 256 512 1024 2048; do for c in 1 2 4 8; do
  python bench.py --exponent $P --plan "m2=$m2,c=$c" --steps 2000 --warmup 200 --preheat-seconds 0.5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['config']['plan'], d['ms_per_step'], d['roofline']['kernel_ms'])" || echo "m2=$m2 c=$c failed"
done; done
