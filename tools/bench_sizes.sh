#!/bin/bash
# ms per squaring over the size ladder (one line per exponent): tools/bench_sizes.sh [exponents...]
PS=${@:-"2976221 9815459 30402457 57885161 136279841 205271257 250000013 332000003 600000001 800000011 1300000003 1800000011 4000000007"}
for p in $PS; do
  steps=1000; [ $p -gt 700000000 ] && steps=60
  python bench.py --exponent $p --no-cpu-baseline --steps $steps --warmup 20 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print($p, d['config']['plan'], d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if b>0}, 'frac', d['roofline']['iteration']['frac'])"
done
