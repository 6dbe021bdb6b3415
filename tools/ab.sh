#!/bin/bash
# A/B two builds of the engine library on the SAME GPU box (boxes differ by a few per cent, so numbers from
# different gpurun calls do not compare).  Usage on the box:  tools/ab.sh path/to/libA.so path/to/libB.so [bench args]
# Build variants with e.g.  make -C prmers_amd/csrc OUT=../libmi355_engine_B.so
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for L in "$A" "$B"; do
    MI355_ENGINE_LIB=$L python bench.py --no-cpu-baseline --steps 1500 --warmup 150 "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('$L', d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"
  done
done
