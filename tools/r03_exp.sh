#!/bin/bash
# round-3 experiment driver (run on the GPU box): quick parity, bench A/B over MI355_TUNE values, optional probe
# usage: tools/r03_exp.sh <tag> "<tune values>" [probe]
TAG=$1; TUNES=$2; PROBE=$3
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "square or c3 or identit or edge or gmp" > $O/${TAG}_tests.log 2>&1
tail -3 $O/${TAG}_tests.log
for t in $TUNES; do
  MI355_TUNE=$t python bench.py --no-cpu-baseline --steps 1500 --warmup 150 > $O/${TAG}_bench_tune$t.json 2> $O/${TAG}_bench_tune$t.err
  python - <<PY
import json
try:
    d = json.load(open("$O/${TAG}_bench_tune$t.json"))
    print("tune=$t", d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["roofline"]["kernel_ms"].items() if v and v > 0})
except Exception as e:
    print("tune=$t failed", e)
PY
done
if [ -n "$PROBE" ]; then
  timeout -k 10 400 python tools/probe.py $TAG $PROBE > $O/${TAG}_probe.log 2>&1
  grep -h "^rows_\|^front_\|^back_" $O/${TAG}_probe.log | cut -c1-330
  tail -2 $O/${TAG}_probe.log | cut -c1-300
fi
