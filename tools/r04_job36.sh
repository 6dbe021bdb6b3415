#!/bin/bash
# round 4 (second session): halved / doubled weight tables instead of per-thread half / double: whole GPU suite, then same-box A/B at C3 and C4
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/job36_pytest.log 2>&1; rc=$?; tail -3 $O/job36_pytest.log; [ $rc -ne 0 ] && exit $rc
for rep in 1 2; do for L in prmers_amd/libmi355_engine_base.so prmers_amd/libmi355_engine.so; do for p in 136279841 205271257; do
  MI355_ENGINE_LIB=$L python bench.py --exponent $p --no-cpu-baseline --steps 1500 --warmup 100 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('$L'.split('/')[-1], $p, d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"
done; done; done > $O/job36_ab.txt 2>&1
cat $O/job36_ab.txt
