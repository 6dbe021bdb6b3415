#!/bin/bash
# round 4 (second session), final tree: whole GPU suite, smoke, default and driver-style bench lines
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/job33_pytest.log 2>&1; rc=$?; tail -10 $O/job33_pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/job33_bench_default.json 2> $O/job33_bench.err; echo "bench default rc=$?"
python bench.py --steps 20 --warmup 5 > $O/job33_bench_driver.json 2>> $O/job33_bench.err; echo "bench driver rc=$?"
python - <<'PY'
import json
for f in ("job33_bench_default.json","job33_bench_driver.json"):
    d=json.loads(open("gpurun_out/r04b/"+f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["iteration"]["frac"], d["roofline"]["valu"]["frac"], d["cpu_baseline"] and d["cpu_baseline"]["value"])
PY
