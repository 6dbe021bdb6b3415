#!/bin/bash
# round 4 (second session), final tree: smoke, default and driver-style bench lines
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/job38_bench_default.json 2> $O/job38_bench.err; echo "bench default rc=$?"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/job38_bench_driver.json 2>> $O/job38_bench.err; echo "bench driver rc=$?"
python - <<'PY'
import json
for f in ("job38_bench_default.json","job38_bench_driver.json"):
    d=json.loads(open("gpurun_out/r04b/"+f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["iteration"]["frac"], d["roofline"]["valu"]["frac"], d["roofline"]["valu"]["insts"])
PY
