#!/bin/bash
# round 4, first box call: the whole GPU suite on the tree after the hygiene batch, then the headline bench lines and a plan sweep at C3
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/job1_tests.log 2>&1; echo "tests rc=$?" | tee $O/job1_tests.rc
tail -3 $O/job1_tests.log
python bench.py > $O/job1_bench_default.json 2> $O/job1_bench_default.err && tail -c 600 $O/job1_bench_default.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/job1_bench_driver.json 2>> $O/job1_bench_default.err
for plan in "m2=2048,c=2" "m2=8192,c=8"; do
  python bench.py --no-cpu-baseline --plan "$plan" > "$O/job1_bench_plan_${plan//[=,]/_}.json" 2>> $O/job1_bench_default.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/job1_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["ms_per_step"], d["roofline"]["kernel_ms"], (d["roofline"].get("valu") or {}).get("shader_clock_ghz"), (d["roofline"].get("valu") or {}).get("shader_clock_source"))
    except Exception as e:
        print(f, "ERR", e)
PY
