#!/bin/bash
# round 4 (second session), final tree: whole GPU suite, smoke, default and driver-style bench lines, C2 / C4 lines, size ladder
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > $O/job25_pytest.log 2>&1; rc=$?; tail -14 $O/job25_pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $O/job25_bench_default.json 2> $O/job25_bench.err; echo "bench default rc=$?"
python bench.py --steps 20 --warmup 5 > $O/job25_bench_driver.json 2>> $O/job25_bench.err; echo "bench driver rc=$?"
python bench.py --exponent 9815459 --no-cpu-baseline > $O/job25_bench_c2.json 2>> $O/job25_bench.err
python bench.py --exponent 205271257 --no-cpu-baseline > $O/job25_bench_c4.json 2>> $O/job25_bench.err
bash tools/bench_sizes.sh 2976221 4800007 9815459 19000013 30402457 50000017 57885161 100000007 136279841 205271257 250000013 332000003 600000001 700000001 > $O/job25_bench_sizes.txt 2>&1; cat $O/job25_bench_sizes.txt
python - <<'PY'
import json
for f in ("job25_bench_default.json","job25_bench_driver.json","job25_bench_c2.json","job25_bench_c4.json"):
    try:
        d=json.loads(open("gpurun_out/r04b/"+f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["iteration"]["frac"], (d["roofline"].get("valu") or {}).get("frac"), (d["roofline"].get("valu") or {}).get("shader_clock_ghz"))
    except Exception as e: print(f, "ERR", e)
PY
