#!/bin/bash
# round 4 (second session): rocprofv3 profile of C4 on the prime-factor radix-5 columns, bench lines of the final tree
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 bash tools/profile.sh r04b_c4 --exponent 205271257 > $O/job29_profile_c4.log 2>&1; echo "profile c4 rc=$?"
cd $GRAFT_REPO_ROOT
python bench.py --exponent 205271257 --no-cpu-baseline > $O/job29_bench_c4.json 2>> $O/job29_bench.err
python bench.py > $O/job29_bench_default.json 2>> $O/job29_bench.err
python bench.py --steps 20 --warmup 5 > $O/job29_bench_driver.json 2>> $O/job29_bench.err
bash tools/bench_sizes.sh 50000017 100000007 205271257 332000003 700000001 > $O/job29_bench_sizes.txt 2>&1; cat $O/job29_bench_sizes.txt
python - <<'PY'
import json
for f in ("job29_bench_default.json","job29_bench_driver.json","job29_bench_c4.json"):
    try:
        d=json.loads(open("gpurun_out/r04b/"+f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["iteration"]["frac"])
    except Exception as e: print(f, "ERR", e)
s=json.load(open("gpurun_out/r04b_c4/summary.json"))
for r in s["kernel_stats"][:3]: print(r["name"], r["avg_ns"])
for k,v in s["sq"].items():
    if "cols5" in k: print(k, v["SQ_INSTS_VALU"], v["SQ_LDS_BANK_CONFLICT"], v["SQ_LDS_IDX_ACTIVE"])
PY
