#!/bin/bash
# round 4, final tree: rocprofv3 profile of the headline command (kernel trace + PMC passes), the same for C2, the default and driver-style
# bench lines, the size ladder, and the complete PRP of the C2 exponent on the final kernels
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $O
cd $GRAFT_REPO_ROOT
bash tools/profile.sh r04_final > $O/job13_profile_c3.log 2>&1; echo "profile c3 rc=$?"
bash tools/profile.sh r04_c2 --exponent 9815459 > $O/job13_profile_c2.log 2>&1; echo "profile c2 rc=$?"
cd $GRAFT_REPO_ROOT
python bench.py > $O/job13_bench_default.json 2> $O/job13_bench.err; echo "bench default rc=$?"
python bench.py --steps 20 --warmup 5 > $O/job13_bench_driver.json 2>> $O/job13_bench.err; echo "bench driver rc=$?"
python bench.py --exponent 9815459 --no-cpu-baseline > $O/job13_bench_c2.json 2>> $O/job13_bench.err
python bench.py --exponent 205271257 --no-cpu-baseline > $O/job13_bench_c4.json 2>> $O/job13_bench.err
bash tools/bench_sizes.sh 2976221 4800007 9815459 19000013 30402457 50000017 57885161 100000007 136279841 205271257 250000013 332000003 600000001 > $O/job13_bench_sizes.txt 2>&1; cat $O/job13_bench_sizes.txt
python tools/full_prp.py 9815459 2>&1 | tee $O/job13_c2_full_prp.txt | tail -2
python - <<'PY'
import json
for f in ("job13_bench_default.json","job13_bench_driver.json","job13_bench_c2.json","job13_bench_c4.json"):
    try:
        d=json.loads(open("gpurun_out/r04/"+f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["iteration"]["frac"], (d["roofline"].get("valu") or {}).get("frac"), (d["roofline"].get("valu") or {}).get("shader_clock_ghz"))
    except Exception as e: print(f, "ERR", e)
PY
