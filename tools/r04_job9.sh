#!/bin/bash
# round 4: whole GPU suite on the tree with the radix-4 set and the new LDS map, the size ladder, and the LDS counters of the C4 kernels
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/job9_tests.log 2>&1; echo "tests rc=$?" | tee $O/job9_tests.rc; tail -3 $O/job9_tests.log
bash tools/bench_sizes.sh > $O/job9_bench_sizes.txt 2>&1; cat $O/job9_bench_sizes.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_c4
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/pmc_c4 -- python3 $GRAFT_REPO_ROOT/bench.py --exponent 205271257 --steps 20 --warmup 5 --no-cpu-baseline --preheat-seconds 0 > /tmp/pmc_c4.log 2>&1
python3 - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r04/job9_c4_lds_counters.txt 2>&1
import csv, glob, collections
f = glob.glob("/tmp/pmc_c4/*/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"].split("(")[0][-40:]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_INSTS_VALU": cnt[k] += 1
for k in acc:
    if cnt[k] >= 20: print(k, {c: round(v / cnt[k]) for c, v in acc[k].items()}, "launches", cnt[k])
PY
cat $GRAFT_REPO_ROOT/gpurun_out/r04/job9_c4_lds_counters.txt
