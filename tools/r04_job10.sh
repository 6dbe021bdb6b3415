#!/bin/bash
# round 4: per-exchange LDS slot maps in the radix-5 column kernels (kernels_v5.hip) against the old skew: parity, A/B at C4 / 5 2^22 / 5 2^20, counters
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_ops.py -x -q -k "400063 or 800283 or 1600589 or 205271257 or c4 or radix4" > $O/job10_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/job10_tests.log
for rep in 1 2 3; do for L in prmers_amd/libmi355_engine_ldsadd3.so prmers_amd/libmi355_engine.so; do for p in 205271257 332000003 100000007; do
  MI355_ENGINE_LIB=$L python bench.py --exponent $p --no-cpu-baseline --steps 1000 --warmup 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('$L'.split('/')[-1], $p, d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"
done; done; done > $O/job10_ab_v5_lds.txt 2>&1; cat $O/job10_ab_v5_lds.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_c4
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/pmc_c4 -- python3 $GRAFT_REPO_ROOT/bench.py --exponent 205271257 --steps 20 --warmup 5 --no-cpu-baseline --preheat-seconds 0 > /tmp/pmc_c4.log 2>&1
python3 - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r04/job10_c4_lds_counters.txt 2>&1
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
cat $GRAFT_REPO_ROOT/gpurun_out/r04/job10_c4_lds_counters.txt
