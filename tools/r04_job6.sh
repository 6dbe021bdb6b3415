#!/bin/bash
# round 4: LDS slot map i ^ ((i >> 3) & 15) against the old skew i + i / 8 (libmi355_engine_ldsadd3.so): parity, same-box A/B, and the
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE counters of both builds
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_ops.py -x -q > $O/job6_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/job6_tests.log
tools/ab_many.sh "136279841" prmers_amd/libmi355_engine_ldsadd3.so prmers_amd/libmi355_engine.so 2>&1 | grep -v amdgpu.ids > $O/job6_ab_lds_c3.txt; cat $O/job6_ab_lds_c3.txt
for L in prmers_amd/libmi355_engine_ldsadd3.so prmers_amd/libmi355_engine.so prmers_amd/libmi355_engine_ldsadd3.so prmers_amd/libmi355_engine.so; do for p in 205271257 57885161 250000013 30402457 600000001; do
  MI355_ENGINE_LIB=$L python bench.py --exponent $p --no-cpu-baseline --steps 1000 --warmup 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('$L'.split('/')[-1], $p, d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"
done; done > $O/job6_ab_lds_sizes.txt 2>&1; cat $O/job6_ab_lds_sizes.txt
cd /tmp && export TMPDIR=/tmp
for L in ldsadd3 new; do
  LIB=$GRAFT_REPO_ROOT/prmers_amd/libmi355_engine.so; [ $L = ldsadd3 ] && LIB=$GRAFT_REPO_ROOT/prmers_amd/libmi355_engine_ldsadd3.so
  export MI355_ENGINE_LIB=$LIB
  rm -rf /tmp/pmc_$L
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/pmc_$L -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --preheat-seconds 0 > /tmp/pmc_$L.log 2>&1
  python3 - "$L" <<'PY'
import csv, glob, sys, collections
L = sys.argv[1]
f = glob.glob("/tmp/pmc_%s/*/*counter_collection.csv" % L)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"].split("(")[0][-40:]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_INSTS_VALU": cnt[k] += 1
for k in acc:
    if cnt[k] >= 20:
        print(L, k, {c: round(v / cnt[k]) for c, v in acc[k].items()}, "launches", cnt[k])
PY
done > $GRAFT_REPO_ROOT/$O/job6_lds_counters.txt 2>&1; cat $GRAFT_REPO_ROOT/$O/job6_lds_counters.txt
