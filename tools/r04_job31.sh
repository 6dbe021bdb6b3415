#!/bin/bash
# round 4 (second session): generic radix-5 stage in prime-factor form: parity (every test file that builds Goldilocks engines), then same-box A/B at generic 5 2^k sizes
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz_plans.py -m gpu -x -q -k "prp_iterations or random_exponent" > $O/job31_pytest_a.log 2>&1; rc=$?; tail -3 $O/job31_pytest_a.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_ops.py tests/test_gpu_runs.py tests/test_gpu_switches.py tests/test_gpu_fuzz_plans.py tests/test_gpu_canon.py tests/test_prp_driver.py tests/test_host_logic.py -m gpu -x -q > $O/job31_pytest.log 2>&1; rc=$?; tail -5 $O/job31_pytest.log; [ $rc -ne 0 ] && exit $rc
for rep in 1 2; do for L in prmers_amd/libmi355_engine_base.so prmers_amd/libmi355_engine.so; do for p in 13466917 25000009 1300000003; do
  steps=1500; [ $p -gt 700000000 ] && steps=60
  MI355_ENGINE_LIB=$L python bench.py --exponent $p --no-cpu-baseline --steps $steps --warmup 20 --preheat-seconds 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('$L'.split('/')[-1], $p, d['config']['plan'], d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"
done; done; done > $O/job31_ab.txt 2>&1
cat $O/job31_ab.txt
# the residues of earlier rounds' soaks (mixed-radix columns, other plans) must come out again: res64 after the same number of squarings
{ timeout -k 10 200 python tools/soak.py 205271257 300000 4 2>&1 | tail -1; timeout -k 10 200 python tools/soak.py 50000017 600000 4 2>&1 | tail -1; timeout -k 10 200 python tools/soak.py 332000003 60000 1 2>&1 | tail -1; } | tee $O/job31_res64.txt
grep -c "72233E6BB35EA90E\|CC34B67954C83DAB\|EC202BAAFD1B267F" $O/job31_res64.txt
