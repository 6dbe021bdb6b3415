#!/bin/bash
# round 4, second box call: numbers behind the dropped butterfly forms, same-box A/B of the round-4 field forms (x 2^48 through the product's
# reduction, three more lazy sums per radix-8) against the round-3 forms, parity of the new library, the experimental cooperative kernel
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
./tools/microbench_alt_butterfly > $O/job2_microbench_alt_butterfly.txt 2>&1; echo "alt_butterfly rc=$?"; cat $O/job2_microbench_alt_butterfly.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_runs.py tests/test_gpu_fused_ops.py -x -q > $O/job2_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/job2_tests.log
tools/ab_many.sh "136279841" prmers_amd/libmi355_engine_r3forms.so prmers_amd/libmi355_engine.so > $O/job2_ab_forms_c3.txt 2>&1; cat $O/job2_ab_forms_c3.txt
PS="205271257 57885161 250000013 30402457"
for L in prmers_amd/libmi355_engine_r3forms.so prmers_amd/libmi355_engine.so prmers_amd/libmi355_engine_r3forms.so prmers_amd/libmi355_engine.so; do for p in $PS; do
  MI355_ENGINE_LIB=$L python bench.py --exponent $p --no-cpu-baseline --steps 1000 --warmup 100 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('$L'.split('/')[-1], $p, d['ms_per_step'], {a:round(b*1e3,1) for a,b in k.items() if a in ('k_front','k_middle','k_back')})"
done; done > $O/job2_ab_forms_sizes.txt 2>&1; cat $O/job2_ab_forms_sizes.txt
MI355_ENGINE_LIB=prmers_amd/libmi355_engine_exp.so MI355_COOP=1 python tools/exp_coop_check.py > $O/job2_exp_coop.txt 2>&1; echo "exp coop rc=$?"; tail -3 $O/job2_exp_coop.txt
