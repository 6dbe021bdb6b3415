#!/bin/bash
# round 4, third box call: field forms with the fall-through reduction tail -- per-operation cycles (old / new), then a same-box A/B of five
# builds of the library at C3
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
./tools/microbench_gf_r3forms > $O/job3_microbench_gf_r3forms.txt 2>&1; echo "gf r3 rc=$?"
./tools/microbench_gf > $O/job3_microbench_gf_new.txt 2>&1; echo "gf new rc=$?"
paste $O/job3_microbench_gf_r3forms.txt $O/job3_microbench_gf_new.txt | cut -c1-200
./tools/microbench_alt_butterfly > $O/job3_microbench_alt_butterfly.txt 2>&1; cat $O/job3_microbench_alt_butterfly.txt
python -c "
import sys; sys.path.insert(0,'.')
from prmers_amd.engine import load_library
L=load_library(); print('selftest', L.mi355_engine_selftest(0), L.mi355_engine_last_error())"
LIBS="prmers_amd/libmi355_engine_r3forms.so prmers_amd/libmi355_engine.so prmers_amd/libmi355_engine_sh48old.so prmers_amd/libmi355_engine_noaddmask.so prmers_amd/libmi355_engine_sh48old_noaddmask.so"
tools/ab_many.sh "136279841" $LIBS 2>&1 | grep -v amdgpu.ids > $O/job3_ab_tail_c3.txt; cat $O/job3_ab_tail_c3.txt
