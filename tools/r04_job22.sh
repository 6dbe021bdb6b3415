#!/bin/bash
# round 4 (second session): non-temporal hint on the work-buffer accesses of the radix-8 kernels (-DMI355_NT: 1 stores, 2 loads, 3 both), same-box A/B
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 bash tools/ab_many.sh "136279841 250000013" prmers_amd/libmi355_engine.so prmers_amd/libmi355_engine_nt1.so prmers_amd/libmi355_engine_nt2.so prmers_amd/libmi355_engine_nt3.so > $O/job22_ab_nt.txt 2>&1
cat $O/job22_ab_nt.txt
