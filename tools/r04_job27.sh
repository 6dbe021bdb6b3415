#!/bin/bash
# round 4 (second session): the row kernels now take the column frequency through col_label(): same-box A/B at shapes without radix-5 columns (nothing may change)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 700 bash tools/ab_many.sh "136279841 57885161 9815459" prmers_amd/libmi355_engine_base.so prmers_amd/libmi355_engine.so 2>&1 | grep -v amdgpu.ids > $O/job27_ab_label.txt
cat $O/job27_ab_label.txt
