#!/bin/bash
# round 4 (second session): prime-factor radix-5 columns, final form (labels as a linear form, root table up to m + 2^20): whole GPU suite, then A/B
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/job28_pytest.log 2>&1; rc=$?; tail -4 $O/job28_pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 500 bash tools/ab_many.sh "205271257 136279841" prmers_amd/libmi355_engine_base.so prmers_amd/libmi355_engine.so 2>&1 | grep -v amdgpu.ids > $O/job28_ab.txt
cat $O/job28_ab.txt
