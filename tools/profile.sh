#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel-trace stats + separate PMC passes for HBM traffic and SQ activity.
#   tools/profile.sh <name> [bench args...]
# Raw rocprof output stays in /tmp on the box; gpurun_out/<name>/ receives the summary (tools/summarize_profile.py), the
# kernel_stats.csv of the trace pass and the bench logs.  FETCH_SIZE and WRITE_SIZE do not fit one pass
# (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -e
NAME=$1; shift
RAW=/tmp/prof_$NAME
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
rm -rf $RAW; mkdir -p $RAW $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline "$@" > $RAW/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $RAW/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $RAW/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $RAW/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $RAW/bench_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $RAW/pmc_sq2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $RAW/bench_sq2.log 2>&1 || echo "second SQ pass failed (counter set)" >> $OUT/notes.txt
python3 $GRAFT_REPO_ROOT/tools/summarize_profile.py $RAW > $OUT/summary.json
cp $RAW/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
cp $RAW/bench_*.log $OUT/
ls -la $OUT
