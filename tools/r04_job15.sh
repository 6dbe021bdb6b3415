#!/bin/bash
# final tree of round 4: whole GPU suite, smoke, the experimental library's checks, default bench line
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/job15_pytest.txt 2>&1 && tail -3 gpurun_out/r04/job15_pytest.txt && \
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/job15_smoke.txt 2>&1 && cat gpurun_out/r04/job15_smoke.txt && \
MI355_ENGINE_LIB=prmers_amd/libmi355_engine_exp.so MI355_COOP=1 python tools/exp_coop_check.py > gpurun_out/r04/job15_exp.txt 2>&1 && tail -3 gpurun_out/r04/job15_exp.txt && \
python bench.py > gpurun_out/r04/job15_bench.json 2> gpurun_out/r04/job15_bench.err && cat gpurun_out/r04/job15_bench.json
