#!/bin/bash
# round 4 (second session), final tree: rocprofv3 profile of the headline command (kernel trace + PMC passes)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 700 bash tools/profile.sh r04b_final > $O/job34_profile_c3.log 2>&1; echo "profile c3 rc=$?"
python - <<'PY'
import json
s=json.load(open("gpurun_out/r04b_final/summary.json"))
for r in s["kernel_stats"][:3]: print(r["name"], r["calls"], r["avg_ns"])
for k,v in s["traffic"].items():
    if "cols" in k or "rows" in k: print(k, v["hbm_bytes_per_launch"])
for k,v in s["sq"].items():
    if "cols" in k or "rows" in k: print(k, v["SQ_INSTS_VALU"], v["SQ_LDS_BANK_CONFLICT"])
PY
