#!/bin/bash
# one-off PMC pass: branch / scalar / instruction-fetch counters of one bench step (run through gpurun)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python3 bench.py --steps 1 --warmup 0 --no-extras"
(cd $R && rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS SQ_WAVES SQ_INSTS_LDS --kernel-trace -d $R/gpurun_out/prof_branch -o res -- $CMD) > $R/gpurun_out/prof_branch.log 2>&1
cd $R
python3 scripts/make_traffic_json.py gpurun_out/pmc_branch.json --version x --samples 3072 $(find gpurun_out/prof_branch -name "*.db")
python3 - <<'PY'
import json
j = json.load(open("gpurun_out/pmc_branch.json"))
for k, e in j["kernels"].items():
    if "sq" in e and ("k_forward" in k or "k_backward" in k):
        print(k, e["launches"], {c: v / max(e["launches"], 1) for c, v in e["sq"].items()})
PY
rm -rf gpurun_out/prof_branch
