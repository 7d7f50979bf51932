#!/bin/bash
# one-off: HBM write / fetch bytes of one bench step (run through gpurun); usage: pmc_write.sh [samples]
R=$GRAFT_REPO_ROOT
export JQ_BENCH_SAMPLES=${1:-3072}
cd /tmp && export TMPDIR=/tmp
CMD="python3 bench.py --steps 1 --warmup 0 --no-extras"
for c in WRITE_SIZE FETCH_SIZE; do
(cd $R && rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/prof_w_$c -o res -- $CMD) > $R/gpurun_out/prof_w_$c.log 2>&1
done
cd $R
python3 scripts/make_traffic_json.py gpurun_out/pmc_write.json --version x --samples $JQ_BENCH_SAMPLES $(find gpurun_out/prof_w_* -name "*.db")
python3 - <<'PY'
import json
j = json.load(open("gpurun_out/pmc_write.json"))
for k, e in j["kernels"].items():
    if "k_forward" in k or "k_backward" in k:
        print(k, {x: e[x] for x in e if x != "sq"})
PY
rm -rf gpurun_out/prof_w_*
