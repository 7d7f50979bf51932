#!/bin/bash
# HBM write / fetch bytes per propagator launch of one bench step for library variants built by scripts/exp_variants.sh
# (run through gpurun); usage: exp_pmc_write.sh tag [tag ...]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  export JQ_LIB=$R/juqbox.jl_amd/exp/libjq_$tag.so
  for c in WRITE_SIZE FETCH_SIZE; do
    (cd $R && rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/prof_x_${tag}_$c -o res -- python3 bench.py --steps 1 --warmup 0 --no-extras) > $R/gpurun_out/prof_x_${tag}_$c.log 2>&1
  done
  (cd $R && python3 scripts/make_traffic_json.py gpurun_out/pmc_x_$tag.json --version "$tag" --samples 3072 $(find gpurun_out/prof_x_${tag}_* -name "*.db") > /dev/null
   python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
j = json.load(open("gpurun_out/pmc_x_%s.json" % tag))
for k, e in j["kernels"].items():
    if k.startswith("k_backward"):
        print("%-14s %s: WRITE %.2f GB  2xFETCH %.2f GB  total %.2f GB per launch" % (tag, k, e["WRITE_SIZE_KiB_per_launch"] * 1024 / 1e9, 2 * e["FETCH_SIZE_KiB_per_launch"] * 1024 / 1e9, e["hbm_bytes_per_launch"] / 1e9))
PY
  )
  rm -rf $R/gpurun_out/prof_x_${tag}_*
done
