#!/bin/bash
# Everything the judged profile files come from, in one GPU call (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats of one bench step           -> gpurun_out/stats_<tag>.txt
#   2. rocprofv3 --pmc passes of the same command (HBM bytes; SQ)   -> gpurun_out/pmc_<tag>.json  (copied to profiles/r06_pmc.json,
#      which the bench line quotes: MFMA count, VALU per MFMA, stall fractions, HBM bytes per launch)
#   3. the default bench.py line                                    -> gpurun_out/bench_<tag>.log
# usage: scripts/profile_round.sh <tag>          (copy the three files into profiles/ afterwards)
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python3 bench.py --steps 1 --warmup 0 --no-extras"
(cd $R && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag}/stats -o res -- $CMD) > $R/gpurun_out/prof_${tag}_stats.log 2>&1
(cd $R && python3 scripts/rocpd_summary.py $(find gpurun_out/prof_${tag}/stats -name "*.db" | head -1) gpurun_out/stats_${tag}.txt; head -6 gpurun_out/stats_${tag}.txt)
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/prof_${tag}/pmc$i -o res -- $CMD) > $R/gpurun_out/prof_${tag}_pmc$i.log 2>&1
done
cd $R
ver=$(python3 -c "import juqbox_jl_amd._lib as l; print(l.load().jq_version().decode())")
python3 scripts/make_traffic_json.py gpurun_out/pmc_${tag}.json --version "$ver" --samples ${JQ_BENCH_SAMPLES:-3072} --nsteps 32386 $(find gpurun_out/prof_${tag}/pmc* -name "*.db") && head -30 gpurun_out/pmc_${tag}.json
# the bench line last: it quotes the PMC figures of THIS build
cp gpurun_out/pmc_${tag}.json profiles/r06_pmc.json
python3 bench.py > gpurun_out/bench_${tag}.log 2> gpurun_out/bench_${tag}.err
tail -1 gpurun_out/bench_${tag}.log | cut -c1-1500
rm -rf gpurun_out/prof_${tag}      # (the rocpd databases are large; their summaries above are what is kept)
