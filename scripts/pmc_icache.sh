cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace -d $R/gpurun_out/pmc_ic -o res -- python3 scripts/bench_quick.py 3072) > $R/gpurun_out/pmc_ic.log 2>&1
tail -3 $R/gpurun_out/pmc_ic.log
cd $R && python3 scripts/rocpd_pmc.py $(find gpurun_out/pmc_ic -name "*.db") | head -20
