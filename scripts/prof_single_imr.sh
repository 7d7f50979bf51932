cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_single_imr -o res -- python3 scripts/prof_case.py cnot3 1 2 imr) > $R/gpurun_out/prof_single_imr.log 2>&1
cd $R && python3 scripts/rocpd_summary.py $(find gpurun_out/prof_single_imr -name "*.db" | head -1) gpurun_out/stats_single_imr.txt; head -8 gpurun_out/stats_single_imr.txt
rm -rf gpurun_out/prof_single_imr
