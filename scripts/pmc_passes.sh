#!/bin/bash
# Development aid: SQ instruction-mix / stall counters of one cnot3 ensemble evaluation, one rocprofv3 --pmc pass per
# counter group (never combined with other trace domains).  usage: scripts/pmc_passes.sh <tag> [nsamples]
tag=$1; ns=${2:-4096}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_BRANCH"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pmc_${tag}/p$i -o res -- python3 scripts/bench_quick.py $ns) > $R/gpurun_out/pmc_${tag}_p$i.log 2>&1
  tail -2 $R/gpurun_out/pmc_${tag}_p$i.log
done
cd $R && python3 scripts/rocpd_pmc.py $(find gpurun_out/pmc_${tag} -name "*.db") > gpurun_out/pmc_${tag}.txt 2>&1; head -60 gpurun_out/pmc_${tag}.txt
