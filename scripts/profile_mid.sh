R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python3 scripts/one_eval_mid.py 32"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/r06G/pmc$i -o res -- $CMD) > $R/gpurun_out/r06G/pmc$i.log 2>&1
done
cd $R
python3 scripts/make_traffic_json.py gpurun_out/r06G/pmc_mid.json $(find gpurun_out/r06G/pmc* -name "*.db")
rm -rf gpurun_out/r06G/pmc1 gpurun_out/r06G/pmc2 gpurun_out/r06G/pmc3
