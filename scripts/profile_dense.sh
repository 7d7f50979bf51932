#!/bin/bash
# Profile of the dense-operator path (bench.py's `dense_operator` block: cnot3 dimensions with a dense Hermitian drift, kernels
# k_forward / k_backward<6, 5>), in one GPU call:
#   1. rocprofv3 --kernel-trace --stats of ONE full-length evaluation     -> gpurun_out/dense_stats_<tag>.txt  (-> profiles/r06_dense_kernel_stats.txt)
#   2. rocprofv3 --pmc passes of the same command                          -> gpurun_out/pmc_dense_<tag>.json   (-> profiles/r06_pmc_dense.json)
#   3. the block itself (quotes the PMC record of THIS build)              -> gpurun_out/dense_<tag>.log
# usage: scripts/profile_dense.sh <tag> [samples]
tag=$1
ns=${2:-4096}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python3 bench.py --dense-only $ns"
(cd $R && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profd_${tag}/stats -o res -- $CMD) > $R/gpurun_out/profd_${tag}_stats.log 2>&1
(cd $R && python3 scripts/rocpd_summary.py $(find gpurun_out/profd_${tag}/stats -name "*.db" | head -1) gpurun_out/dense_stats_${tag}.txt; head -8 gpurun_out/dense_stats_${tag}.txt)
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/profd_${tag}/pmc$i -o res -- $CMD) > $R/gpurun_out/profd_${tag}_pmc$i.log 2>&1
done
cd $R
ver=$(python3 -c "import juqbox_jl_amd._lib as l; print(l.load().jq_version().decode())")
python3 scripts/make_traffic_json.py gpurun_out/pmc_dense_${tag}.json --version "$ver" --samples $ns --nsteps 32386 $(find gpurun_out/profd_${tag}/pmc* -name "*.db") && head -40 gpurun_out/pmc_dense_${tag}.json
cp gpurun_out/pmc_dense_${tag}.json profiles/r06_pmc_dense.json
python3 bench.py --dense-only $ns > gpurun_out/dense_${tag}.log 2> gpurun_out/dense_${tag}.err
tail -1 gpurun_out/dense_${tag}.log | cut -c1-3000
rm -rf gpurun_out/profd_${tag}
