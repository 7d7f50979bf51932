#!/bin/bash
# Development aid: LDS bank-conflict counters of an arbitrary python command.  usage: scripts/pmc_lds.sh <tag> <script> [args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace -d $R/gpurun_out/pmc_lds_$tag -o res -- python3 "$@") > $R/gpurun_out/pmc_lds_$tag.log 2>&1
cd $R && python3 scripts/rocpd_pmc.py $(find gpurun_out/pmc_lds_$tag -name "*.db") | grep -E "^_Z|BANK|IDX" | head -24
