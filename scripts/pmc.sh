#!/bin/bash
# usage: scripts/pmc.sh <tag> <python script + args>   -- separate rocprofv3 --pmc passes (never mixed with sys traces)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
i=0
for ctrs in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" \
            "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/$tag/p$i -- python "$@" > gpurun_out/$tag/p$i.log 2>&1
  i=$((i+1))
done
ls gpurun_out/$tag/*/*/ | head -30
