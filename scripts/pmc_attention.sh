#!/bin/bash
# stall / pipe counters of the cross-attention kernels (separate --pmc passes, no tracing domains)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_attn
mkdir -p $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 100 rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -o a -- python $REPO/scripts/attn_one.py > $OUT/$tag.log 2>&1
done
for f in $(find $OUT -name "*counter_collection.csv"); do python $REPO/scripts/pmc_summary.py $f attention; done
