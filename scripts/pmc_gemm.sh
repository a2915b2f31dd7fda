#!/bin/bash
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_gemm
mkdir -p $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -o g -- python $REPO/scripts/gemm_one.py $@ > $OUT/$tag.log 2>&1
done
find $OUT -name "*counter_collection.csv" | head
