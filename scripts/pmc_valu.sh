#!/bin/bash
# instruction-mix / stall counters of every kernel of the training step (one --pmc pass, no tracing domains)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_valu
mkdir -p $OUT
timeout 170 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $OUT -o v -- python $REPO/bench.py --steps 2 --warmup 1 --no-decode --no-cpu-baseline > $OUT/run.log 2>&1
find $OUT -name "*counter_collection.csv"
