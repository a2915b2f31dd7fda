#!/bin/bash
# rocprofv3 kernel trace of the training step (run on the GPU box via gpurun); output: gpurun_out/prof_<tag>/step_results.db
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$1
mkdir -p $OUT
timeout 170 rocprofv3 --kernel-trace --stats -d $OUT -o step -- python $REPO/bench.py --steps 10 --warmup 2 --no-decode --no-cpu-baseline > $OUT/bench.log 2>&1
ls $OUT
