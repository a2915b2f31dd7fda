#!/bin/bash
# rocprofv3 kernel trace of the search loops (greedy batch 64 x 7 runs, beam-3 batch 20; untrained weights: every run takes all
# 29 steps): usage r2_prof_decode.sh <tag>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$1
mkdir -p $OUT/prof_decode
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_decode -o dec -- python $REPO/bench.py --steps 1 --warmup 0 --decode-train-steps 0 --no-cpu-baseline --no-extras > $OUT/prof_decode/bench.log 2>&1
python $REPO/scripts/rocpd_stats.py $(find $OUT/prof_decode -name "*results.db" | head -1) 203 > $OUT/kernel_stats_decode.md 2>&1
find $OUT/prof_decode -name "*.db" -delete
head -45 $OUT/kernel_stats_decode.md | cut -c1-170
