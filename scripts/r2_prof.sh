#!/bin/bash
# rocprofv3 kernel trace of the training step for one dtype: usage r2_prof.sh <tag> <f32|bf16>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$1
mkdir -p $OUT/prof_$2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$2 -o step -- python $REPO/bench.py --dtype $2 --steps 10 --warmup 2 --no-decode --no-cpu-baseline --no-extras > $OUT/prof_$2/bench.log 2>&1
python $REPO/scripts/rocpd_stats.py $(find $OUT/prof_$2 -name "*results.db" | head -1) 15 > $OUT/kernel_stats_$2.md 2>&1
find $OUT/prof_$2 -name "*.db" -delete
tail -40 $OUT/kernel_stats_$2.md
