#!/bin/bash
# Copy one round-4 GPU session's outputs (gpurun_out/<tag>) into profiles/ under the names DESIGN.md cites, stamping every
# file with the commit it was built from and the training rate of the box it ran on (the pool has a slow and a fast class).
#   bash scripts/r4_collect.sh r4final <commit>
TAG=${1:-r4final}; COMMIT=${2:-$(git rev-parse --short HEAD)}
S=gpurun_out/$TAG
VAL=$(python -c "import json;print(round(json.loads(open('$S/bench_f32.json').read().strip().splitlines()[-1])['value']))")
STAMP="<!-- round 4, session $TAG, built from commit $COMMIT; this box: $VAL commits/s on the default bench line (pool classes at this commit: ~10 700 slow, ~11 400 fast) -->"
stamp() { { echo "$STAMP"; echo; cat "$1"; } > "$2"; }
for DT in f32 bf16; do
  [ -f $S/kernel_stats_$DT.md ] && stamp $S/kernel_stats_$DT.md profiles/r4_kernel_stats_$DT.md
  [ -f $S/timeline_$DT.md ] && stamp $S/timeline_$DT.md profiles/r4_timeline_$DT.md
  [ -f $S/pmc_traffic_$DT.md ] && stamp $S/pmc_traffic_$DT.md profiles/r4_pmc_traffic_$DT.md
  [ -f $S/bench_$DT.json ] && tail -n 1 $S/bench_$DT.json > profiles/r4_bench_$DT.json
done
[ -f $S/kernel_stats_decode.md ] && stamp $S/kernel_stats_decode.md profiles/r4_kernel_stats_decode.md
[ -f $S/traffic.json ] && cp $S/traffic.json profiles/traffic.json
[ -f $S/kernel_classes.json ] && cp $S/kernel_classes.json profiles/r4_kernel_classes.json
[ -f $S/tests.log ] && { echo "# $STAMP" > profiles/r4_gpu_tests.log; tail -n 14 $S/tests.log >> profiles/r4_gpu_tests.log; }
ls -la profiles | grep r4_
