#!/bin/bash
# Copy one round-6 GPU session's outputs (gpurun_out/<tag>) into profiles/ under the names DESIGN.md cites, stamping every
# file with the commit it was built from and the training rate of the box it ran on (the pool has a slow and a fast class).
#   bash scripts/r6_collect.sh r6final <commit>
TAG=${1:-r6final}; COMMIT=${2:-$(git rev-parse --short HEAD)}
S=gpurun_out/$TAG
VAL=$(python -c "import json;print(round(json.loads(open('$S/bench_f32.json').read().strip().splitlines()[-1])['value']))")
STAMP="<!-- round 6, session $TAG, built from commit $COMMIT; this box: $VAL commits/s on the default bench line (pool classes this round: ~12 600 slow, ~13 600 fast at this commit) -->"
stamp() { { echo "$STAMP"; echo; cat "$1"; } > "$2"; }
for DT in f32 bf16; do
  [ -f $S/kernel_stats_$DT.md ] && stamp $S/kernel_stats_$DT.md profiles/r6_kernel_stats_$DT.md
  [ -f $S/timeline_$DT.md ] && stamp $S/timeline_$DT.md profiles/r6_timeline_$DT.md
  [ -f $S/pmc_traffic_$DT.md ] && stamp $S/pmc_traffic_$DT.md profiles/r6_pmc_traffic_$DT.md
  [ -f $S/bench_$DT.json ] && tail -n 1 $S/bench_$DT.json > profiles/r6_bench_$DT.json
done
[ -f $S/kernel_stats_decode.md ] && stamp $S/kernel_stats_decode.md profiles/r6_kernel_stats_decode.md
[ -f $S/traffic.json ] && cp $S/traffic.json profiles/traffic.json
[ -f $S/kernel_classes.json ] && cp $S/kernel_classes.json profiles/r6_kernel_classes.json
[ -f $S/tests.log ] && { echo "# $STAMP" > profiles/r6_gpu_tests.log; tail -n 14 $S/tests.log >> profiles/r6_gpu_tests.log; }

for DT in f32 bf16; do [ -f $S/bench_detail_$DT.json ] && cp $S/bench_detail_$DT.json profiles/r6_bench_detail_$DT.json; done
ls profiles | grep r6_
