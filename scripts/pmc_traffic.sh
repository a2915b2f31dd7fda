#!/bin/bash
# HBM traffic of the step's kernels from the L2 memory-side counters (separate --pmc passes, no tracing domains):
#   FETCH_SIZE / WRITE_SIZE in KiB per dispatch; on gfx950 FETCH_SIZE reports half of a wide coalesced read
#   (MI355X_MICROARCH.md "HBM") -> the summary doubles it for the 16-byte-per-lane streaming kernels.
# usage: pmc_traffic.sh <f32|bf16>
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
DT=${1:-f32}
OUT=$REPO/gpurun_out/pmc_traffic_$DT
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 170 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o t -- python $REPO/bench.py --dtype $DT --steps 2 --warmup 1 --no-decode --no-cpu-baseline --no-extras > $OUT/$c.log 2>&1
  # keep only the counter table (the pass also writes large agent / kernel tables)
  find $OUT/$c -type f ! -name "*counter_collection.csv" -delete
done
find $OUT -name "*counter_collection.csv" | head
