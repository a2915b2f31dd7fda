#!/bin/bash
# HBM traffic of the step's kernels from the L2 memory-side counters (separate --pmc passes, no tracing domains):
#   FETCH_SIZE / WRITE_SIZE in KiB per dispatch; on gfx950 FETCH_SIZE reports half of a wide coalesced read
#   (MI355X_MICROARCH.md "HBM") -> the summary doubles it for the 16-byte-per-lane streaming kernels.
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_traffic
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 170 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o t -- python $REPO/bench.py --steps 2 --warmup 1 --no-decode --no-cpu-baseline > $OUT/$c.log 2>&1
done
find $OUT -name "*counter_collection.csv"
