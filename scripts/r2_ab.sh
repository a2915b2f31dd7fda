#!/bin/bash
# A/B runs of bench.py on ONE box (the pool's boxes differ by a few per cent): usage r2_ab.sh <tag> "<env A>" "<env B>" [bench args]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; A=$2; B=$3; shift 3
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    env $E timeout 300 python bench.py --no-decode --no-cpu-baseline --no-extras --steps 40 "$@" > $OUT/${v}_$rep.json 2> $OUT/${v}_$rep.err
    python - <<PY
import json
d=json.load(open("$OUT/${v}_$rep.json"))
print("$v rep $rep [$E]", round(d["value"]), "commits/s", round(d["ms_per_step"],3), "ms", {k:round(x,2) for k,x in d["kernel_time_ms_per_step"].items()})
PY
  done
done
