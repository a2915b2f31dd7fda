#!/bin/bash
# Round-2 GPU session: parity tests, bench lines (fp32 / bf16), per-shape GEMM timings, rocprofv3 kernel traces.
# usage (through gpurun): bash scripts/r2_gpu.sh <tag> [tests|notests]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2a}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
if [ "${2:-tests}" = "tests" ]; then
  timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 300 > $OUT/tests.log 2>&1
  echo "tests rc=$?" >> $OUT/tests.log
  tail -n 40 $OUT/tests.log
fi
timeout 600 python bench.py > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "bench f32 rc=$?"
timeout 600 python bench.py --dtype bf16 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; echo "bench bf16 rc=$?"
timeout 300 python scripts/gemm_step_shapes.py f32 32 > $OUT/gemm_shapes_f32_b32.txt 2>&1
timeout 300 python scripts/gemm_step_shapes.py bf16 32 > $OUT/gemm_shapes_bf16_b32.txt 2>&1
timeout 300 python scripts/gemm_step_shapes.py bf16 64 > $OUT/gemm_shapes_bf16_b64.txt 2>&1
cd /tmp
for DT in f32 bf16; do
  mkdir -p $OUT/prof_$DT
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$DT -o step -- python $REPO/bench.py --dtype $DT --steps 10 --warmup 2 --no-decode --no-cpu-baseline --no-extras > $OUT/prof_$DT/bench.log 2>&1
  python $REPO/scripts/rocpd_stats.py $(find $OUT/prof_$DT -name "*results.db" | head -1) 15 > $OUT/kernel_stats_$DT.md 2>&1
  find $OUT/prof_$DT -name "*.db" -delete      # the sqlite traces are large; the summaries are what is kept
done
cd $REPO
head -c 600 $OUT/bench_f32.json; echo; head -c 600 $OUT/bench_bf16.json; echo
tail -n 3 $OUT/gemm_shapes_bf16_b32.txt
