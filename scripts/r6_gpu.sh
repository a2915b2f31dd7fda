#!/bin/bash
# Round-6 GPU session (through gpurun): bash scripts/r6_gpu.sh <tag> <stages...>   (stages as in r4_gpu.sh, trimmed)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r6a}; shift
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
PREV=$REPO/fira_icse_amd/libfira_hip_prev.so
one() { env $1 timeout 500 python bench.py $2 --no-decode --no-cpu-baseline --no-extras --steps 40 --detail $OUT/one_detail.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_time_ms_per_step']; print(round(d['value']), 'commits/s', round(d['ms_per_step'],3), 'ms', 'attn', round(k['attention'],3), 'gemm', round(k['gemm'],3), 'rowops', round(k['rowops'],3), 'copy', round(k['copy'],3), 'gcn', round(k['gcn'],3), 'host', round(d['host_enqueue_ms_per_step'],2))"; }
for ST in "$@"; do
  case $ST in
    tests)
      timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 300 > $OUT/tests.log 2>&1
      echo "tests rc=$?" >> $OUT/tests.log; tail -n 25 $OUT/tests.log ;;
    optests)
      timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_spmm_bench_gpu.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 ${OPK:+-k "$OPK"} > $OUT/optests.log 2>&1
      echo "optests rc=$?"; tail -n 15 $OUT/optests.log ;;
    modeltests)
      timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_decode_gpu.py tests/test_edge_cases_gpu.py tests/test_dropout_gpu.py tests/test_large_gpu.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -x > $OUT/modeltests.log 2>&1
      echo "modeltests rc=$?"; tail -n 15 $OUT/modeltests.log ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 3 ;;
    ab3)   # two BUILDS on one box: libfira_hip_prev.so (the previous commit's build) vs the tree's library
      for i in 1 2 3; do
        for V in "FIRA_HIP_LIB=$PREV" "FIRA_X=1"; do
          echo -n "${V##*/} f32 b32: "; one "$V" "--batch 32"
          [ $i = 1 ] && { echo -n "${V##*/} f32 b64: "; one "$V" "--batch 64"; echo -n "${V##*/} bf16 b64: "; one "$V" "--dtype bf16 --batch 64"; }
        done
      done 2>&1 | tee $OUT/ab3.txt ;;
    abenv)   # environment switches of ONE build, alternating: ABENV="A=1|B=2|..."
      IFS='|' read -ra VARS <<< "$ABENV"
      for i in 1 2 3; do
        for V in "${VARS[@]}"; do
          echo -n "$V f32 b${ABBATCH:-32}: "; one "$V" "--batch ${ABBATCH:-32}"
        done
      done 2>&1 | tee $OUT/abenv.txt ;;
    abenv16)
      IFS='|' read -ra VARS <<< "$ABENV"
      for i in 1 2 3; do
        for V in "${VARS[@]}"; do
          echo -n "$V bf16 b64: "; one "$V" "--dtype bf16 --batch 64"
        done
      done 2>&1 | tee $OUT/abenv16.txt ;;
    bf16tests)
      timeout 900 python -m pytest tests/test_bf16_gpu.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 > $OUT/bf16tests.log 2>&1
      echo "bf16tests rc=$?"; tail -n 8 $OUT/bf16tests.log ;;
    abdec)
      for i in 1 2; do
        for V in "FIRA_HIP_LIB=$PREV" "FIRA_X=1"; do echo -n "${V##*/} decode: "; env $V timeout 200 python scripts/decode_only.py 2>/dev/null | tail -n 2 | tr "\n" ";"; echo; done
      done 2>&1 | tee $OUT/abdec.txt ;;
    abdecenv)   # environment switches of ONE build on the greedy search, alternating: ABENV_DEC="A=1|B=2"
      IFS='|' read -ra VARS <<< "$ABENV_DEC"
      for i in 1 2; do
        for V in "${VARS[@]}"; do echo -n "$V decode: "; env $V timeout 200 python scripts/decode_only.py 2>/dev/null | tail -n 2 | tr "\n" ";"; echo; done
      done 2>&1 | tee $OUT/abdecenv.txt ;;
    waitprobe)   # where the caller's stream stands waiting for the library's streams (FIRA_WAIT_PROBE=1: engine.hip lines)
      for B in 32 64; do
        echo "batch $B f32:"; FIRA_WAIT_PROBE=1 timeout 300 python bench.py --batch $B --steps 40 --no-decode --no-cpu-baseline --no-extras --detail $OUT/one_detail.json 2>&1 | grep "wait probe" | tail -n 2
      done 2>&1 | tee $OUT/waitprobe.txt
      echo "batch 64 bf16:"; FIRA_WAIT_PROBE=1 timeout 300 python bench.py --dtype bf16 --batch 64 --steps 40 --no-decode --no-cpu-baseline --no-extras --detail $OUT/one_detail.json 2>&1 | grep "wait probe" | tail -n 2 | tee -a $OUT/waitprobe.txt ;;
    bench)
      timeout 900 python bench.py --detail $OUT/bench_detail_f32.json > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "bench f32 rc=$? bytes=$(wc -c < $OUT/bench_f32.json)"; cat $OUT/bench_f32.json; tail -n 5 $OUT/bench_f32.err ;;
    bench16)
      timeout 600 python bench.py --dtype bf16 --detail $OUT/bench_detail_bf16.json > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; echo "bench bf16 rc=$?"; head -c 900 $OUT/bench_bf16.json; echo ;;
    prof|prof16)
      DT=f32; [ $ST = prof16 ] && DT=bf16
      cd /tmp; mkdir -p $OUT/prof_$DT
      timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$DT -o step -- python $REPO/bench.py --dtype $DT --steps 10 --warmup 2 --no-decode --no-cpu-baseline --no-extras --detail $OUT/prof_$DT/detail.json > $OUT/prof_$DT/bench.log 2>&1
      VAL=$(grep -o '"value": [0-9.]*' $OUT/prof_$DT/bench.log | head -1 | cut -d' ' -f2)
      BS=32; [ $DT = bf16 ] && BS=64
      ROCPD_CLASSES_JSON=$OUT/kernel_classes.json ROCPD_DTYPE=$DT ROCPD_BATCH=$BS ROCPD_COMMIT=${R6_COMMIT:-unknown} ROCPD_VALUE=$VAL \
        ROCPD_SOURCE="rocprofv3 --kernel-trace --stats -- python bench.py --dtype $DT --steps 10 --warmup 2 --no-decode --no-cpu-baseline --no-extras (15 steps in the trace; under the profiler)" \
        python $REPO/scripts/rocpd_stats.py $(find $OUT/prof_$DT -name "*results.db" | head -1) 15 $OUT/timeline_$DT.md > $OUT/kernel_stats_$DT.md 2>&1
      find $OUT/prof_$DT -name "*.db" -delete
      cd $REPO; head -n 60 $OUT/kernel_stats_$DT.md | cut -c1-200 ;;
    profdec)
      cd /tmp; mkdir -p $OUT/prof_dec
      timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_dec -o dec -- python $REPO/scripts/decode_only.py > $OUT/prof_dec/run.log 2>&1
      python $REPO/scripts/rocpd_stats.py $(find $OUT/prof_dec -name "*results.db" | head -1) 1 > $OUT/kernel_stats_decode.md 2>&1
      find $OUT/prof_dec -name "*.db" -delete
      cd $REPO; head -n 40 $OUT/kernel_stats_decode.md | cut -c1-200 ;;
    pmc)
      for DT in f32 bf16; do
        bash scripts/pmc_traffic.sh $DT > /dev/null 2>&1
        ROCPD_COMMIT=${R6_COMMIT:-unknown} python scripts/pmc_traffic_summary.py gpurun_out/pmc_traffic_$DT $OUT/pmc_traffic_$DT.md $OUT/traffic.json $DT; echo "pmc $DT rc=$?"
        find gpurun_out/pmc_traffic_$DT -name "*.csv" -delete
      done
      cat $OUT/traffic.json | head -40 ;;
    *) echo "running: $ST"; eval "$ST" ;;
  esac
done
