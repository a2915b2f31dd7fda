#!/bin/bash
# Round-3 GPU session (through gpurun): bash scripts/r3_gpu.sh <tag> <stages...>
#   stages: chain tests bench bench16 prof prof16 shapes
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r3a}; shift
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
for ST in "$@"; do
  case $ST in
    chain)
      FIRA_SMALL_GEMM=1 timeout 300 python scripts/gemm_chain.py f32 > $OUT/chain_f32_old.md 2>&1; echo "chain old rc=$?"
      FIRA_SMALL_GEMM=2 timeout 300 python scripts/gemm_chain.py f32 > $OUT/chain_f32_new.md 2>&1; echo "chain new rc=$?"
      FIRA_SMALL_GEMM=1 timeout 300 python scripts/gemm_chain.py bf16 > $OUT/chain_bf16_old.md 2>&1; echo "chain bf16 old rc=$?"
      FIRA_SMALL_GEMM=2 timeout 300 python scripts/gemm_chain.py bf16 > $OUT/chain_bf16_new.md 2>&1; echo "chain bf16 new rc=$?"
      FIRA_SMALL_TILES=1024 timeout 300 python scripts/gemm_chain.py f32 2880,5100,7680 > $OUT/chain_f32_big_t1024.md 2>&1
      FIRA_SMALL_TILES=8192 timeout 300 python scripts/gemm_chain.py f32 2880,5100,7680 > $OUT/chain_f32_big_t8192.md 2>&1
      tail -n 26 $OUT/chain_f32_new.md; tail -n 14 $OUT/chain_bf16_old.md; tail -n 14 $OUT/chain_bf16_new.md; tail -n 14 $OUT/chain_f32_big_t1024.md; tail -n 14 $OUT/chain_f32_big_t8192.md ;;
    chain16)
      for T in 1024 2048 4096; do
        FIRA_SMALL_TILES16=$T timeout 300 python scripts/gemm_chain.py bf16 960,1920,2880,5100 > $OUT/chain_bf16_t$T.md 2>&1; echo "chain16 $T rc=$?"; grep "^| [0-9]" $OUT/chain_bf16_t$T.md
      done ;;
    events)
      timeout 120 python scripts/event_cost.py > $OUT/event_cost.txt 2>&1; cat $OUT/event_cost.txt ;;
    decab)
      FIRA_DECODE_ATTN=0 timeout 300 python scripts/decode_only.py > $OUT/decode_old.txt 2>&1; cat $OUT/decode_old.txt | tail -2
      FIRA_DECODE_ATTN=1 timeout 300 python scripts/decode_only.py > $OUT/decode_new.txt 2>&1; cat $OUT/decode_new.txt | tail -2 ;;
    dectests)
      timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_large_gpu.py tests/test_ops_gpu.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -k "decode or search or greedy or beam or attention" > $OUT/dectests.log 2>&1
      tail -n 8 $OUT/dectests.log ;;
    tests)
      timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 300 > $OUT/tests.log 2>&1
      echo "tests rc=$?" >> $OUT/tests.log; tail -n 25 $OUT/tests.log ;;
    bench)
      timeout 900 python bench.py > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "bench f32 rc=$?"; head -c 1500 $OUT/bench_f32.json; echo; tail -n 5 $OUT/bench_f32.err ;;
    bench16)
      timeout 600 python bench.py --dtype bf16 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; echo "bench bf16 rc=$?"; head -c 600 $OUT/bench_bf16.json; echo ;;
    prof|prof16)
      DT=f32; [ $ST = prof16 ] && DT=bf16
      cd /tmp; mkdir -p $OUT/prof_$DT
      timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$DT -o step -- python $REPO/bench.py --dtype $DT --steps 10 --warmup 2 --no-decode --no-cpu-baseline --no-extras > $OUT/prof_$DT/bench.log 2>&1
      python $REPO/scripts/rocpd_stats.py $(find $OUT/prof_$DT -name "*results.db" | head -1) 15 $OUT/timeline_$DT.md > $OUT/kernel_stats_$DT.md 2>&1
      find $OUT/prof_$DT -name "*.db" -delete
      cd $REPO; head -n 45 $OUT/kernel_stats_$DT.md ;;
    profdec)
      cd /tmp; mkdir -p $OUT/prof_dec
      timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_dec -o dec -- python $REPO/scripts/decode_only.py > $OUT/prof_dec/run.log 2>&1
      python $REPO/scripts/rocpd_stats.py $(find $OUT/prof_dec -name "*results.db" | head -1) 1 > $OUT/kernel_stats_decode.md 2>&1
      find $OUT/prof_dec -name "*.db" -delete
      cd $REPO; head -n 40 $OUT/kernel_stats_decode.md ;;
    cross)
      timeout 600 python scripts/spmm_crossover.py > $OUT/spmm_crossover.md 2> $OUT/spmm_crossover.err; echo "crossover rc=$?"; cat $OUT/spmm_crossover.md; tail -n 3 $OUT/spmm_crossover.err ;;
    newtests)
      timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_spmm_bench_gpu.py tests/test_model_gpu.py tests/test_dropout_gpu.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -x > $OUT/newtests.log 2>&1
      tail -n 15 $OUT/newtests.log ;;
    ab)
      for i in 1 2; do
        for V in "FIRA_COMPACT_DEC=0 FIRA_ENC_WGRAD_GROUP=0" "FIRA_COMPACT_DEC=0 FIRA_ENC_WGRAD_GROUP=1" "FIRA_COMPACT_DEC=1 FIRA_ENC_WGRAD_GROUP=1"; do
          for BS in 32 64; do
            echo -n "$V batch $BS: "; env $V timeout 300 python bench.py --batch $BS --no-decode --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), 'commits/s', round(d['ms_per_step'],3), 'ms', 'host', round(d['host_enqueue_ms_per_step'],2))"
          done
        done
      done 2>&1 | tee $OUT/ab.txt ;;
    ab2)
      one() { env $1 timeout 200 python bench.py $2 --no-decode --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_time_ms_per_step']; print(round(d['value']), 'commits/s', round(d['ms_per_step'],3), 'ms', 'attn', round(k['attention'],3), 'rowops', round(k['rowops'],3), 'head', round(k['head'],3))"; }
      for i in 1 2; do
        for V in "FIRA_KV_PAD=0 FIRA_HEAD_FAST=0" "FIRA_KV_PAD=64 FIRA_HEAD_FAST=0" "FIRA_KV_PAD=64 FIRA_HEAD_FAST=1"; do
          echo -n "$V f32 b32: "; one "$V" "--batch 32"
          [ $i = 1 ] && { echo -n "$V bf16 b64: "; one "$V" "--dtype bf16 --batch 64"; }
        done
      done 2>&1 | tee $OUT/ab2.txt
      for V in "FIRA_KV_PAD=0" "FIRA_KV_PAD=64"; do echo -n "$V decode: "; env $V timeout 200 python scripts/decode_only.py 2>/dev/null | tail -n 2 | tr "\n" ";"; echo; done 2>&1 | tee -a $OUT/ab2.txt ;;
    ab3)   # two BUILDS on one box: fira_icse_amd/libfira_hip_prev.so (built from the previous commit) vs the tree's library
      one() { env $1 timeout 200 python bench.py $2 --no-decode --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_time_ms_per_step']; print(round(d['value']), 'commits/s', round(d['ms_per_step'],3), 'ms', 'host', round(d['host_enqueue_ms_per_step'],2))"; }
      for i in 1 2 3; do
        for V in "FIRA_HIP_LIB=$REPO/fira_icse_amd/libfira_hip_prev.so" "FIRA_X=1"; do
          echo -n "${V##*/} f32 b32: "; one "$V" "--batch 32"
          [ $i = 1 ] && { echo -n "${V##*/} bf16 b64: "; one "$V" "--dtype bf16 --batch 64"; }
        done
      done 2>&1 | tee $OUT/ab3.txt ;;
    gradtests)
      timeout 900 python -m pytest tests/test_model_gpu.py tests/test_large_gpu.py tests/test_dp_gpu.py tests/test_bf16_gpu.py tests/test_dropout_gpu.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -x > $OUT/gradtests.log 2>&1
      tail -n 6 $OUT/gradtests.log ;;
    abdec)   # decode loop: previous build vs the tree's library, alternating
      for i in 1 2; do
        for V in "FIRA_HIP_LIB=$REPO/fira_icse_amd/libfira_hip_prev.so" "FIRA_X=1"; do echo -n "${V##*/} decode: "; env $V timeout 200 python scripts/decode_only.py 2>/dev/null | tail -n 2 | tr "\n" ";"; echo; done
      done 2>&1 | tee $OUT/abdec.txt ;;
    pmc)
      for DT in f32 bf16; do
        bash scripts/pmc_traffic.sh $DT > /dev/null 2>&1
        python scripts/pmc_traffic_summary.py gpurun_out/pmc_traffic_$DT $OUT/pmc_traffic_$DT.md $OUT/traffic.json $DT; echo "pmc $DT rc=$?"
        find gpurun_out/pmc_traffic_$DT -name "*.csv" -delete
      done
      cat $OUT/traffic.json | head -40 ;;
    chainfinal)
      timeout 300 python scripts/gemm_chain.py f32 > $OUT/gemm_chain_f32.md 2>&1; timeout 300 python scripts/gemm_chain.py bf16 > $OUT/gemm_chain_bf16.md 2>&1
      tail -n 12 $OUT/gemm_chain_f32.md ;;
    shapes)
      timeout 300 python scripts/gemm_step_shapes.py f32 32 > $OUT/gemm_shapes_f32_b32.txt 2>&1
      timeout 300 python scripts/gemm_step_shapes.py f32 64 > $OUT/gemm_shapes_f32_b64.txt 2>&1
      timeout 300 python scripts/gemm_step_shapes.py bf16 64 > $OUT/gemm_shapes_bf16_b64.txt 2>&1 ;;
    *) echo "running: $ST"; eval "$ST" ;;
  esac
done
