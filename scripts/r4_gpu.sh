#!/bin/bash
# Round-4 GPU session (through gpurun): bash scripts/r4_gpu.sh <tag> <stages...>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4a}; shift
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
PREV=$REPO/fira_icse_amd/libfira_hip_prev.so
one() { env $1 timeout 500 python bench.py $2 --no-decode --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_time_ms_per_step']; print(round(d['value']), 'commits/s', round(d['ms_per_step'],3), 'ms', 'attn', round(k['attention'],3), 'gemm', round(k['gemm'],3), 'rowops', round(k['rowops'],3), 'spmm', round(k['spmm'],3), 'host', round(d['host_enqueue_ms_per_step'],2))"; }
for ST in "$@"; do
  case $ST in
    newtests)
      timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_spmm_bench_gpu.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -k "attention or config5" > $OUT/newtests.log 2>&1
      echo "newtests rc=$?"; tail -n 15 $OUT/newtests.log ;;
    modeltests)
      timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_decode_gpu.py tests/test_edge_cases_gpu.py tests/test_dropout_gpu.py tests/test_large_gpu.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -x > $OUT/modeltests.log 2>&1
      echo "modeltests rc=$?"; tail -n 15 $OUT/modeltests.log ;;
    tests)
      timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 300 > $OUT/tests.log 2>&1
      echo "tests rc=$?" >> $OUT/tests.log; tail -n 25 $OUT/tests.log ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 3 ;;
    ab3)   # two BUILDS on one box: libfira_hip_prev.so (the previous commit's build) vs the tree's library
      for i in 1 2 3; do
        for V in "FIRA_HIP_LIB=$PREV" "FIRA_X=1"; do
          echo -n "${V##*/} f32 b32: "; one "$V" "--batch 32"
          [ $i = 1 ] && { echo -n "${V##*/} f32 b64: "; one "$V" "--batch 64"; echo -n "${V##*/} bf16 b64: "; one "$V" "--dtype bf16 --batch 64"; }
        done
      done 2>&1 | tee $OUT/ab3.txt ;;
    abdec)   # decode loop: previous build vs the tree's library, alternating
      for i in 1 2; do
        for V in "FIRA_HIP_LIB=$PREV" "FIRA_X=1"; do echo -n "${V##*/} decode: "; env $V timeout 200 python scripts/decode_only.py 2>/dev/null | tail -n 2 | tr "\n" ";"; echo; done
      done 2>&1 | tee $OUT/abdec.txt ;;
    probe)
      for L in "3136 dense" "3136 ragged" "3072 ragged" "512 ragged" "256 ragged"; do
        timeout 120 python scripts/attn_probe.py 32 $L 2>&1 | grep "us per launch"
      done | tee $OUT/attn_probe.txt
      timeout 120 python scripts/attn_probe.py 64 3136 ragged 2>&1 | grep "us per launch" | tee -a $OUT/attn_probe.txt
      timeout 120 python scripts/attn_probe.py 32 768 self 2>&1 | grep "us per launch" | tee -a $OUT/attn_probe.txt ;;
    probeprev)   # the same probe on the previous build
      for L in "32 3136 ragged" "64 3136 ragged" "32 768 self"; do
        FIRA_HIP_LIB=$PREV timeout 120 python scripts/attn_probe.py $L 2>&1 | grep "us per launch"
      done | tee $OUT/attn_probe_prev.txt ;;
    nwprobe)
      for NW in 12 6 4; do
        echo "FIRA_ATTN_NW=$NW"; FIRA_ATTN_NW=$NW timeout 120 python scripts/attn_probe.py 32 3136 ragged 2>&1 | grep "us per launch"
        FIRA_ATTN_NW=$NW timeout 120 python scripts/attn_probe.py 64 3136 ragged 2>&1 | grep "us per launch"
      done | tee $OUT/attn_nw_probe.txt ;;
    gcnprobe)
      timeout 200 python scripts/gcn_probe.py 32 0 2>&1 | tail -n 6 | tee $OUT/gcn_probe.txt
      timeout 200 python scripts/gcn_probe.py 64 1 2>&1 | tail -n 6 | tee -a $OUT/gcn_probe.txt
      cd /tmp; mkdir -p $OUT/prof_gcn
      timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_gcn -o g -- python $REPO/scripts/gcn_probe.py 32 0 > $OUT/prof_gcn/run.log 2>&1
      python $REPO/scripts/rocpd_stats.py $(find $OUT/prof_gcn -name "*results.db" | head -1) 1 > $OUT/gcn_kernel_stats.md 2>&1
      find $OUT/prof_gcn -name "*.db" -delete
      grep "gcn_fused\|spmm_rowwave\|gemm_f32_kernel" $OUT/gcn_kernel_stats.md | cut -c1-200
      for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
        tag=$(echo $grp | cut -d' ' -f1)
        timeout 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_gcn_$tag -o a -- python $REPO/scripts/gcn_probe.py 32 0 > $OUT/pmc_gcn_$tag.log 2>&1
      done
      for f in $(find $OUT -name "*counter_collection.csv"); do python $REPO/scripts/pmc_summary.py $f gcn_fused; done | tee $OUT/gcn_pmc.txt
      find $OUT -name "*.csv" -delete
      cd $REPO ;;
    gcntests)
      timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -k "gcn_layer" > $OUT/gcntests.log 2>&1
      echo "gcntests rc=$?"; tail -n 12 $OUT/gcntests.log ;;
    abenv16)
      IFS='|' read -ra VARS <<< "$ABENV"
      for i in 1 2; do
        for V in "${VARS[@]}"; do
          echo -n "$V bf16 b64: "; one "$V" "--dtype bf16 --batch 64"
        done
      done 2>&1 | tee $OUT/abenv16.txt ;;
    abenv)   # environment switches of ONE build, alternating: ABENV="A=1|B=2|..." 
      IFS='|' read -ra VARS <<< "$ABENV"
      for i in 1 2 3; do
        for V in "${VARS[@]}"; do
          echo -n "$V f32 b${ABBATCH:-32}: "; one "$V" "--batch ${ABBATCH:-32}"
          [ $i = 1 ] && [ -z "$ABBATCH" ] && { echo -n "$V f32 b64: "; one "$V" "--batch 64"; }
        done
      done 2>&1 | tee $OUT/abenv.txt ;;
    adopt)   # make this session's counters / traces the ones bench.py quotes (profiles/traffic.json, r4_kernel_classes.json)
      cp $OUT/traffic.json $REPO/profiles/traffic.json; cp $OUT/kernel_classes.json $REPO/profiles/r4_kernel_classes.json; echo adopted ;;
    bench)
      timeout 900 python bench.py > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "bench f32 rc=$?"; head -c 1500 $OUT/bench_f32.json; echo; tail -n 5 $OUT/bench_f32.err ;;
    bench16)
      timeout 600 python bench.py --dtype bf16 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; echo "bench bf16 rc=$?"; head -c 600 $OUT/bench_bf16.json; echo ;;
    prof|prof16)
      DT=f32; [ $ST = prof16 ] && DT=bf16
      cd /tmp; mkdir -p $OUT/prof_$DT
      timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$DT -o step -- python $REPO/bench.py --dtype $DT --steps 10 --warmup 2 --no-decode --no-cpu-baseline --no-extras > $OUT/prof_$DT/bench.log 2>&1
      VAL=$(grep -o '"value": [0-9.]*' $OUT/prof_$DT/bench.log | head -1 | cut -d' ' -f2)
      BS=32; [ $DT = bf16 ] && BS=64
      ROCPD_CLASSES_JSON=$OUT/kernel_classes.json ROCPD_DTYPE=$DT ROCPD_BATCH=$BS ROCPD_COMMIT=${R4_COMMIT:-unknown} ROCPD_VALUE=$VAL \
        ROCPD_SOURCE="rocprofv3 --kernel-trace --stats -- python bench.py --dtype $DT --steps 10 --warmup 2 --no-decode --no-cpu-baseline --no-extras (15 steps in the trace; under the profiler)" \
        python $REPO/scripts/rocpd_stats.py $(find $OUT/prof_$DT -name "*results.db" | head -1) 15 $OUT/timeline_$DT.md > $OUT/kernel_stats_$DT.md 2>&1
      find $OUT/prof_$DT -name "*.db" -delete
      cd $REPO; head -n 60 $OUT/kernel_stats_$DT.md ;;
    profdec)
      cd /tmp; mkdir -p $OUT/prof_dec
      timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_dec -o dec -- python $REPO/scripts/decode_only.py > $OUT/prof_dec/run.log 2>&1
      python $REPO/scripts/rocpd_stats.py $(find $OUT/prof_dec -name "*results.db" | head -1) 1 > $OUT/kernel_stats_decode.md 2>&1
      find $OUT/prof_dec -name "*.db" -delete
      cd $REPO; head -n 40 $OUT/kernel_stats_decode.md ;;
    pmc)
      for DT in f32 bf16; do
        bash scripts/pmc_traffic.sh $DT > /dev/null 2>&1
        ROCPD_COMMIT=${R4_COMMIT:-unknown} python scripts/pmc_traffic_summary.py gpurun_out/pmc_traffic_$DT $OUT/pmc_traffic_$DT.md $OUT/traffic.json $DT; echo "pmc $DT rc=$?"
        find gpurun_out/pmc_traffic_$DT -name "*.csv" -delete
      done
      cat $OUT/traffic.json | head -40 ;;
    *) echo "running: $ST"; eval "$ST" ;;
  esac
done
