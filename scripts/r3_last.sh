#!/bin/bash
# last GPU call of round 3: attention tile-skip -- op tests, model / decode parity, one A/B pair (two builds, one box)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/r3last; mkdir -p $OUT; cd $REPO
timeout 90 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k attention -p no:cacheprovider > $OUT/t1.log 2>&1; echo "t1 rc=$?" >> $OUT/t1.log; tail -n 2 $OUT/t1.log
timeout 150 python -m pytest tests/test_model_gpu.py tests/test_decode_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/t2.log 2>&1; echo "t2 rc=$?" >> $OUT/t2.log; tail -n 2 $OUT/t2.log
one() { env $1 timeout 100 python bench.py --batch 32 --no-decode --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_time_ms_per_step']; print(round(d['value']), 'commits/s', round(d['ms_per_step'],3), 'ms', 'attn', round(k['attention'],3))"; }
for V in "FIRA_HIP_LIB=$REPO/fira_icse_amd/libfira_hip_prev.so" "FIRA_X=1"; do echo -n "${V##*/}: "; one "$V"; done 2>&1 | tee $OUT/ab.txt
