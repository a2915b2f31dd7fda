#!/bin/bash
# Copy the summaries of a round-3 GPU session (gpurun_out/<tag>, written by scripts/r3_gpu.sh) into profiles/ (tracked).
TAG=${1:?tag}
S=gpurun_out/$TAG
cp_if() { [ -s "$1" ] && cp "$1" "$2" && echo "  $2"; }
cp_if $S/bench_f32.json profiles/r3_bench_f32.json
cp_if $S/bench_bf16.json profiles/r3_bench_bf16.json
cp_if $S/kernel_stats_f32.md profiles/r3_kernel_stats_f32.md
cp_if $S/kernel_stats_bf16.md profiles/r3_kernel_stats_bf16.md
cp_if $S/timeline_f32.md profiles/r3_timeline_f32.md
cp_if $S/kernel_stats_decode.md profiles/r3_kernel_stats_decode.md
cp_if $S/pmc_traffic_f32.md profiles/r3_pmc_traffic_f32.md
cp_if $S/pmc_traffic_bf16.md profiles/r3_pmc_traffic_bf16.md
cp_if $S/traffic.json profiles/traffic.json
cp_if $S/spmm_crossover.md profiles/r3_spmm_crossover.md
cp_if $S/gemm_chain_f32.md profiles/r3_gemm_chain_f32.md
cp_if $S/gemm_chain_bf16.md profiles/r3_gemm_chain_bf16.md
cp_if $S/tests.log profiles/r3_gpu_tests.log
cp_if $S/event_cost.txt profiles/r3_event_cost.txt
