"""Time the bf16 weight-shadow products (forward / data gradient through the k-contiguous bf16 B operand) at the batch-64
step's shapes.  Run twice to compare kernels: FIRA_PANEL_GEMM=0 (tiled kernel) vs 1 (A-stationary panel kernel)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import ops
from scripts.bench_kernels import timeit

SHAPES = [("enc fc", 24000, 256, 256, 24), ("enc qk fwd", 10000, 512, 256, 6), ("enc o", 10000, 256, 256, 12),
          ("kv_all fwd", 17000, 3072, 256, 1), ("src fwd", 17000, 256, 256, 1), ("dec qkv fwd", 1920, 768, 256, 6),
          ("dec ffn1 fwd", 1920, 1024, 256, 12), ("dec 256", 1920, 256, 256, 38), ("out_fc fwd", 1400, 24650, 256, 1),
          ("enc fc b32", 12000, 256, 256, 0), ("kv_all b32", 8500, 3072, 256, 0)]
print("FIRA_PANEL_GEMM=%s" % os.environ.get("FIRA_PANEL_GEMM", "(default)"))
tot = 0.0
for name, M, N, K, cnt in SHAPES:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
    wb, _ = ops.weight_shadow(W)
    C = torch.empty(M, N, device="cuda")
    t = timeit(lambda: ops.gemm_wb(A, wb, out=C), iters=100)
    byts = 4.0 * M * K + 2.0 * N * K + 4.0 * M * N
    tot += t * cnt
    print("%-14s %6d %6d %5d  x%-2d %8.1f us  %6.1f TF  %5.2f TB/s (algorithmic)" % (name, M, N, K, cnt, t * 1e6, 2.0 * M * N * K / t / 1e12, byts / t / 1e12))
print("sum over the step's calls: %.0f us" % (tot * 1e6))
