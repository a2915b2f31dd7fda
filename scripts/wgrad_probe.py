"""Stand-alone timing of the weight-gradient products of the training step: the panel kernel (gemm_wgrad_panel.hip) against the
tiled kernels, fp32 (three-term split vs fp32 MFMA) and bf16.    python scripts/wgrad_probe.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    Bc = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    nodes, code, td, mem, R = 313 * Bc, 110 * Bc, 17 * Bc, 172 * Bc, 13 * Bc
    shapes = [("GCN dW21", 256, 256, nodes, 256), ("Comb q|k", 512, 256, code, 512), ("decoder qkv", 768, 256, td, 768),
              ("FFN w1", 1024, 256, td, 1024), ("FFN w2", 256, 1024, td, 256), ("K|V stacked", 3072, 256, mem, 3136),
              ("vocabulary", 24650, 256, R, 24704)]
    for name, M, N, K, lda in shapes:
        A = torch.randn(K, lda, device="cuda")[:, :M]
        B = torch.randn(K, N, device="cuda")
        C = torch.zeros(M, N, device="cuda")
        cs = torch.zeros(M, device="cuda")
        flop = 2.0 * M * N * K
        by = 4.0 * (M * K + N * K + M * N)
        row = "%-12s M %5d N %4d K %6d: " % (name, M, N, K)
        for label, fn in (("panel f32x3", lambda: ops.gemm_wgrad_panel(A, B, C, cs, dtype=0)),
                          ("tiled f32", lambda: ops.gemm(A, B, transA=True, transB=False, out=C, accumulate=True, splitk=0)),
                          ("panel bf16", lambda: ops.gemm_wgrad_panel(A, B, C, cs, dtype=1)),
                          ("tiled bf16", lambda: ops.gemm(A, B, transA=True, transB=False, out=C, accumulate=True, splitk=0, dtype="bf16"))):
            us = timeit(fn)
            row += "%s %6.1f us (%5.1f TF/s, %4.2f TB/s) | " % (label, us, flop / us / 1e6, by / us / 1e6)
        print(row, flush=True)


if __name__ == "__main__":
    main()
