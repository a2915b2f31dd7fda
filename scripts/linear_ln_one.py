"""Time the fused linear+LayerNorm launch against the two-kernel composition at the step's shapes (GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fira_icse_amd import ops

def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for M, K in [(960, 256), (960, 1024), (1920, 256), (1920, 1024), (6000, 256), (12000, 256), (64, 256), (192, 1024)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(256, K, device="cuda") * K ** -0.5
    b = torch.randn(256, device="cuda"); res = torch.randn(M, 256, device="cuda")
    g = torch.ones(256, device="cuda"); be = torch.zeros(256, device="cuda")
    fused = t(lambda: ops.linear_layernorm_fwd(x, w, b, res, g, be, 0.1, 1, 1))
    def two():
        lin = ops.gemm(x, w, bias=b, transB=True)
        ops.add_layernorm_fwd(lin, res, g, be, 0.1, 1, 1)
    print(f"M={M:6d} K={K:5d} fused {fused:7.1f} us   gemm+ln {t(two):7.1f} us (incl. torch.empty + ctypes overhead in both)")
