"""Dispatch floor: back-to-back dependent launches of a 1-thread kernel / a small memset on one stream."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import ops
from scripts.bench_kernels import timeit

n = torch.ones(1, dtype=torch.int32, device="cuda")
out = torch.zeros(1, device="cuda")
z = torch.zeros(1024, device="cuda")
print("1-thread kernel: %.2f us per launch" % (timeit(lambda: ops.inv_count(n, out), iters=200) * 1e6))
print("4 KB memset:     %.2f us per launch" % (timeit(lambda: z.zero_(), iters=200) * 1e6))
a = torch.randn(960, 256, device="cuda"); w = torch.randn(256, 256, device="cuda"); c = torch.empty(960, 256, device="cuda")
print("gemm 960x256x256: %.2f us" % (timeit(lambda: ops.gemm(a, w, out=c, splitk=0), iters=200) * 1e6))
