"""Run one GEMM shape a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py M N K tA tB [splitk]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import ops
M, N, K, tA, tB = (int(x) for x in sys.argv[1:6])
sk = int(sys.argv[6]) if len(sys.argv) > 6 else 1
A = torch.randn((K, M) if tA else (M, K), device="cuda")
B = torch.randn((N, K) if tB else (K, N), device="cuda")
C = torch.zeros(M, N, device="cuda")
for _ in range(10):
    ops.gemm(A, B, transA=bool(tA), transB=bool(tB), out=C, accumulate=sk > 1, splitk=sk)
torch.cuda.synchronize()
sys.stdout.flush()

