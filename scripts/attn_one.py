"""A few cross-attention forward/backward launches at the training shape (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import ops
B, Tq, Tk = 32, 30, 370
q, do = torch.randn(B, Tq, 256, device="cuda"), torch.randn(B, Tq, 256, device="cuda")
kv = torch.randn(B, Tk, 3072, device="cuda")
valid = (torch.rand(B, Tk, device="cuda") > 0.3).to(torch.int32)
for _ in range(5):
    o = ops.attention_fwd(q, kv[:, :, :256], kv[:, :, 256:512], valid)
    ops.attention_bwd(q, kv[:, :, :256], kv[:, :, 256:512], valid, o, do)
torch.cuda.synchronize()
