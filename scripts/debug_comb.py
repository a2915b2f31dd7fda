import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import ops
torch.manual_seed(0)
n = 840
Xc = torch.randn(n, 256, device="cuda")
Wqk, bqk = torch.randn(512, 256, device="cuda") * 0.08, torch.randn(512, device="cuda") * 0.1
Wo, bo = torch.randn(256, 256, device="cuda") * 0.08, torch.randn(256, device="cuda") * 0.1
vtab = torch.randn(4, 256, device="cuda")
mark = torch.randint(0, 4, (n,), device="cuda", dtype=torch.int32)
gamma, beta = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
outs = ops.combination_block_fwd(Xc, Wqk, bqk, Wo, bo, vtab, mark, gamma, beta)
outs2 = ops.combination_block_fwd(Xc, Wqk, bqk, Wo, bo, vtab, mark, gamma, beta)
print("run-to-run bitwise:", [bool(torch.equal(a, b)) for a, b in zip(outs, outs2)])
perm = torch.randperm(n, device="cuda")
sub = perm[:411]
outs3 = ops.combination_block_fwd(Xc[sub].contiguous(), Wqk, bqk, Wo, bo, vtab, mark[sub].contiguous(), gamma, beta)
names = ["qk", "c", "sum", "y", "stats"]
for nm, a, b in zip(names, outs, outs3):
    d = (a[sub] - b).abs().max().item()
    print(nm, "subset/permuted vs full: max abs diff", d, "bitwise", bool(torch.equal(a[sub], b)))
