import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import ops
torch.manual_seed(0)
N = 64
for variant in (3, 4):
    # identity
    rowptr = torch.arange(N + 1, dtype=torch.int32, device="cuda")
    col = torch.arange(N, dtype=torch.int32, device="cuda")
    val = torch.ones(N, device="cuda")
    X = torch.randn(N, 256, device="cuda")
    Y = ops.csr_spmm(rowptr, col, val, X, graph_rows=N, variant=variant)
    print("variant", variant, "identity max err", float((Y - X).abs().max()))
    bad = ((Y - X).abs() > 0.05).nonzero()
    print(" bad count", bad.shape[0], bad[:10].tolist())
    # single entry A[r0, c0] = 1 -> Y[r0] = X[c0]
    for (r0, c0) in ((3, 17), (40, 5), (63, 63), (1, 33)):
        rp = torch.zeros(N + 1, dtype=torch.int32); rp[r0 + 1:] = 1
        Y = ops.csr_spmm(rp.cuda(), torch.tensor([c0], dtype=torch.int32).cuda(), torch.ones(1).cuda(), X, graph_rows=N, variant=variant)
        nzr = (Y.abs().sum(1) > 0).nonzero().flatten().tolist()
        src = None
        if nzr:
            d = (X - Y[nzr[0]]).abs().sum(1)
            src = int(d.argmin()), float(d.min())
        print("  A[%d,%d]=1 -> nonzero rows %s, row equals X[%s]" % (r0, c0, nzr[:8], src))
