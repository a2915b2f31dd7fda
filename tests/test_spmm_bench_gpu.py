"""BASELINE config 5 (dense synthetic graphs) through the four aggregation kernels at full size: all 128 graphs against the
dense fp64 bmm, plus linearity and symmetry (size-independent properties)."""
import numpy as np
import pytest
import torch

import util  # noqa: F401
from fira_icse_amd import graphs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", [1, 2, 3, 4])
def test_config5_full_size_properties(variant):
    from fira_icse_amd import ops
    B, N = 128, 512
    rowptr, col, val = graphs.dense_stress_batch(B, N, n_types=4, edges_per_type=8192, seed=0)
    rp, c, v = (torch.from_numpy(a).cuda() for a in (rowptr, col, val))
    g = torch.Generator().manual_seed(0)
    X = torch.randn(B * N, 256, generator=g).cuda()
    Y = torch.randn(B * N, 256, generator=g).cuda()
    AX = ops.csr_spmm(rp, c, v, X, graph_rows=N, variant=variant)
    AY = ops.csr_spmm(rp, c, v, Y, graph_rows=N, variant=variant)
    # linearity: A(2X - 3Y) == 2AX - 3AY
    lin = ops.csr_spmm(rp, c, v, (2 * X - 3 * Y).contiguous(), graph_rows=N, variant=variant)
    # (bf16 operands: 2X - 3Y is rounded on its own, so linearity holds to bf16 resolution only)
    assert float((lin - (2 * AX - 3 * AY)).abs().max()) < (2e-4 if variant != 4 else 0.15)
    # symmetry of A_hat: <Y, A X> == <A Y, X>   (why the backward reuses the forward kernel)
    a, b = float((Y.double() * AX.double()).sum()), float((AY.double() * X.double()).sum())
    assert abs(a - b) / abs(a) < (1e-6 if variant != 4 else 5e-2)
    # EVERY graph of the batch against the dense fp64 product (torch.bmm(edge, x), gnn_transformer.py:80)
    rows = torch.from_numpy(np.repeat(np.arange(B * N), np.diff(rowptr))).cuda()
    cols = c.long()
    dense = torch.zeros(B, N, N, dtype=torch.float64, device="cuda")
    dense.view(B * N, N).index_put_((rows, cols - (rows // N) * N), v.double(), accumulate=True)
    Xg = X.view(B, N, 256)
    ref = torch.bmm(dense, Xg.double())
    err = (AX.view(B, N, 256).double() - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)      # per graph
    assert float(err.max()) < (1e-6 if variant != 4 else 6e-3), (int(err.argmax()), float(err.max()))
    if variant == 4:                                             # exact up to fp32 accumulation on the bf16-rounded operands
        ref_b = torch.bmm(dense.float().bfloat16().double(), Xg.bfloat16().double())
        err_b = (AX.view(B, N, 256).double() - ref_b).flatten(1).norm(dim=1) / ref_b.flatten(1).norm(dim=1)
        assert float(err_b.max()) < 2e-6, (int(err_b.argmax()), float(err_b.max()))
