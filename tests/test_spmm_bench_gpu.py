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


@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_config5_full_gcn_layer_fwd_bwd(dtype):
    """SURVEY.md §8(d) row 5's second unit: one full GCN layer (gnn_transformer.py:74-86, folded form) forward + backward on
    the config-5 graphs -- 128 x 512 nodes, ~59 k entries per graph, i.e. ~116 entries per row: the fused kernels' gather
    runs its tail path (> 16 entries) for every row.  Every graph against the fp64 statement (dense bmm per graph) with the
    engine's own dropout mask; dtype 1 = bf16 operands of the product against the fp64 product of the ROUNDED operands;
    dtype 2 = FIRA_F32X3 (three bf16 terms per operand: the fp32 tolerance)."""
    import torch.nn.functional as F
    from fira_icse_amd import ops
    B, N, D = 128, 512, 256
    rowptr, col, val = graphs.dense_stress_batch(B, N, n_types=4, edges_per_type=8192, seed=0)
    assert 55000 < len(col) / B < 66049
    rp, c, v = (torch.from_numpy(a).cuda() for a in (rowptr, col, val))
    n = B * N
    g = torch.Generator().manual_seed(5)
    rn = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).cuda()
    X = rn(n, D)
    W21, b2, c21 = rn(D, D, scale=0.06), rn(D, scale=0.1), rn(D, scale=0.1)
    gamma, beta = 1 + rn(D, scale=0.1), rn(D, scale=0.1)
    seed, site, p = 77, 21, 0.2
    summ, y, stats, rs = ops.gcn_layer_fwd(rp, c, v, X, W21.t().contiguous(), b2, c21, gamma, beta, dropout=p, seed=seed,
                                           site=site, dtype=dtype)
    rows = torch.from_numpy(np.repeat(np.arange(n), np.diff(rowptr))).cuda()
    dense = torch.zeros(B, N, N, dtype=torch.float64, device="cuda")
    dense.view(n, N).index_put_((rows, c.long() - (rows // N) * N), v.double(), accumulate=True)
    r16 = (lambda t: t.float().bfloat16().double()) if dtype == 1 else (lambda t: t.double())
    U = torch.bmm(dense, X.view(B, N, D).double()).view(n, D)
    rowsum = dense.sum(2).view(n, 1)
    pre = r16(U) @ r16(W21).t() + b2.double() + rowsum * c21.double()
    mask = ops.dropout_mask(seed, site, n * D, p).view(n, D).double()
    ref_sum = pre * mask + X.double()
    ref_y = F.layer_norm(ref_sum, (D,), gamma.double(), beta.double(), 1e-5)
    per_graph = lambda a, b: float(((a.double() - b).view(B, -1).norm(dim=1) / b.view(B, -1).norm(dim=1)).max())
    tol = 3e-5 if dtype == 1 else 2e-6
    assert per_graph(rs.view(n, 1), rowsum) < 1e-6
    assert per_graph(summ, ref_sum) < tol and per_graph(y, ref_y) < 5 * tol
    # backward: V = A_hat dY, dX += V W21 (the identity the engine uses: A_hat (dY W21) = (A_hat dY) W21)
    dY, dX0 = rn(n, D), rn(n, D)
    dX = dX0.clone()
    V = ops.gcn_layer_bwd(rp, c, v, dY, W21, dX, dtype=dtype)
    refV = torch.bmm(dense, dY.view(B, N, D).double()).view(n, D)
    assert per_graph(V, refV) < 1e-6
    assert per_graph(dX, dX0.double() + r16(refV) @ r16(W21)) < tol
