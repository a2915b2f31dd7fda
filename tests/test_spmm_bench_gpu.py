"""BASELINE config 5 (dense synthetic graphs) through the aggregation kernel: parity at full size via linearity and
symmetry (size-independent properties), plus the realistic-density batch against a dense bmm."""
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
    # one graph against the dense product
    b0 = 17
    lo, hi = int(rowptr[b0 * N]), int(rowptr[(b0 + 1) * N])
    dense = torch.zeros(N, N, dtype=torch.float64)
    rows = np.repeat(np.arange(N), np.diff(rowptr[b0 * N:(b0 + 1) * N + 1]))
    dense[rows, col[lo:hi] - b0 * N] = torch.from_numpy(val[lo:hi]).double()
    ref = dense.cuda() @ X[b0 * N:(b0 + 1) * N].double()
    err = float((AX[b0 * N:(b0 + 1) * N].double() - ref).norm() / ref.norm())
    assert err < (1e-6 if variant != 4 else 6e-3)
    if variant == 4:                                             # exact up to fp32 accumulation on the bf16-rounded operands
        ref_b = dense.float().bfloat16().double().cuda() @ X[b0 * N:(b0 + 1) * N].bfloat16().double()
        assert float((AX[b0 * N:(b0 + 1) * N].double() - ref_b).norm() / ref_b.norm()) < 2e-6
