"""Per-kernel parity: every HIP op (through the C ABI) against a plain PyTorch fp32/fp64 reference of the same op.

Tolerances are stated per test: fp32 kernels with a different summation order than the reference agree to a few
1e-6 relative (SURVEY.md §8c measured 2.5e-7 between fp32 and fp64 of the reference itself).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import util  # noqa: F401

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def randn(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(960, 256, 256), (20800, 256, 256), (130, 70, 50), (4, 1536, 256), (64, 24650, 256),
                                   (333, 256, 24650), (960, 1024, 256), (960, 2, 256)])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn", "nt-t1", "nn-t2", "tn-t3", "nt-t3", "tn-t2"])
def test_gemm_layouts(M, N, K, layout):
    from fira_icse_amd import ops
    if M * N * K > 3e9:
        pytest.skip("too large")
    lds = int(layout[-1]) if "-t" in layout else 0     # force each tile shape (1 128x128, 2 64x128, 3 64x64)
    tA, tB = {"nt": (False, True), "nn": (False, False), "tn": (True, False)}[layout[:2]]
    A = randn(*((K, M) if tA else (M, K)), seed=1)
    B = randn(*((N, K) if tB else (K, N)), seed=2)       # asymmetric operands: a swapped tile cannot pass
    bias = randn(N, seed=3)
    ref = ((A.t() if tA else A).double() @ (B.t() if tB else B).double()) + bias.double()
    tol = 2e-6 if K <= 4096 else 6e-6             # a k-ordered fp32 chain: error grows ~ sqrt(K) * 2^-24
    out = ops.gemm(A, B, transA=tA, transB=tB, bias=bias, tile=lds)
    assert rel_err(out, ref) < tol
    out = ops.gemm(A, B, transA=tA, transB=tB, bias=bias, relu=True, tile=lds)
    assert rel_err(out, ref.clamp_min(0)) < tol
    if not tA:
        C0 = randn(M, N, seed=9)
        out = ops.gemm(A, B, transA=tA, transB=tB, out=C0.clone(), accumulate=True, tile=lds)
        assert rel_err(out, ref - bias.double() + C0.double()) < tol


@pytest.mark.parametrize("M,N,K", [(64, 256, 1024), (530, 256, 1024), (530, 256, 768), (1000, 256, 512), (2000, 256, 1024),
                                   (33, 64, 1024), (45, 96, 768)])
@pytest.mark.parametrize("layout", ["nt", "nn"])
def test_gemm_long_reduction_on_a_small_grid(M, N, K, layout):
    """The 32x32 tile kernel with 8 / 12 / 16 waves per workgroup (two K chunks per wave: gemm_small.hip, round 5) -- the
    FFN-shaped products of the decoder chain and of the decode step; FIRA_TILE32_WAVES=0 restores four waves."""
    from fira_icse_amd import ops
    tB = layout == "nt"
    A = randn(M, K, seed=11)
    B = randn(*((N, K) if tB else (K, N)), seed=12)
    bias = randn(N, seed=13)
    ref = A.double() @ (B.t() if tB else B).double() + bias.double()
    out = ops.gemm(A, B, transB=tB, bias=bias)
    assert rel_err(out, ref) < 2e-6
    out = ops.gemm(A, B, transB=tB, bias=bias, relu=True)
    assert rel_err(out, ref.clamp_min(0)) < 2e-6
    C0 = randn(M, N, seed=14)
    out = ops.gemm(A, B, transB=tB, out=C0.clone(), accumulate=True)
    assert rel_err(out, ref - bias.double() + C0.double()) < 2e-6


def test_gemm_accumulate_and_splitk():
    from fira_icse_amd import ops
    M, N, K = 256, 256, 20800                      # the weight-gradient shape of a GCN layer at batch 32
    A, B = randn(K, M, seed=4), randn(K, N, seed=5)
    C0 = randn(M, N, seed=6)
    ref = C0.double() + A.t().double() @ B.double()
    for sk in (1, 8, 37):
        C = C0.clone()
        ops.gemm(A, B, transA=True, transB=False, out=C, accumulate=True, splitk=sk)
        assert rel_err(C, ref) < 6e-6, sk


def test_gemm_strided_output_and_padded_ld():
    from fira_icse_amd import ops
    # decode writes K/V rows into the cache with a row stride of T*256; logits rows are padded to 24704
    X, W = randn(64, 256, seed=7), randn(256, 256, seed=8)
    cache = torch.zeros(64, 30, 256, device=DEV)
    ops.gemm(X, W, out=cache[:, 5, :])
    assert rel_err(cache[:, 5, :], X.double() @ W.double().t()) < 2e-6
    assert float(cache[:, 4, :].abs().max()) == 0 and float(cache[:, 6, :].abs().max()) == 0
    dl = torch.zeros(100, 24704, device=DEV)
    dl[:, :24650] = randn(100, 24650, seed=9)
    Wo = randn(24650, 256, seed=10)
    out = ops.gemm(dl[:, :24650], Wo, transB=False)
    assert rel_err(out, dl[:, :24650].double() @ Wo.double()) < 6e-6


# ------------------------------------------------------------------------------------------------ SpMM
def random_graph_batch(B, N, nnz_per_graph, seed):
    rng = np.random.default_rng(seed)
    rowptr, col, val = [0], [], []
    dense = np.zeros((B, N, N), dtype=np.float32)
    for b in range(B):
        pairs = set((i, i) for i in range(N))
        while len(pairs) < nnz_per_graph:
            i, j = rng.integers(0, N, 2)
            pairs.add((int(i), int(j))); pairs.add((int(j), int(i)))
        pairs = sorted(pairs)
        for r in range(N):
            cs = [c for (rr, c) in pairs if rr == r]
            for c in cs:
                v = np.float32(rng.uniform(0.05, 1.0))
                dense[b, r, c] = v
                col.append(b * N + c); val.append(v)
            rowptr.append(len(col))
    t = lambda a, dt: torch.tensor(np.array(a, dtype=dt), device=DEV)
    return t(rowptr, np.int32), t(col, np.int32), t(val, np.float32), torch.tensor(dense, device=DEV)


@pytest.mark.parametrize("variant,N,nnz", [(1, 650, 2500), (1, 97, 4000), (1, 33, 1000), (5, 650, 2500), (5, 97, 4000), (2, 512, 9000), (2, 500, 3000), (2, 64, 3000)])
def test_csr_spmm_vs_dense_bmm(variant, N, nnz):
    from fira_icse_amd import ops
    B = 3
    rowptr, col, val, dense = random_graph_batch(B, N, nnz, seed=N)
    X = randn(B * N, 256, seed=11)
    ref = torch.bmm(dense.double(), X.view(B, N, 256).double()).view(B * N, 256)
    Y = ops.csr_spmm(rowptr, col, val, X, graph_rows=N, variant=variant)
    assert rel_err(Y, ref) < 1e-6


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
@pytest.mark.parametrize("N,nnz,p", [(650, 2500, 0.0), (97, 4000, 0.2), (33, 200, 0.2)])
def test_gcn_layer_fused_fwd_bwd(N, nnz, p, dtype):
    """fira_gcn_layer_{fwd,bwd} (gcn_fused.hip): one launch per direction for the folded GCN layer
    LN(dropout((A_hat X) W21^T + b2 + (A_hat 1) c21^T) + X) of gnn_transformer.py:74-86, against the fp64 statement of the
    same formula with the engine's own dropout mask; graphs with hub rows (97 nodes x 4000 entries: > 16 entries per row
    -> the tail path of the gather), ragged last row block (B*N not a multiple of 32).  Backward: V = A_hat dY and
    dX += V W21 against autograd of the aggregation + product.  dtype 1: bf16 operands of the product (fp32 gather,
    accumulation, LayerNorm) vs the fp64 product of the ROUNDED operands.  dtype 2 (FIRA_F32X3, what the engine's fp32 mode
    runs): three bf16 terms per operand on the bf16 matrix cores -- held to the fp32 launch's tolerance.  dtype 3
    (FIRA_BF16X1, what the engine's bf16 mode runs): one bf16 plane of the same kernels, held to dtype 1's tolerance."""
    from fira_icse_amd import ops
    B = 3
    rowptr, col, val, dense = random_graph_batch(B, N, nnz, seed=N + 1)
    n = B * N
    X = randn(n, 256, seed=1)
    W21, b2, c21 = randn(256, 256, seed=2, scale=0.06), randn(256, seed=3, scale=0.1), randn(256, seed=4, scale=0.1)
    gamma, beta = 1 + randn(256, seed=5, scale=0.1), randn(256, seed=6, scale=0.1)
    seed, site = 1234, 19
    summ, y, stats, rs = ops.gcn_layer_fwd(rowptr, col, val, X, W21.t().contiguous(), b2, c21, gamma, beta, dropout=p, seed=seed, site=site,
                                           dtype=dtype)
    A = torch.block_diag(*[dense[b] for b in range(B)]).double()
    U = A @ X.double()
    r16 = (lambda t: t.float().bfloat16().double()) if dtype in (1, 3) else (lambda t: t.double())
    pre = r16(U) @ r16(W21).t() + b2.double() + A.sum(1, keepdim=True) * c21.double()
    mask = ops.dropout_mask(seed, site, n * 256, p).view(n, 256).double() if p > 0 else 1.0
    ref_sum = pre * mask + X.double()
    ref_y = F.layer_norm(ref_sum, (256,), gamma.double(), beta.double(), 1e-5)
    # bf16: exact products of the rounded operands, fp32 accumulation; the kernel rounds the fp32 aggregate, the reference
    # the fp64 one -- a handful of elements land on the other side of a bf16 rounding boundary
    tol = 3e-5 if dtype in (1, 3) else 2e-6
    assert rel_err(rs, A.sum(1)) < 1e-6
    assert rel_err(summ, ref_sum) < tol and rel_err(y, ref_y) < 5 * tol
    mean, rstd = ref_sum.mean(1), 1.0 / torch.sqrt(ref_sum.var(1, unbiased=False) + 1e-5)
    assert rel_err(stats[:, 0], mean) < 1e-4 and rel_err(stats[:, 1], rstd) < 1e-5
    # backward
    dY = randn(n, 256, seed=7)
    dX0 = randn(n, 256, seed=8)
    dX = dX0.clone()
    V = ops.gcn_layer_bwd(rowptr, col, val, dY, W21, dX, dtype=dtype)
    refV = A @ dY.double()
    assert rel_err(V, refV) < 1e-6
    assert rel_err(dX, dX0.double() + r16(refV) @ r16(W21)) < tol


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
@pytest.mark.parametrize("n,p", [(3667, 0.0), (3667, 0.1), (530, 0.1), (37, 0.1), (6861, 0.1)])
def test_combination_block_fused_fwd(n, p, dtype):
    """fira_combination_block_fwd (comb_fused.hip): the Combination block of gnn_transformer.py:176-205 as one launch -- q|k
    projections, two-way gate (combination_layer.py:7-17), output projection, dropout, residual, LayerNorm -- against (a) the
    four launches it replaces with the SAME dropout masks (q|k bit-comparable up to fp32 re-association; c, sum, y, stats) and
    (b) the fp64 statement of the formulas with the engine's masks.  Row counts: batch 32 / 64 sizes (one and two tiles per
    CU, ragged last tile), fewer tiles than workgroups, a single partial pass.  dtype 1 (bf16 mode): the operands of the three
    products rounded to bf16, fp32 accumulation -- against the fp64 products of the ROUNDED operands."""
    from fira_icse_amd import ops
    Xc = randn(n, 256, seed=1)
    Wqk, bqk = randn(512, 256, seed=2, scale=0.08), randn(512, seed=3, scale=0.1)
    Wo, bo = randn(256, 256, seed=4, scale=0.08), randn(256, seed=5, scale=0.1)
    vtab = randn(4, 6 * 256, seed=6)[:, 256:512]                    # a slice of the stacked table: row stride 1536
    mark = torch.randint(0, 4, (n,), device=DEV, dtype=torch.int32)
    gamma, beta = 1 + randn(256, seed=7, scale=0.1), randn(256, seed=8, scale=0.1)
    rows = torch.randperm(2 * n, device=DEV)[:n].to(torch.int32)    # scattered output rows (the node buffer's code rows)
    ybuf = torch.full((2 * n, 256), float("nan"), device=DEV)
    seed, sg, so = 4321, 17, 18
    qk, c, summ, y, stats = ops.combination_block_fwd(Xc, Wqk, bqk, Wo, bo, vtab, mark, gamma, beta, dropout=p, seed=seed,
                                                      site_gate=sg, site_out=so, y=ybuf, y_rows=rows, dtype=dtype)
    # (b) fp64 reference with the engine's masks
    # (dtype 2 = FIRA_F32X3, three bf16 terms per operand, what the engine's fp32 mode runs: the fp32 tolerances)
    r16 = (lambda t: t.float().bfloat16().double()) if dtype in (1, 3) else (lambda t: t.double())
    q64 = r16(Xc) @ r16(Wqk[:256]).t() + bqk[:256].double()
    k64 = r16(Xc) @ r16(Wqk[256:]).t() + bqk[256:].double()
    v64 = vtab.double()[mark.long()]
    a, b = q64 * k64 / math.sqrt(32), q64 * v64 / math.sqrt(32)
    g = torch.softmax(torch.stack([a, b], -1), -1)
    mg = ops.dropout_mask(seed, sg, n * 256, p).view(n, 256).double() if p > 0 else 1.0
    mo = ops.dropout_mask(seed, so, n * 256, p).view(n, 256).double() if p > 0 else 1.0
    c64 = (g[..., 0] * k64 + g[..., 1] * v64) * mg
    # (bf16: the kernel rounds ITS fp32 c, the reference the fp64 one -- a few elements fall on the other side of a boundary)
    s64 = (r16(c64) @ r16(Wo).t() + bo.double()) * mo + Xc.double()
    y64 = F.layer_norm(s64, (256,), gamma.double(), beta.double(), 1e-5)
    t1, t2 = (2e-6, 5e-5) if dtype in (1, 3) else (2e-6, 5e-6)
    assert rel_err(qk[:, :256], q64) < t1 and rel_err(qk[:, 256:], k64) < t1
    assert rel_err(c, c64) < 5e-6 and rel_err(summ, s64) < t2
    assert rel_err(y[rows.long()], y64) < 2 * t2
    untouched = torch.ones(2 * n, dtype=torch.bool, device=DEV)
    untouched[rows.long()] = False
    assert bool(torch.isnan(y[untouched]).all())                     # only the listed rows are written
    assert rel_err(stats[:, 0], s64.mean(1)) < 1e-4 and rel_err(stats[:, 1], 1.0 / torch.sqrt(s64.var(1, unbiased=False) + 1e-5)) < 1e-5
    if dtype in (1, 3):
        return
    # (a) the separate launches, same masks
    qk2 = ops.gemm(Xc, Wqk, bias=bqk)
    c2 = ops.combination_fwd(qk2, vtab.contiguous(), mark, dropout=p, seed=seed, site=sg)
    lin = ops.gemm(c2, Wo, bias=bo)
    y2, sum2, st2 = ops.add_layernorm_fwd(lin, Xc, gamma, beta, dropout=p, seed=seed, site=so)
    assert rel_err(qk, qk2) < 2e-6 and rel_err(c, c2) < 5e-6 and rel_err(summ, sum2) < 5e-6 and rel_err(y[rows.long()], y2) < 1e-5
    if p > 0:                                                        # the same elements are dropped
        assert bool(((c == 0) == (c2 == 0)).all())


@pytest.mark.parametrize("dtype", [0, 1, 2, 3])
@pytest.mark.parametrize("n,p", [(3667, 0.0), (3667, 0.1), (37, 0.1), (6861, 0.1)])
def test_combination_block_fused_bwd(n, p, dtype):
    """fira_combination_block_bwd (comb_fused.hip): LayerNorm backward, data gradient through the output projection, gate
    backward and data gradient through q | k in one launch, against fp64 autograd of the block's own formulas with the engine's
    dropout masks: dG rows (in place at the listed rows, other rows untouched), dYc, dq|dk, dgamma, dbeta, dvtab.  bf16: the
    products on bf16-rounded operands (reference: the same rounding through a straight-through estimator)."""
    from fira_icse_amd import ops
    Xc = randn(n, 256, seed=1)
    Wqk, bqk = randn(512, 256, seed=2, scale=0.08), randn(512, seed=3, scale=0.1)
    Wo, bo = randn(256, 256, seed=4, scale=0.08), randn(256, seed=5, scale=0.1)
    vtab = randn(4, 256, seed=6)
    mark = torch.randint(0, 4, (n,), device=DEV, dtype=torch.int32)
    gamma, beta = 1 + randn(256, seed=7, scale=0.1), randn(256, seed=8, scale=0.1)
    rows = torch.randperm(2 * n, device=DEV)[:n].to(torch.int32)
    seed, sg, so = 4321, 17, 18
    qk, c, summ, y, stats = ops.combination_block_fwd(Xc, Wqk, bqk, Wo, bo, vtab, mark, gamma, beta, dropout=p, seed=seed,
                                                      site_gate=sg, site_out=so, dtype=dtype)
    dG0 = randn(2 * n, 256, seed=9)
    dG = dG0.clone()
    dYc, dqk, dgamma, dbeta, dvtab = ops.combination_block_bwd(dG, rows, summ, stats, gamma, Wo, Wqk, qk, vtab, mark, dropout=p,
                                                               seed=seed, site_gate=sg, site_out=so, dtype=dtype)
    # fp64 autograd from the SAVED q|k and pre-norm rows (what the kernel reads), with the engine's masks
    mg = ops.dropout_mask(seed, sg, n * 256, p).view(n, 256).double() if p > 0 else torch.ones(n, 256, device=DEV, dtype=torch.float64)
    mo = ops.dropout_mask(seed, so, n * 256, p).view(n, 256).double() if p > 0 else torch.ones(n, 256, device=DEV, dtype=torch.float64)

    class R16(torch.autograd.Function):                               # bf16 rounding of a product's operand, identity gradient
        @staticmethod
        def forward(ctx, x):
            return x.float().bfloat16().double()

        @staticmethod
        def backward(ctx, g):
            return g
    r16 = R16.apply if dtype in (1, 3) else (lambda t: t)
    dy = dG0[rows.long()].double()
    s_ = summ.double().requires_grad_(True)
    g_, b_ = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yy = F.layer_norm(s_, (256,), g_, b_, 1e-5)
    ds, dgam_ref, dbet_ref = torch.autograd.grad(yy, (s_, g_, b_), dy)
    dYc_ref = ds * mo
    dc = r16(dYc_ref) @ r16(Wo.double())                               # d(c Wo^T) / dc
    q_ = qk[:, :256].double().requires_grad_(True)
    k_ = qk[:, 256:].double().requires_grad_(True)
    vt_ = vtab.double().requires_grad_(True)
    v_ = vt_[mark.long()]
    gg = torch.softmax(torch.stack([q_ * k_ / math.sqrt(32), q_ * v_ / math.sqrt(32)], -1), -1)
    cc = (gg[..., 0] * k_ + gg[..., 1] * v_) * mg
    dq_ref, dk_ref, dvt_ref = torch.autograd.grad(cc, (q_, k_, vt_), dc)
    dX_ref = r16(dq_ref) @ r16(Wqk[:256].double()) + r16(dk_ref) @ r16(Wqk[256:].double())
    t1 = 2e-4 if dtype in (1, 3) else 5e-6
    assert rel_err(dYc, dYc_ref) < 5e-6
    assert rel_err(dqk[:, :256], dq_ref) < t1 and rel_err(dqk[:, 256:], dk_ref) < t1
    assert rel_err(dG[rows.long()], ds + dX_ref) < t1
    untouched = torch.ones(2 * n, dtype=torch.bool, device=DEV)
    untouched[rows.long()] = False
    assert torch.equal(dG[untouched], dG0[untouched])
    assert rel_err(dgamma, dgam_ref) < 1e-5 and rel_err(dbeta, dbet_ref) < 1e-5
    assert rel_err(dvtab, dvt_ref) < (2e-4 if dtype in (1, 3) else 1e-5)


def dense_graph_batch(B, N, density, seed):
    """symmetric random adjacency blocks with a full diagonal -> block-diagonal CSR (sorted unique columns) + dense copy"""
    rng = np.random.default_rng(seed)
    m = rng.random((B, N, N)) < density / 2
    m = m | m.transpose(0, 2, 1) | np.eye(N, dtype=bool)[None]
    dense = np.where(m, rng.uniform(0.05, 1.0, (B, N, N)), 0.0).astype(np.float32)
    b, r, c = np.nonzero(dense)
    rowptr = np.zeros(B * N + 1, dtype=np.int64)
    np.cumsum(np.bincount(b * N + r, minlength=B * N), out=rowptr[1:])
    t = lambda a, dt: torch.tensor(np.asarray(a, dtype=dt), device=DEV)
    return t(rowptr, np.int32), t(b * N + c, np.int32), t(dense[b, r, c], np.float32), torch.tensor(dense, device=DEV)


@pytest.mark.parametrize("N,density", [(512, 0.23), (512, 0.03), (500, 0.01), (96, 0.3), (64, 0.2), (33, 0.2)])
def test_block_dense_spmm_fp32_and_bf16(N, density):
    """Variants 3 / 4 (spmm_dense.hip): a workgroup densifies its rows of the graph's adjacency into LDS and runs the
    reference's literal bmm (gnn_transformer.py:80) on the matrix cores.  fp32 MFMA: the fp32 product (1e-6); bf16 MFMA:
    equal to the fp64 product of the bf16-ROUNDED operands (fp32 accumulation only), 4e-3 from the unrounded one."""
    from fira_icse_amd import ops
    B = 3
    rowptr, col, val, dense = dense_graph_batch(B, N, density, seed=N + 1)
    X = randn(B * N, 256, seed=12)
    ref = torch.bmm(dense.double(), X.view(B, N, 256).double()).view(B * N, 256)
    Y = ops.csr_spmm(rowptr, col, val, X, graph_rows=N, variant=3)
    assert rel_err(Y, ref) < 1e-6
    Yb = ops.csr_spmm(rowptr, col, val, X, graph_rows=N, variant=4)
    ref_b = torch.bmm(dense.bfloat16().double(), X.bfloat16().view(B, N, 256).double()).view(B * N, 256)
    assert rel_err(Yb, ref_b) < 2e-6
    assert rel_err(Yb, ref) < 6e-3
    # the density-based choice (fira_csr_spmm, variant 0): the same numbers as whichever kernel it picks
    Ya = ops.csr_spmm(rowptr, col, val, X, graph_rows=N, variant=0, auto=True)
    assert rel_err(Ya, ref) < 1e-6
    Yab = ops.csr_spmm(rowptr, col, val, X, graph_rows=N, variant=0, auto=True, dtype=1)
    assert rel_err(Yab, ref) < 6e-3
    # strided feature rows (ld 320) and an output that must keep its other columns
    Xw = torch.zeros(B * N, 320, device=DEV); Xw[:, :256] = X
    Yw = torch.full((B * N, 320), 7.0, device=DEV)
    ops.csr_spmm(rowptr, col, val, Xw[:, :256], graph_rows=N, variant=3, out=Yw[:, :256])
    assert torch.equal(Yw[:, :256], Y) and float((Yw[:, 256:] - 7.0).abs().max()) == 0


def test_row_wave_spmm_on_a_large_batch():
    """A launch of the size the engine issues at batch 64+ (19 500 rows) incl. rows longer than one 64-entry chunk."""
    from fira_icse_amd import ops
    B, N = 30, 650
    _, _, _, dense = dense_graph_batch(B, N, 0.006, seed=3)
    dense[:, 5, :200] = 0.25                                    # a few rows / columns with ~200 entries
    dense[:, :200, 5] = 0.25
    dn = dense.cpu().numpy()
    b, r, c = np.nonzero(dn)
    rp = np.zeros(B * N + 1, dtype=np.int64)
    np.cumsum(np.bincount(b * N + r, minlength=B * N), out=rp[1:])
    t = lambda a, dt: torch.tensor(np.asarray(a, dtype=dt), device=DEV)
    X = randn(B * N, 256, seed=13)
    ref = torch.bmm(dense.double(), X.view(B, N, 256).double()).view(B * N, 256)
    Y = ops.csr_spmm(t(rp, np.int32), t(b * N + c, np.int32), t(dn[b, r, c], np.float32), X, graph_rows=N, variant=1)
    assert rel_err(Y, ref) < 1e-6


def test_block_dense_spmm_sums_repeated_columns():
    """Sorted rows may repeat a column (a multigraph edge list that was not merged): the dense tile holds their sum."""
    from fira_icse_amd import ops
    N = 64
    rowptr, col, val = [0], [], []
    dense = np.zeros((N, N), dtype=np.float64)
    rng = np.random.default_rng(5)
    for r in range(N):
        cs = sorted(rng.integers(0, N, 9).tolist() + [r, r])          # repeats on purpose
        for c in cs:
            v = float(np.float32(rng.uniform(0.1, 1.0)))
            dense[r, c] += v
            col.append(c); val.append(v)
        rowptr.append(len(col))
    t = lambda a, dt: torch.tensor(np.array(a, dtype=dt), device=DEV)
    X = randn(N, 256, seed=3)
    ref = torch.tensor(dense, device=DEV) @ X.double()
    for variant in (1, 3):
        Y = ops.csr_spmm(t(rowptr, np.int32), t(col, np.int32), t(val, np.float32), X, graph_rows=N, variant=variant)
        assert rel_err(Y, ref) < 1e-6, variant


def test_block_dense_spmm_long_rows_with_runs_across_slot_boundaries():
    """densify_rows requests a row's entries in slots of 64 (three per row, then a tail loop): rows of ~330 entries whose runs
    of equal columns straddle entry 63|64, 127|128, 191|192 (the last slot's successor comes from memory) and sit in the tail."""
    from fira_icse_amd import ops
    N, B = 512, 2
    rng = np.random.default_rng(11)
    rowptr, col, val = [0], [], []
    dense = np.zeros((B, N, N), dtype=np.float64)
    for g in range(B):
        for r in range(N):
            n = 40 if r % 3 == 0 else 330                                  # short rows between the long ones
            cs = sorted(rng.choice(N, size=n, replace=False).tolist())
            for pos in (62, 63, 126, 127, 190, 191, 192, 255, 256, 300):   # duplicate the entry AT pos: a run across pos | pos + 1
                if pos < len(cs):
                    cs.insert(pos + 1, cs[pos])
            if r % 5 == 0 and len(cs) > 66:
                cs[61:66] = [cs[61]] * 5                                   # a run of five over the first slot boundary
                cs.sort()
            for c in cs:
                v = float(np.float32(rng.uniform(0.1, 1.0)))
                dense[g, r, c] += v
                col.append(g * N + c); val.append(v)
            rowptr.append(len(col))
    t = lambda a, dt: torch.tensor(np.array(a, dtype=dt), device=DEV)
    X = randn(B * N, 256, seed=4)
    dn = torch.tensor(dense, device=DEV)
    ref = torch.bmm(dn, X.view(B, N, 256).double()).view(B * N, 256)
    Y = ops.csr_spmm(t(rowptr, np.int32), t(col, np.int32), t(val, np.float32), X, graph_rows=N, variant=3)
    assert rel_err(Y, ref) < 1e-6
    Yb = ops.csr_spmm(t(rowptr, np.int32), t(col, np.int32), t(val, np.float32), X, graph_rows=N, variant=4)
    # the tile holds the bf16 rounding of the fp32 SUM of a run (summed last-to-first in fp32)
    ref_b = torch.bmm(dn.float().bfloat16().double(), X.bfloat16().view(B, N, 256).double()).view(B * N, 256)
    assert rel_err(Yb, ref_b) < 2e-4                                       # (a run's fp32 sum order may round the bf16 differently)
    assert rel_err(Yb, ref) < 6e-3


# ------------------------------------------------------------------------------------------------ row ops
def test_embed_gather_and_scatter():
    from fira_icse_amd import ops
    B, L, V = 5, 210, 1000
    g = torch.Generator().manual_seed(0)
    idx = torch.randint(0, V, (B, L), generator=g).to(torch.int32).to(DEV)
    idx[:, -20:] = 0
    table, pos = randn(V, 256, seed=1), randn(L, 256, seed=2)
    node = torch.zeros(B, 650, 256, device=DEV)
    ops.embed_gather(idx, table, pos, out=node, out_bstride=650, out_off=0)
    assert torch.equal(node[:, :L], table[idx.long()] + pos)
    assert float(node[:, L:].abs().max()) == 0
    dout = randn(B, 650, 256, seed=3)
    dtab = torch.zeros_like(table)
    ops.embed_scatter_add(idx, dtab, dout, 650, 0, padding_idx=0)
    ref = torch.zeros_like(table).double().index_add_(0, idx.long().view(-1), dout[:, :L].reshape(-1, 256).double())
    ref[0] = 0
    assert rel_err(dtab, ref) < 1e-6 and float(dtab[0].abs().max()) == 0


def test_combination_gate_fwd_bwd():
    from fira_icse_amd import ops
    M = 6720 + 3
    qk = randn(M, 512, seed=1).requires_grad_(True)
    vtab = randn(4, 256, seed=2).requires_grad_(True)
    mark = (torch.arange(M) % 4).to(torch.int32).to(DEV)
    q, k = qk[:, :256], qk[:, 256:]
    v = vtab[mark.long()]
    s = math.sqrt(32)
    w = torch.softmax(torch.stack([q * k / s, q * v / s], -1), -1)
    ref = (w * torch.stack([k, v], -1)).sum(-1)
    out = ops.combination_fwd(qk.detach(), vtab.detach(), mark)
    assert rel_err(out, ref) < 1e-6
    dout = randn(M, 256, seed=3)
    ref.backward(dout)
    dqk, dvtab = ops.combination_bwd(qk.detach(), vtab.detach(), mark, dout)
    assert rel_err(dqk, qk.grad) < 2e-6
    assert rel_err(dvtab, vtab.grad) < 1e-5


@pytest.mark.parametrize("M,K,N", [(960, 256, 256), (960, 1024, 1024), (1921, 256, 768), (64, 1024, 256), (7, 256, 2), (5100, 256, 1024)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_residual_block_split_at_its_layernorm(M, K, N, p):
    """fira_linear_presum_f32 + fira_ln_linear_f32 (LayerNorm in the consumer's prologue) against the three launches they
    replace -- product, add_layernorm_fwd (same dropout stream), product: the pre-norm sums are IDENTICAL (same MFMA chain,
    same epilogue order, same mask), the normalised rows / statistics / consumer output differ by reduction order only."""
    from fira_icse_amd import ops
    x = randn(M, K + 8, seed=1)[:, :K]                       # a strided view: ldx != K
    w, b = randn(256, K, seed=2, scale=K ** -0.5), randn(256, seed=3)
    res, gamma, beta = randn(M, 256, seed=4), 1 + 0.1 * randn(256, seed=5), 0.1 * randn(256, seed=6)
    w2, b2 = randn(N, 256, seed=7, scale=1 / 16), randn(N, seed=8)
    s = ops.linear_presum(x, w, b, res, p, seed=91, site=13)
    lin = ops.gemm(x, w, bias=b, transB=True)
    y0, s0, st0 = ops.add_layernorm_fwd(lin.clone(), res, gamma, beta, p, seed=91, site=13)
    if ((M + 31) // 32) * 8 <= 1024:                          # ops.gemm runs the same tile kernel: one MFMA chain
        assert torch.equal(s, s0)
    else:                                                    # beyond 1024 tiles ops.gemm is the LDS-tiled kernel (another k order)
        assert rel_err(s, s0) < 1e-6
    if p > 0:
        dropped = (s - res) == 0
        assert 0.07 < float(dropped.float().mean()) < 0.13
    for relu in (False, True):
        out, y, st = ops.ln_linear(s, w2, b2, gamma, beta, relu=relu)
        ref_out = ops.gemm(y0, w2, bias=b2, transB=True, relu=relu)
        assert rel_err(y, y0) < 1e-5 and rel_err(st, st0) < 1e-5 and rel_err(out, ref_out) < 1e-5
    ref_y = F.layer_norm(s.double(), (256,), gamma.double(), beta.double())
    assert rel_err(y, ref_y) < 5e-6
    assert rel_err(out, F.relu(F.linear(ref_y, w2.double(), b2.double()))) < 5e-6


@pytest.mark.parametrize("M,N,relu", [(530, 256, False), (530, 1024, True), (960, 256, False), (33, 256, False), (1921, 1024, True)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_layernorm_backward_in_the_dgrad_prologue(M, N, relu, p):
    """fira_ln_bwd_linear_f32 (LayerNorm backward in the prologue of the data-gradient product, gemm_small.hip LnB) against
    the two launches it replaces -- add_layernorm_bwd (same dropout stream) and the product (with the ReLU mask of the FFN) --
    and against autograd of LayerNorm in fp64.  Ragged last row block (M % 32 != 0), decoder- and FFN-sized outputs."""
    from fira_icse_amd import ops
    x = randn(M, 256, seed=1)
    gamma, beta = 1 + 0.1 * randn(256, seed=2), 0.1 * randn(256, seed=3)
    y, s, stats = ops.add_layernorm_fwd(x.clone(), None, gamma, beta)
    dy = randn(M, 256, seed=4)
    Wt = randn(256, N, seed=5, scale=1 / 16)
    mask = randn(M, N, seed=6) if relu else None
    dX, ds, dxd, dg, db = ops.ln_bwd_linear(dy, Wt, s, stats, gamma, relu_mask=mask, dropout=p, seed=55, site=21)
    ds0, dxd0, dg0, db0 = ops.add_layernorm_bwd(dy, s, stats, gamma, p, seed=55, site=21, want_dx_drop=True)
    ref_dX = ops.gemm(dxd0, Wt, transB=False)
    if relu:
        ref_dX = ref_dX * (mask > 0)
    assert rel_err(ds, ds0) < 2e-6 and rel_err(dxd, dxd0) < 2e-6
    assert rel_err(dg, dg0) < 1e-5 and rel_err(db, db0) < 1e-5
    assert rel_err(dX, ref_dX) < 3e-6
    if p > 0:
        assert torch.equal(dxd == 0, dxd0 == 0)                  # the same mask
    # fp64 autograd of the LayerNorm
    xs = s.double().requires_grad_(True)
    g64 = gamma.double().requires_grad_(True)
    b64 = beta.double().requires_grad_(True)
    F.layer_norm(xs, (256,), g64, b64, 1e-5).backward(dy.double())
    assert rel_err(ds, xs.grad) < 5e-6 and rel_err(dg, g64.grad) < 1e-5 and rel_err(db, b64.grad) < 1e-5


def test_add_layernorm_fwd_bwd_and_dropout_mask():
    from fira_icse_amd import ops
    M = 1237
    x, res = randn(M, 256, seed=1).requires_grad_(True), randn(M, 256, seed=2).requires_grad_(True)
    gamma, beta = (1 + 0.1 * randn(256, seed=3)).requires_grad_(True), randn(256, seed=4).requires_grad_(True)
    ref = F.layer_norm(x + res, (256,), gamma, beta, 1e-5)
    y, s, stats = ops.add_layernorm_fwd(x.detach().clone(), res.detach(), gamma.detach(), beta.detach())
    assert rel_err(y, ref) < 1e-6
    assert rel_err(s, x + res) < 1e-7
    dy = randn(M, 256, seed=5)
    ref.backward(dy)
    ds, _, dg, db = ops.add_layernorm_bwd(dy, s, stats, gamma.detach())
    assert rel_err(ds, x.grad) < 5e-6 and rel_err(ds, res.grad) < 5e-6
    assert rel_err(dg, gamma.grad) < 1e-5 and rel_err(db, beta.grad) < 1e-5
    # dropout: y = LN(x*mask/(1-p) (+ res)); the same counter-based mask is re-derived by the backward
    p = 0.2
    x0 = randn(M, 256, seed=6)
    y, s, stats = ops.add_layernorm_fwd(x0.clone(), None, gamma.detach(), beta.detach(), p, seed=77, site=5)
    kept = s != 0
    assert abs(float(kept.float().mean()) - (1 - p)) < 0.01
    assert rel_err(s[kept], x0[kept] / (1 - p)) < 1e-6
    ds, dxd, _, _ = ops.add_layernorm_bwd(dy, s, stats, gamma.detach(), p, seed=77, site=5, want_dx_drop=True)
    assert torch.equal(dxd != 0, kept & (ds != 0))
    assert rel_err(dxd[kept], ds[kept] / (1 - p)) < 1e-6
    y2, _, _ = ops.add_layernorm_fwd(x0.clone(), None, gamma.detach(), beta.detach(), p, seed=78, site=5)
    assert not torch.equal(y, y2)                     # a different seed draws a different mask


def test_colsum():
    from fira_icse_amd import ops
    X = randn(20800, 300, seed=1)
    assert rel_err(ops.colsum(X[:, :256].contiguous()), X[:, :256].double().sum(0)) < 1e-6


# ------------------------------------------------------------------------------------------------ attention
def torch_attention(q, k, v, key_valid, causal, q_pos0=0):
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    qh, kh, vh = (t.view(B, -1, 8, 32).transpose(1, 2) for t in (q, k, v))
    w = qh @ kh.transpose(-1, -2) / math.sqrt(32)
    mask = key_valid[:, None, None, :].bool()
    if causal:
        tri = (torch.arange(Tk, device=q.device)[None, :] <= torch.arange(Tq, device=q.device)[:, None] + q_pos0)
        mask = mask & tri[None, None]
    w = torch.softmax(w.masked_fill(~mask, -1e9), -1)
    return (w @ vh).transpose(1, 2).reshape(B, Tq, 256)


@pytest.mark.parametrize("Tq,Tk,causal", [(30, 30, True), (30, 370, False), (1, 17, False), (1, 370, False), (7, 33, False)])
def test_attention_fwd_bwd(Tq, Tk, causal):
    from fira_icse_amd import ops
    B = 5
    q = randn(B, Tq, 256, seed=1).double().requires_grad_(True)
    k = randn(B, Tk, 256, seed=2).double().requires_grad_(True)
    v = randn(B, Tk, 256, seed=3).double().requires_grad_(True)      # asymmetric V: catches transposed tiles
    g = torch.Generator().manual_seed(4)
    key_valid = (torch.rand(B, Tk, generator=g) > 0.3).to(torch.int32).to(DEV)
    key_valid[:, 0] = 1
    ref = torch_attention(q, k, v, key_valid, causal)
    o = ops.attention_fwd(q.detach().float(), k.detach().float(), v.detach().float(), key_valid, causal)
    assert rel_err(o, ref) < 2e-6
    do = randn(B, Tq, 256, seed=5)
    ref.backward(do.double())
    dq, dk, dv = ops.attention_bwd(q.detach().float(), k.detach().float(), v.detach().float(), key_valid, o, do, causal)
    assert rel_err(dq, q.grad) < 5e-6
    assert rel_err(dk, k.grad) < 5e-6
    assert rel_err(dv, v.grad) < 5e-6


@pytest.mark.parametrize("self_kv", [True, False])
def test_attention_on_ragged_query_rows_and_bf16_operands(self_kv):
    """fira_attention_{fwd,bwd}_ex, the forms the engine calls: (1) RAGGED query rows -- commit b's queries are a prefix
    of its 30 positions, stored compactly (fira_batch.dec_off); with self_kv the keys / values are the same compact rows
    (causal self-attention), otherwise a dense [B,370] memory -- must equal the dense computation on the rows that exist;
    (2) dtype bf16: the four matmuls on bf16-rounded operands, fp32 accumulation and soft-max (torch.autocast's
    Attention.forward): forward equal to the fp64 result of the ROUNDED operands / probabilities, gradients within
    bf16 resolution of the exact ones."""
    from fira_icse_amd import ops
    B, T, S = 6, 30, 370
    lens = [30, 1, 17, 29, 8, 12]
    off = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    rows = torch.cat([torch.arange(n) + b * T for b, n in enumerate(lens)]).to(DEV)
    Tk = T if self_kv else S
    q = randn(B, T, 256, seed=1).double().requires_grad_(True)
    k = randn(B, Tk, 256, seed=2).double().requires_grad_(True)
    v = randn(B, Tk, 256, seed=3).double().requires_grad_(True)
    g = torch.Generator().manual_seed(4)
    if self_kv:                                    # keys = the prefix positions themselves (all valid), causal
        key_valid = torch.zeros(B, T, dtype=torch.int32, device=DEV)
        for b, n in enumerate(lens):
            key_valid[b, :n] = 1
    else:
        key_valid = (torch.rand(B, S, generator=g) > 0.3).to(torch.int32).to(DEV)
        key_valid[:, 0] = 1
    ref = torch_attention(q, k, v, key_valid, self_kv)
    do_dense = torch.zeros(B * T, 256, device=DEV)
    do_dense[rows] = randn(len(rows), 256, seed=5)
    ref.backward(do_dense.view(B, T, 256).double())
    qc = q.detach().float().view(B * T, 256)[rows].contiguous()
    if self_kv:
        kc, vc = (t.detach().float().view(B * T, 256)[rows].contiguous() for t in (k, v))
    else:
        kc, vc = (t.detach().float().reshape(B * S, 256) for t in (k, v))
    pick = (lambda t: t.reshape(B * T, 256)[rows]) if self_kv else (lambda t: t.reshape(B * S, 256))
    doc = do_dense[rows].contiguous()
    for dtype, tol_o, tol_g in ((0, 2e-6, 5e-6), (1, 1e-2, 2e-2)):
        o = ops.attention_ragged_fwd(qc, kc, vc, key_valid, off, T, Tk, causal=self_kv, self_kv=self_kv, dtype=dtype)
        assert rel_err(o, ref.reshape(B * T, 256)[rows]) < tol_o, dtype
        dq, dk, dv = ops.attention_ragged_bwd(qc, kc, vc, key_valid, o, doc, off, T, Tk, causal=self_kv, self_kv=self_kv,
                                              dtype=dtype)
        assert rel_err(dq, q.grad.reshape(B * T, 256)[rows]) < tol_g, dtype
        assert rel_err(dk, pick(k.grad)) < tol_g and rel_err(dv, pick(v.grad)) < tol_g, dtype
    # bf16 forward against the same arithmetic in fp64: rounded q, k, v and rounded probabilities
    r16 = lambda t: t.detach().float().bfloat16().double()
    qh, kh, vh = (r16(t).view(B, -1, 8, 32).transpose(1, 2) for t in (q, k, v))
    w = qh @ kh.transpose(-1, -2) / math.sqrt(32)
    mask = key_valid[:, None, None, :].bool()
    if self_kv:
        mask = mask & (torch.arange(Tk, device=DEV)[None, :] <= torch.arange(T, device=DEV)[:, None])[None, None]
    w = torch.softmax(w.masked_fill(~mask, -1e9), -1).float().bfloat16().double()
    ref16 = (w @ vh).transpose(1, 2).reshape(B * T, 256)[rows]
    assert rel_err(o, ref16) < 5e-5


def _padding_style_memory(B, S=370, L=210, seed=7, holes=True):
    """FIRA-shaped key masks: a valid prefix of the 210 code slots and of the 160 sub-token slots, padding behind each
    (contiguous fully masked 32-key tiles next to live ones), and -- holes -- a few masked slots INSIDE the valid runs
    (computed memory rows whose id is 0: in the batch's node list because of an edge, masked as keys)."""
    rng = np.random.default_rng(seed)
    valid = np.zeros((B, S), np.int32)
    listed = np.zeros((B, S), bool)                   # the slots a batch would list as computed memory rows
    n_code = [200, 33, 64, 1, 97, 150, 31, 210][:B] + list(rng.integers(2, L, size=max(0, B - 8)))
    n_sub = [0, 5, 160, 0, 40, 64, 1, 100][:B] + list(rng.integers(0, S - L, size=max(0, B - 8)))
    for b in range(B):
        valid[b, :n_code[b]] = 1
        valid[b, L:L + n_sub[b]] = 1
        listed[b] = valid[b] != 0
        if holes and n_code[b] > 8:
            valid[b, [3, n_code[b] - 2]] = 0          # listed, but masked
    return valid, listed


@pytest.mark.parametrize("dtype", [0, 1])
def test_cross_attention_with_fully_masked_key_tiles_and_ragged_key_rows(dtype):
    """The two forms of the engine's cross attention against the fp64 soft-max of gnn_transformer.py:149-156, forward and
    backward, fp32 and bf16 operands:
      dense   K / V rows [B*370] with padding-style masks: whole 32-key tiles are masked (the dead-tile skip of
              attention.hip: scores -1e9, weight exactly 0, no MFMA chain) right next to live ones;
      ragged  (k_off) the computed memory rows only, commit after commit -- what the engine stores since round 4: the
              masked padding rows do not exist at all, masked-but-listed rows stay in the list.
    Both must give the same O / dQ as the reference and the same dK / dV on every listed row; the ragged launch must
    overwrite every dK / dV row it owns (the gradient buffers start as NaN)."""
    from fira_icse_amd import ops
    B, T, S = 8, 30, 370
    lens = [30, 1, 17, 29, 8, 12, 30, 5]
    off = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    rows = torch.cat([torch.arange(n) + b * T for b, n in enumerate(lens)]).to(DEV)
    valid_np, listed_np = _padding_style_memory(B)
    key_valid = torch.from_numpy(valid_np).to(DEV)
    assert any(not valid_np[b, 32 * t:32 * t + 32].any() and valid_np[b, 32 * (t - 1):32 * t].any()
               for b in range(B) for t in range(1, 11)), "the fixture must contain a dead tile behind a live one"
    q = randn(B, T, 256, seed=1).double().requires_grad_(True)
    k = randn(B, S, 256, seed=2).double().requires_grad_(True)
    v = randn(B, S, 256, seed=3).double().requires_grad_(True)
    ref = torch_attention(q, k, v, key_valid, False)
    do_dense = torch.zeros(B * T, 256, device=DEV)
    do_dense[rows] = randn(len(rows), 256, seed=5)
    ref.backward(do_dense.view(B, T, 256).double())
    qc = q.detach().float().view(B * T, 256)[rows].contiguous()
    doc = do_dense[rows].contiguous()
    kd, vd = (t.detach().float().reshape(B * S, 256) for t in (k, v))
    tol_o, tol_g = ((2e-6, 5e-6), (1e-2, 2e-2))[dtype]
    # ---- dense rows, padding-style masks
    o = ops.attention_ragged_fwd(qc, kd, vd, key_valid, off, T, S, dtype=dtype)
    assert rel_err(o, ref.reshape(B * T, 256)[rows]) < tol_o
    dq, dk, dv = ops.attention_ragged_bwd(qc, kd, vd, key_valid, o, doc, off, T, S, dtype=dtype, fill=float("nan"))
    assert rel_err(dq, q.grad.reshape(B * T, 256)[rows]) < tol_g
    assert rel_err(dk, k.grad.reshape(B * S, 256)) < tol_g and rel_err(dv, v.grad.reshape(B * S, 256)) < tol_g
    masked = torch.from_numpy(valid_np.reshape(-1) == 0).to(DEV)
    assert float(dk[masked].abs().max()) == 0.0 and float(dv[masked].abs().max()) == 0.0      # exact zeros, as masked_fill
    # ---- ragged key rows: the listed slots only, inside a wider buffer (the engine's [rows, 6*512 + pad] K|V)
    sel = torch.from_numpy(np.flatnonzero(listed_np.reshape(-1))).to(DEV)
    k_off = torch.tensor([0] + list(np.cumsum(listed_np.sum(1))), dtype=torch.int32, device=DEV)
    buf = randn(len(sel), 3136, seed=11)
    buf[:, 512:768], buf[:, 768:1024] = kd[sel], vd[sel]
    kc, vc = buf[:, 512:768], buf[:, 768:1024]
    kvc = key_valid.reshape(-1)[sel].contiguous()
    o2 = ops.attention_ragged_fwd(qc, kc, vc, kvc, off, T, S, dtype=dtype, k_off=k_off)
    assert rel_err(o2, ref.reshape(B * T, 256)[rows]) < tol_o
    if dtype == 0:
        assert rel_err(o2, o) < 1e-6                           # same arithmetic, other tile boundaries
    dq2, dk2, dv2 = ops.attention_ragged_bwd(qc, kc, vc, kvc, o2, doc, off, T, S, dtype=dtype, k_off=k_off, fill=float("nan"))
    assert bool(torch.isfinite(dk2).all()) and bool(torch.isfinite(dv2).all())      # every owned row was written
    assert rel_err(dq2, q.grad.reshape(B * T, 256)[rows]) < tol_g
    assert rel_err(dk2, k.grad.reshape(B * S, 256)[sel]) < tol_g and rel_err(dv2, v.grad.reshape(B * S, 256)[sel]) < tol_g
    hole = kvc == 0
    assert int(hole.sum()) > 0 and float(dk2[hole].abs().max()) == 0.0 and float(dv2[hole].abs().max()) == 0.0


def test_attention_row_without_any_valid_key_is_documented_zero():
    """include/fira_hip.h states the precondition "every query sees at least one unmasked key".  The reference's
    masked_fill(-1e9) + softmax (gnn_transformer.py:153-154) turns an all-masked row into the uniform average of V over
    all Tk keys, masked ones included; the kernels never read rows of masked keys (the engine does not even store them),
    so they return 0 for such a row and send no gradient anywhere.  FIRA batches cannot produce one (position 0 of every
    commit is <start>, Dataset.py:140).  This test pins the documented behaviour -- finite, exactly zero -- for the dense,
    the ragged and the one-query (decode) kernel, and that the OTHER commits of the launch are unaffected."""
    from fira_icse_amd import ops
    B, T, S = 3, 30, 370
    off = torch.tensor([0, 30, 60, 90], dtype=torch.int32, device=DEV)
    key_valid = torch.ones(B, S, dtype=torch.int32, device=DEV)
    key_valid[1] = 0                                             # commit 1: no valid key at all
    q, k, v = randn(B * T, 256, seed=1), randn(B * S, 256, seed=2), randn(B * S, 256, seed=3)
    ref = torch_attention(q.view(B, T, 256).double(), k.view(B, S, 256).double(), v.view(B, S, 256).double(), key_valid, False)
    o = ops.attention_ragged_fwd(q, k, v, key_valid, off, T, S)
    assert float(o[30:60].abs().max()) == 0.0
    assert rel_err(o[:30], ref[0]) < 2e-6 and rel_err(o[60:], ref[2]) < 2e-6
    dq, dk, dv = ops.attention_ragged_bwd(q, k, v, key_valid, o, randn(B * T, 256, seed=4), off, T, S, fill=float("nan"))
    assert float(dq[30:60].abs().max()) == 0.0 and float(dk[S:2 * S].abs().max()) == 0.0 and float(dv[S:2 * S].abs().max()) == 0.0
    assert bool(torch.isfinite(dq).all() and torch.isfinite(dk).all() and torch.isfinite(dv).all())
    # ragged: commit 1 lists no row at all (k_off[1] == k_off[2]); the decode kernel with the same ranges
    k_off = torch.tensor([0, S, S, 2 * S], dtype=torch.int32, device=DEV)
    kc, vc = torch.cat([k[:S], k[2 * S:]]), torch.cat([v[:S], v[2 * S:]])
    o2 = ops.attention_ragged_fwd(q, kc, vc, torch.ones(2 * S, dtype=torch.int32, device=DEV), off, T, S, k_off=k_off)
    assert float(o2[30:60].abs().max()) == 0.0 and rel_err(o2[60:], ref[2]) < 2e-6
    od = ops.decode_attention(q[[0, 30, 60]], kc, vc, torch.ones(2 * S, dtype=torch.int32, device=DEV), tk=S, k_off=k_off)
    assert float(od[1].abs().max()) == 0.0 and rel_err(od[2], ref[2, 0]) < 2e-6 and rel_err(od[0], ref[0, 0]) < 2e-6


def test_decode_attention_on_ragged_key_rows():
    """The decode step's cross attention on the engine's compact K|V rows (k_off): beam rows of a commit share its range."""
    from fira_icse_amd import ops
    Bk, qpk, S = 8, 3, 370
    valid_np, listed_np = _padding_style_memory(Bk)
    sel = torch.from_numpy(np.flatnonzero(listed_np.reshape(-1))).to(DEV)
    k_off = torch.tensor([0] + list(np.cumsum(listed_np.sum(1))), dtype=torch.int32, device=DEV)
    kd, vd = randn(Bk, S, 256, seed=2), randn(Bk, S, 256, seed=3)
    buf = randn(len(sel), 3136, seed=11)
    buf[:, 1024:1280], buf[:, 1280:1536] = kd.view(-1, 256)[sel], vd.view(-1, 256)[sel]
    q = randn(Bk * qpk, 256, seed=1)
    kv = torch.from_numpy(valid_np).to(DEV)
    o = ops.decode_attention(q, buf[:, 1024:1280], buf[:, 1280:1536], kv.reshape(-1)[sel].contiguous(), tk=S, qpk=qpk, k_off=k_off)
    ref = torch_attention(q.double()[:, None], kd.repeat_interleave(qpk, 0).double(), vd.repeat_interleave(qpk, 0).double(),
                          kv.repeat_interleave(qpk, 0), False)[:, 0]
    assert rel_err(o, ref) < 2e-6


def test_attention_strided_qkv_buffer():
    """Q|K|V read straight out of the fused [rows, 768] projection buffer (row stride 768)."""
    from fira_icse_amd import ops
    B, T = 4, 30
    qkv = randn(B, T, 768, seed=1)
    kv = torch.ones(B, T, dtype=torch.int32, device=DEV)
    o = ops.attention_fwd(qkv[:, :, :256], qkv[:, :, 256:512], qkv[:, :, 512:], kv, causal=True)
    ref = torch_attention(qkv[:, :, :256].double(), qkv[:, :, 256:512].double(), qkv[:, :, 512:].double(), kv, True)
    assert rel_err(o, ref) < 2e-6


@pytest.mark.parametrize("Tk,kb,qpk", [(370, 370, 1), (370, 370, 3), (17, 30, 1), (1, 30, 1), (333, 370, 2)])
def test_decode_attention_one_query_per_row(Tk, kb, qpk):
    """decode_attention (Tq = 1, valid keys streamed once per commit and head) against the fp64 soft-max, with the K/V rows
    inside a wider buffer (row stride 3072 like the engine's cross-K|V of six layers) and beam rows sharing a commit."""
    from fira_icse_amd import ops
    Bk = 6
    BR = Bk * qpk
    q = randn(BR, 256, seed=1)
    kvbuf = randn(Bk, kb, 3072, seed=2)
    k, v = kvbuf[:, :, 512:768], kvbuf[:, :, 768:1024]
    g = torch.Generator().manual_seed(3)
    key_valid = (torch.rand(Bk, kb, generator=g) > 0.5).to(torch.int32).to(DEV)
    key_valid[:, 0] = 1
    key_valid[1, :] = 1                                          # a commit with every key valid
    o = ops.decode_attention(q, k, v, key_valid, tk=Tk, qpk=qpk)
    kk = k[:, :Tk].repeat_interleave(qpk, 0).double()
    vv = v[:, :Tk].repeat_interleave(qpk, 0).double()
    ref = torch_attention(q.double()[:, None], kk, vv, key_valid[:, :Tk].repeat_interleave(qpk, 0), False)[:, 0]
    assert rel_err(o, ref) < 2e-6


def test_decode_attention_merged_new_key_is_used_and_appended():
    """Self-attention of the decode step with the merged q|k|v projection: key / value number Tk-1 come from the projection's
    output row and are appended to the cache; equal to attending over a cache that already holds them."""
    from fira_icse_amd import ops
    BR, T, Tk = 10, 30, 12
    qkv = randn(BR, 768, seed=4)
    kc, vc = randn(BR, T, 256, seed=5), randn(BR, T, 256, seed=6)
    hist = torch.ones(BR, T, dtype=torch.int32, device=DEV)
    hist[3, 5] = 0
    hist[4, Tk - 1] = 0                                          # a finished row: its newest key is masked, still appended
    kc_ref, vc_ref = kc.clone(), vc.clone()
    kc_ref[:, Tk - 1], vc_ref[:, Tk - 1] = qkv[:, 256:512], qkv[:, 512:]
    ref = torch_attention(qkv[:, None, :256].double(), kc_ref[:, :Tk].double(), vc_ref[:, :Tk].double(), hist[:, :Tk], False)[:, 0]
    o = ops.decode_attention(qkv[:, :256], kc, vc, hist, tk=Tk, knew=qkv[:, 256:512], vnew=qkv[:, 512:])
    assert rel_err(o, ref) < 2e-6
    assert torch.equal(kc, kc_ref) and torch.equal(vc, vc_ref)


# ------------------------------------------------------------------------------------------------ copy head / loss
def test_copy_score_fwd_bwd():
    from fira_icse_amd import ops
    B, T, S = 3, 30, 370
    src = randn(B, S, 256, seed=1).double().requires_grad_(True)
    tgt = randn(B, T, 256, seed=2).double().requires_grad_(True)
    w = randn(256, seed=3, scale=0.1).double().requires_grad_(True)
    b = randn(1, seed=4).double().requires_grad_(True)
    ref = (torch.tanh(src[:, None] + tgt[:, :, None]) * w).sum(-1) + b
    sc = ops.copy_score_fwd(src.detach().float(), tgt.detach().float(), w.detach().float(), b.detach().float())
    assert rel_err(sc, ref) < 2e-6
    ds = randn(B, T, S, seed=5)
    ref.backward(ds.double())
    dsrc, dtgt, dw, db = ops.copy_score_bwd(src.detach().float(), tgt.detach().float(), w.detach().float(), ds)
    assert rel_err(dsrc, src.grad) < 5e-6 and rel_err(dtgt, tgt.grad) < 5e-6
    assert rel_err(dw, w.grad) < 1e-5 and rel_err(db, b.grad) < 1e-5


def test_copy_score_bwd_sparse_rows_and_masked_slots():
    """The engine's shape of dscore: a few non-zero target rows (copy labels), whole commits without any, zeros on
    masked slots; the kernel skips the rows / tiles without gradient."""
    from fira_icse_amd import ops
    B, T, S = 4, 30, 370
    src = randn(B, S, 256, seed=1).double().requires_grad_(True)
    tgt = randn(B, T, 256, seed=2).double().requires_grad_(True)
    w = randn(256, seed=3, scale=0.1).double().requires_grad_(True)
    g = torch.Generator().manual_seed(0)
    valid = (torch.rand(B, S, generator=g) > 0.4).to(torch.int32).to(DEV)
    ds = torch.zeros(B, T, S, device=DEV)
    for b, t in ((0, 1), (0, 7), (2, 0), (2, 29), (3, 12)):          # commit 1 has no copy row at all
        ds[b, t] = randn(S, seed=10 + t) * valid[b]
    ds[3, 12, 100:] = 0                                                # a row that is zero on most slot tiles
    ref = (torch.tanh(src[:, None] + tgt[:, :, None]) * w).sum(-1)
    ref.backward(ds.double())
    dsrc, dtgt, dw, db = ops.copy_score_bwd(src.detach().float(), tgt.detach().float(), w.detach().float(), ds)
    assert rel_err(dsrc, src.grad) < 5e-6 and rel_err(dtgt, tgt.grad) < 5e-6 and rel_err(dw, w.grad) < 1e-5
    assert float(dsrc[1].abs().max()) == 0.0 and float(dtgt[1].abs().max()) == 0.0
    assert abs(float(db) - float(ds.sum())) < 1e-4 * float(ds.abs().sum())


@pytest.mark.parametrize("compact,V,ldl", [(False, 24650, 24704), (True, 24650, 24704), (False, 1001, 1001),
                                           (True, 1000, 1000)])
def test_head_loss_fwd_bwd(compact, V, ldl):
    """V = 24650 / 1000: the register-resident row path; V = 1001 (odd): the streaming path."""
    from fira_icse_amd import ops
    B, T, S = 4, 30, 370
    g = torch.Generator().manual_seed(0)
    logits = randn(B * T, V, seed=1, scale=2.0).double().requires_grad_(True)
    score = randn(B * T, S, seed=2, scale=2.0).double().requires_grad_(True)
    gate = randn(B * T, 2, seed=3).double().requires_grad_(True)
    mem_valid = (torch.rand(B, S, generator=g) > 0.4).to(torch.int32).to(DEV)
    mem_valid[:, :3] = 1
    lab = torch.zeros(B, T, dtype=torch.int64)
    lab[:, 0] = 2
    for b in range(B):
        n = 5 + 3 * b
        lab[b, 1:n] = torch.randint(4, V, (n - 1,), generator=g)
        valid = mem_valid[b].cpu().nonzero().view(-1)
        lab[b, 2] = V + int(valid[1])                 # a copy label
        lab[b, 4] = V + int((mem_valid[b] == 0).cpu().nonzero().view(-1)[0])   # copy label on a masked slot
        lab[b, n] = 1
    lab = lab.to(DEV)
    # reference (Model.py:54-82)
    p_gen = torch.softmax(logits, -1).view(B, T, V)
    p_copy = torch.softmax(score.view(B, T, S).masked_fill(mem_valid[:, None, :] == 0, -1e9), -1)
    gt = torch.softmax(gate, -1).view(B, T, 2)
    p = torch.cat([gt[..., :1] * p_gen, gt[..., 1:] * p_copy], -1)
    logp = torch.log(p.clamp(min=1e-10, max=1))
    label = torch.cat([lab[:, 1:], torch.zeros_like(lab[:, :1])], 1)
    nll = F.nll_loss(logp.view(-1, V + S), label.view(-1), reduction="none").masked_fill(label.view(-1) == 0, 0)
    nll.sum().backward()
    ids_ref = logp.argmax(-1)

    lg = torch.zeros(B * T, ldl, device=DEV)
    lg[:, :V] = logits.detach().float()
    sc, gl = score.detach().float().clone(), gate.detach().float().clone()
    compact_row = None
    rows = torch.arange(B * T, device=DEV)
    if compact:
        rows = ((label.view(-1) > 0) & (label.view(-1) < V)).nonzero().view(-1)
        lg = lg[rows].contiguous()
        compact_row = torch.full((B * T,), -1, dtype=torch.int32, device=DEV)
        compact_row[rows] = torch.arange(rows.numel(), dtype=torch.int32, device=DEV)
    else:
        _, _, ids = ops.head_loss(lg.clone(), sc.clone(), mem_valid, gl.clone(), lab.to(torch.int32), V,
                                  want_grad=False, argmax=True)
        assert torch.equal(ids.long(), ids_ref)
    loss, ntok, _ = ops.head_loss(lg, sc, mem_valid, gl, lab.to(torch.int32), V, compact_row=compact_row)
    assert int(ntok) == int((label != 0).sum())
    assert abs(float(loss) - float(nll.sum())) / float(nll.sum()) < 2e-6
    assert rel_err(lg[:, :V], logits.grad[rows]) < 5e-6
    assert rel_err(sc, score.grad) < 5e-6
    assert rel_err(gl, gate.grad) < 5e-6


def test_adam_matches_torch():
    from fira_icse_amd import ops
    n = 100003
    p0, grads = randn(n, seed=1), [randn(n, seed=10 + i, scale=0.01) for i in range(4)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-4)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    ntok = torch.tensor([8], dtype=torch.int32, device=DEV)
    inv = ops.inv_count(ntok, torch.zeros(1, device=DEV))
    for i, gr in enumerate(grads):
        ref.grad = gr / 8
        opt.step()
        ops.adam_step(p, gr, m, v, 1e-4, i + 1, inv_scale=inv)
        assert float((p - ref.data).abs().max()) < 2e-7


def test_adam_micro_batch_form_matches_torch():
    """fira_adam_step_mb: g = g0 (+ g1), normaliser 1 / max(n_tok0 (+ n_tok1), 1) formed on the device."""
    from fira_icse_amd import ops
    n = 70001
    p0 = randn(n, seed=1)
    ga, gb = [randn(n, seed=20 + i, scale=0.01) for i in range(3)], [randn(n, seed=30 + i, scale=0.01) for i in range(3)]
    n0 = torch.tensor([5], dtype=torch.int32, device=DEV)
    n1 = torch.tensor([3], dtype=torch.int32, device=DEV)
    for two in (False, True):
        ref = torch.nn.Parameter(p0.clone())
        opt = torch.optim.Adam([ref], lr=1e-4)
        p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        for i in range(3):
            ref.grad = (ga[i] + gb[i]) / 8 if two else ga[i] / 5
            opt.step()
            ops.adam_step_mb(p, ga[i], gb[i] if two else None, m, v, 1e-4, i + 1, n0, n1 if two else None)
            assert float((p - ref.data).abs().max()) < 2e-7
    zero = torch.zeros(1, dtype=torch.int32, device=DEV)        # no labelled token: the normaliser is 1, not inf
    p = p0.clone()
    ops.adam_step_mb(p, ga[0], None, torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), 1e-4, 1, zero)
    assert bool(torch.isfinite(p).all())


def test_adam_count_form_and_pack_stats():
    """Data-parallel form: fira_pack_stats writes {loss_sum, float(n_tok)}; fira_adam_step_count scales by 1 / max(count, 1)
    formed inside the kernel from the (all-reduced) float count."""
    from fira_icse_amd import ops
    n = 50021
    p0, grads = randn(n, seed=1), [randn(n, seed=40 + i, scale=0.01) for i in range(3)]
    loss = torch.tensor([12.5], device=DEV)
    ntok = torch.tensor([11], dtype=torch.int32, device=DEV)
    stats = ops.pack_stats(loss, ntok, torch.zeros(2, device=DEV))
    assert stats.tolist() == [12.5, 11.0]
    stats += stats                                               # "all-reduce" over two identical ranks: count 22
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-4)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for i, gr in enumerate(grads):
        ref.grad = gr / 22
        opt.step()
        ops.adam_step_count(p, gr, m, v, 1e-4, i + 1, stats[1:2])
        assert float((p - ref.data).abs().max()) < 2e-7
    p = p0.clone()                                               # an all-empty global batch: the normaliser is 1, not inf
    ops.adam_step_count(p, grads[0], torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), 1e-4, 1, torch.zeros(1, device=DEV))
    assert bool(torch.isfinite(p).all())


# ------------------------------------------------------------------------------------------------ SURVEY 8(b) block entries
@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("M,p", [(530, 0.0), (530, 0.1), (37, 0.1), (2000, 0.1)])
def test_ffn_block_fwd_bwd(M, p, dtype):
    """fira_ffn_fwd / fira_ffn_bwd: the FeedForward block of gnn_transformer.py:163-174 as one entry each, against the fp64
    statement LN(dropout(relu(x W1^T + b1) W2^T + b2) + x) with the engine's own dropout mask and its autograd; dtype 1:
    operands of the two products (and of their gradients) rounded to bf16 -- tolerance of the bf16 level."""
    from fira_icse_amd import ops
    F_ = 1024
    x = randn(M, 256, seed=1)
    w1, b1 = randn(F_, 256, seed=2, scale=0.06), randn(F_, seed=3, scale=0.1)
    w2, b2 = randn(256, F_, seed=4, scale=0.03), randn(256, seed=5, scale=0.1)
    gamma, beta = 1 + randn(256, seed=6, scale=0.1), randn(256, seed=7, scale=0.1)
    seed, site = 99, 46
    h, summ, y, stats = ops.ffn_fwd(x, w1, b1, w2, b2, gamma, beta, dropout=p, seed=seed, site=site, dtype=dtype)
    mask = ops.dropout_mask(seed, site, M * 256, p).view(M, 256).double() if p > 0 else torch.ones(M, 256, device=DEV).double()
    leaf = [t.double().clone().requires_grad_(True) for t in (x, w1, b1, w2, b2, gamma, beta)]
    X, W1, B1, W2, B2, G, Bt = leaf
    H = torch.relu(X @ W1.t() + B1)
    S = (H @ W2.t() + B2) * mask + X
    Y = F.layer_norm(S, (256,), G, Bt, 1e-5)
    # (bf16: every product rounds both operands, 2^-9 each; the gradients pass two products and a ReLU mask taken from the
    #  rounded forward pass -- a few per cent against the exact fp64 gradient, as in tests/test_bf16_gpu.py)
    t1, t2 = (2e-6, 1e-5) if dtype == 0 else (6e-3, 5e-2)
    assert rel_err(h, H.detach()) < t1 and rel_err(summ, S.detach()) < t1 and rel_err(y, Y.detach()) < 2 * t1
    dy = randn(M, 256, seed=8)
    Y.backward(dy.double())
    dx, dw1, db1, dw2, db2, dg, db = ops.ffn_bwd(dy, x, h, summ, stats, w1, w2, gamma, dropout=p, seed=seed, site=site, dtype=dtype)
    for got, want, name in ((dx, X.grad, "dx"), (dw1, W1.grad, "dw1"), (db1, B1.grad, "db1"), (dw2, W2.grad, "dw2"),
                            (db2, B2.grad, "db2"), (dg, G.grad, "dgamma"), (db, Bt.grad, "dbeta")):
        assert rel_err(got, want) < t2, (name, rel_err(got, want))


@pytest.mark.parametrize("R,V,k", [(64, 24650, 1), (192, 24650, 3), (7, 1001, 8)])
def test_head_topk(R, V, k):
    """fira_head_topk: generator logits + the k best of every row, value descending, ties by ascending id -- against torch.topk
    on the fp64 logits (values), with exact duplicates planted to pin the tie rule."""
    from fira_icse_amd import ops
    x = randn(R, 256, seed=1)
    w, b = randn(V, 256, seed=2, scale=0.2), randn(V, seed=3)
    w[5], b[5] = w[900].clone(), b[900].clone()            # ids 5 and 900 tie exactly in every row
    ids, vals, logits = ops.head_topk(x, w, b, k)
    ref = x.double() @ w.double().t() + b.double()
    assert rel_err(logits, ref) < 2e-6
    tv, ti = torch.topk(logits, k, dim=1)                   # the kernel ranks ITS logits: compare on those
    assert torch.equal(vals, tv)
    # ids: identical wherever the value is unique in the row; on ties the lower id comes first
    want = torch.empty_like(ids)
    lg = logits.clone()
    for j in range(k):
        m = lg.max(dim=1, keepdim=True).values
        first = (lg == m).float().argmax(dim=1)             # argmax of a 0/1 row = the FIRST maximal position
        want[:, j] = first.to(torch.int32)
        lg[torch.arange(R, device=DEV), first] = -float("inf")
    assert torch.equal(ids, want)
    boost = torch.zeros(V, device=DEV)
    boost[5] = boost[900] = 100.0
    ids2, vals2, _ = ops.head_topk(x, w, b + boost, min(k, 2) if k > 1 else 1)
    assert bool((ids2[:, 0] == 5).all()) and (k == 1 or bool((ids2[:, 1] == 900).all()))


@pytest.mark.parametrize("M,N,K,lda", [(256, 256, 10031, 256), (512, 256, 3531, 512), (256, 1024, 530, 256), (24650, 256, 411, 24704),
                                       (768, 256, 530, 768), (96, 256, 17, 96), (3072, 256, 5000, 3136)])
@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("split", [True, False])
def test_wgrad_panel_product(M, N, K, lda, dtype, split):
    """fira_gemm_wgrad_panel (gemm_wgrad_panel.hip): C += A^T B, colsum += column sums of A, on the shapes of the training step
    (encoder / decoder / vocabulary / stacked K|V weight gradients; ragged last tile, K not a multiple of 16, padded pitches).
    dtype 0: the three-term bf16 split must be fp32-ACCURATE -- against fp64 at the tolerance of the exact-fp32 kernels, and not
    worse than the fp32 MFMA kernel on the same operands; dtype 1: exact on the bf16-rounded operands."""
    from fira_icse_amd import ops
    Afull = randn(K, lda, seed=1)
    A = Afull[:, :M]
    B = randn(K, N, seed=2)
    C0 = randn(M, N, seed=3)
    cs0 = randn(M, seed=4)
    C, cs = C0.clone(), cs0.clone()
    if not split and K > 6000:
        pytest.skip("one slab: covered by the shorter reductions")
    ops.gemm_wgrad_panel(A, B, C, cs, dtype=dtype, split=split)
    r16 = (lambda t: t.float().bfloat16().double()) if dtype else (lambda t: t.double())
    ref = C0.double() + r16(A).t() @ r16(B)
    assert rel_err(cs, cs0.double() + A.double().sum(0)) < 2e-6          # (column sums are taken from the fp32 registers)
    err = rel_err(C, ref)
    assert err < (3e-6 if K <= 4096 else 6e-6), err
    if dtype == 0:
        C32 = C0.clone()
        ops.gemm(A, B, transA=True, transB=False, out=C32, accumulate=True)
        e32 = rel_err(C32, ref)
        assert err < 2.0 * e32 + 2e-7, (err, e32)                        # as accurate as the k-ordered fp32 chains


@pytest.mark.parametrize("R,V", [(530, 24650), (37, 24650), (1021, 4099), (64, 250), (2049, 1000)])
def test_head_logits_three_term_product(R, V):
    """fira_head_logits_x3 (head_x3.hip): out_fc (Model.py:54) as three bf16 terms per operand on the bf16 matrix cores --
    against the fp64 product, at the tolerance of the fp32 product (the engine's fp32 mode runs this kernel); ragged row blocks
    (R not a multiple of 16), ragged vocabulary tiles (V not a multiple of 16 or 4), fewer tiles than workgroups, more than one
    chunk of row blocks (R > 512); the fp32 product of the tiled kernel beside it for comparison."""
    from fira_icse_amd import ops
    x = randn(R, 256, seed=1)
    W, b = randn(V, 256, seed=2, scale=0.06), randn(V, seed=3, scale=0.1)
    ref = x.double() @ W.double().t() + b.double()
    got = ops.head_logits_x3(x, W, b)
    f32 = ops.gemm(x, W, bias=b)
    e3, e32 = rel_err(got, ref), rel_err(f32, ref)
    assert e3 < 1e-6, (e3, e32)
    assert e3 < 3 * e32 + 1e-7, (e3, e32)                      # no worse than the fp32 chain
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6


@pytest.mark.parametrize("M,N,ldo", [(4331, 3072, 3136), (8104, 1024, 1024), (17, 256, 320), (600, 512, 512)])
def test_linear_three_term_product(M, N, ldo):
    """fira_linear_x3 (comb_fused.hip: linear_x3_kernel) -- the training step's cross-attention K|V projection of all layers on the
    memory rows (gnn_transformer.py:139-141): against the fp64 product at the tolerance of the fp32 product (dtype 2: three bf16
    terms) and against the fp64 product of the bf16-rounded operands (dtype 3: one plane); ragged last tile, fewer tiles than
    workgroups, two tiles per workgroup, a padded output pitch (the engine's K|V rows) whose padding stays untouched."""
    from fira_icse_amd import ops
    x = randn(M, 256, seed=1)
    W, b = randn(N, 256, seed=2, scale=0.06), randn(N, seed=3, scale=0.1)
    ref = x.double() @ W.double().t() + b.double()
    got = ops.linear_x3(x, W, b, dtype=2, ldo=ldo)
    f32 = ops.gemm(x, W, bias=b)
    e3, e32 = rel_err(got, ref), rel_err(f32, ref)
    assert e3 < 1e-6, (e3, e32)
    assert e3 < 3 * e32 + 1e-7, (e3, e32)
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6
    if ldo > N:
        assert not got.untyped_storage() is None and float(got.as_strided((M, ldo - N), (ldo, 1), N).abs().max()) == 0.0
    xb, Wb = x.bfloat16().double(), W.bfloat16().double()
    ref1 = xb @ Wb.t() + b.double()
    got1 = ops.linear_x3(x, W, b, dtype=3, ldo=ldo)
    assert rel_err(got1, ref1) < 2e-6, rel_err(got1, ref1)
    got0 = ops.linear_x3(x, W, None, dtype=2)
    assert rel_err(got0, x.double() @ W.double().t()) < 1e-6


@pytest.mark.parametrize("M,K,pitch", [(4331, 1024, 3136), (8104, 512, 512), (17, 256, 256), (700, 2048, 2048)])
def test_linear_dgrad_three_term_product(M, K, pitch):
    """fira_linear_dgrad_x3 (comb_fused.hip: linear_x3_kacc_kernel) -- the d-memory products of the decoder's backward pass
    (d memory += dK|dV Wkv, gnn_transformer.py:139-141 backward): a column slice of padded rows as dy, accumulation into dx,
    against fp64 at the fp32 product's tolerance (dtype 2) and against the bf16-rounded operands (dtype 3)."""
    from fira_icse_amd import ops
    full = randn(M, pitch, seed=4)
    dy = full[:, :K]
    W = randn(K, 256, seed=5, scale=0.06)
    base = randn(M, 256, seed=6)
    ref = base.double() + dy.double() @ W.double()
    dx = base.clone()
    ops.linear_dgrad_x3(dy, W, dx, dtype=2)
    f32 = base.clone()
    ops.gemm(dy.contiguous(), W, transB=False, out=f32, accumulate=True)
    e3, e32 = rel_err(dx, ref), rel_err(f32, ref)
    assert e3 < 1e-6, (e3, e32)
    assert e3 < 3 * e32 + 1e-7, (e3, e32)
    fresh = ops.linear_dgrad_x3(dy, W, None, dtype=2)
    assert rel_err(fresh, dy.double() @ W.double()) < 1e-6
    ref1 = base.double() + dy.bfloat16().double() @ W.bfloat16().double()
    dx1 = base.clone()
    ops.linear_dgrad_x3(dy, W, dx1, dtype=3)
    assert rel_err(dx1, ref1) < 2e-6, rel_err(dx1, ref1)


@pytest.mark.parametrize("M,K", [(530, 24650), (1130, 24650), (37, 4099), (16, 250), (2049, 1000)])
def test_vocabulary_dgrad_three_term_split_k(M, K):
    """fira_dgrad_x3_splitk (comb_fused.hip: dgrad_x3_splitk_kernel): the generator projection's data gradient d dec += dlogits Wout
    (Model.py:54 backward) -- the reduction over the vocabulary split over workgroups, float atomics; ragged last K block (24650 =
    96 x 256 + 74; K not a multiple of 4 either), padded row pitch with GARBAGE in the padding, ragged row tiles; against fp64 at
    the fp32 product's tolerance (dtype 2) and against the bf16-rounded operands (dtype 3)."""
    from fira_icse_amd import ops
    pitch = (K + 63) // 64 * 64
    full = torch.full((M, pitch), float("nan"), device="cuda")
    full[:, :K] = randn(M, K, seed=7, scale=0.05)
    dy = full[:, :K]
    W = randn(K, 256, seed=8, scale=0.06)
    base = randn(M, 256, seed=9)
    ref = base.double() + dy.double() @ W.double()
    dx = ops.dgrad_x3_splitk(dy, W, base.clone(), dtype=2)
    f32 = base.clone()
    ops.gemm(dy.contiguous(), W, transB=False, out=f32, accumulate=True)
    e3, e32 = rel_err(dx, ref), rel_err(f32, ref)
    assert e3 < 1e-6, (e3, e32)
    assert e3 < 3 * e32 + 2e-7, (e3, e32)
    ref1 = base.double() + dy.bfloat16().double() @ W.bfloat16().double()
    dx1 = ops.dgrad_x3_splitk(dy, W, base.clone(), dtype=3)
    assert rel_err(dx1, ref1) < 2e-6, rel_err(dx1, ref1)


@pytest.mark.parametrize("dtype", [2, 3])
def test_gcn_layer_batch64_sized(dtype):
    """The fused GCN launch at batch 64's size (19 500 rows = 4.8 tiles per workgroup: two passes, the second for one tile; hub rows of
    13 and 40 entries: the gather's tail path) -- forward and backward, every graph against the fp64 statement; dtype 2 = three bf16
    terms (fp32 tolerance), 3 = one plane (bf16 tolerance: 5e-5 at this size)."""
    from fira_icse_amd import ops
    B, N, nnz, p = 30, 650, 2600, 0.2
    rng = np.random.default_rng(77)
    dense_np = np.zeros((B, N, N), dtype=np.float32)
    for b in range(B):                                       # symmetric pattern + diagonal + two hub rows (13 .. 40 entries: the tail path)
        i, j = rng.integers(0, N, (2, nnz // 2))
        dense_np[b, i, j] = rng.uniform(0.05, 1.0, i.shape).astype(np.float32)
        dense_np[b, j, i] = dense_np[b, i, j]
        for hub, deg in ((5, 40), (311, 13)):
            cols = rng.choice(N, deg, replace=False)
            dense_np[b, hub, cols] = rng.uniform(0.05, 1.0, deg).astype(np.float32)
            dense_np[b, cols, hub] = dense_np[b, hub, cols]
        dense_np[b, np.arange(N), np.arange(N)] = rng.uniform(0.05, 1.0, N).astype(np.float32)
    bb, rr, cc = np.nonzero(dense_np)                        # row-major: sorted by (graph, row, column)
    rowptr_np = np.zeros(B * N + 1, dtype=np.int64)
    np.add.at(rowptr_np, bb * N + rr + 1, 1)
    rowptr = torch.tensor(np.cumsum(rowptr_np).astype(np.int32), device=DEV)
    col = torch.tensor((bb * N + cc).astype(np.int32), device=DEV)
    val = torch.tensor(dense_np[bb, rr, cc], device=DEV)
    dense = torch.tensor(dense_np, device=DEV)
    n = B * N
    X = randn(n, 256, seed=1)
    W21, b2, c21 = randn(256, 256, seed=2, scale=0.06), randn(256, seed=3, scale=0.1), randn(256, seed=4, scale=0.1)
    gamma, beta = 1 + randn(256, seed=5, scale=0.1), randn(256, seed=6, scale=0.1)
    seed, site = 99, 23
    summ, y, stats, rs = ops.gcn_layer_fwd(rowptr, col, val, X, W21.t().contiguous(), b2, c21, gamma, beta, dropout=p, seed=seed, site=site,
                                           dtype=dtype)
    r16 = (lambda t: t.float().bfloat16().double()) if dtype == 3 else (lambda t: t.double())
    U = torch.bmm(dense.double(), X.view(B, N, 256).double()).view(n, 256)
    rowsum = dense.double().sum(2).view(n, 1)
    pre = r16(U) @ r16(W21).t() + b2.double() + rowsum * c21.double()
    mask = ops.dropout_mask(seed, site, n * 256, p).view(n, 256).double()
    ref_sum = pre * mask + X.double()
    ref_y = F.layer_norm(ref_sum, (256,), gamma.double(), beta.double(), 1e-5)
    per_graph = lambda a, b: float(((a.double() - b).view(B, -1).norm(dim=1) / b.view(B, -1).norm(dim=1)).max())
    tol = 5e-5 if dtype == 3 else 2e-6
    assert per_graph(rs.view(n, 1), rowsum) < 1e-6
    assert per_graph(summ, ref_sum) < tol and per_graph(y, ref_y) < 5 * tol
    dY, dX0 = randn(n, 256, seed=7), randn(n, 256, seed=8)
    dX = dX0.clone()
    V = ops.gcn_layer_bwd(rowptr, col, val, dY, W21, dX, dtype=dtype)
    refV = torch.bmm(dense.double(), dY.view(B, N, 256).double()).view(n, 256)
    assert per_graph(V, refV) < 1e-6
    assert per_graph(dX, dX0.double() + r16(refV) @ r16(W21)) < tol
