"""bf16 compute path (BASELINE configs[2]; SURVEY.md §8b "Dtypes fp32 and bf16 (fp32 accumulate)").

Gates: the bf16 GEMM equals an fp32/fp64 product of the bf16-ROUNDED operands to fp32 summation-order
accuracy (2e-6 relative: the only difference to the fp32 kernel is the rounding of the inputs, which the test applies to
the reference too); whole model: loss within 1e-2 relative of the fp32 engine and of the reference's golden value,
teacher-forced argmax ids >= 99 % identical, and a 20-step Adam loss curve at batch 32 that tracks the reference's fp32
curve within 2e-2.
"""
import json
import os

import numpy as np
import pytest
import torch

import util
from fira_icse_amd import data, synth
from fira_icse_amd.config import FiraConfig

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def randn(*shape, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g).to(DEV)


def bf(x):
    return x.to(torch.bfloat16).double()          # round-to-nearest-even, the MFMA path's input rounding


@pytest.mark.parametrize("M,N,K", [(960, 256, 256), (20800, 256, 256), (130, 70, 50), (64, 24650, 256),
                                   (333, 256, 24650), (1920, 1024, 256), (1920, 256, 1024), (700, 768, 256),
                                   (65, 129, 200)])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn", "tt", "nt-t1", "nn-t1", "tn-t1", "nt-t3", "tn-t3"])
def test_gemm_bf16_layouts(M, N, K, layout):
    from fira_icse_amd import ops
    tile = int(layout[-1]) if "-t" in layout else 0          # 1: 128x128, 3: 64x64, 0: automatic
    tA, tB = {"nt": (False, True), "nn": (False, False), "tn": (True, False), "tt": (True, True)}[layout[:2]]
    A = randn(*((K, M) if tA else (M, K)), seed=1)
    B = randn(*((N, K) if tB else (K, N)), seed=2)           # asymmetric operands: a swapped tile cannot pass
    bias = randn(N, seed=3)
    aligned = A.stride(0) % 4 == 0 and B.stride(0) % 4 == 0 and min(M, N, K) >= 32
    rd = bf if aligned else (lambda x: x.double())           # what the library computes for shapes it forwards to fp32
    ref = (rd(A.t() if tA else A) @ rd(B.t() if tB else B)) + bias.double()
    tol = 2e-6 if K <= 4096 else 6e-6
    out = ops.gemm(A, B, transA=tA, transB=tB, bias=bias, tile=tile, dtype="bf16")
    assert rel_err(out, ref) < tol
    out = ops.gemm(A, B, transA=tA, transB=tB, bias=bias, relu=True, tile=tile, dtype="bf16")
    assert rel_err(out, ref.clamp_min(0)) < tol
    C0 = randn(M, N, seed=9)
    for sk in (1, 0, 3):                                       # plain accumulate, automatic split, forced split-K
        out = ops.gemm(A, B, transA=tA, transB=tB, out=C0.clone(), accumulate=True, tile=tile, splitk=sk, dtype="bf16")
        assert rel_err(out, ref - bias.double() + C0.double()) < tol, sk


def test_gemm_bf16_rounding_is_nearest_even_and_differs_from_fp32():
    from fira_icse_amd import ops
    A, B = randn(256, 256, seed=4), randn(256, 256, seed=5)
    out = ops.gemm(A, B, dtype="bf16")
    exact = A.double() @ B.double().t()
    assert rel_err(out, bf(A) @ bf(B).t()) < 2e-6
    assert 1e-3 < rel_err(out, exact) < 1e-2              # bf16 inputs: ~2^-9 relative per element


def test_gemm_bf16_strided_output_row_map_and_padded_ld():
    from fira_icse_amd import ops
    dl = torch.zeros(100, 24704, device=DEV)
    dl[:, :24650] = randn(100, 24650, seed=9)
    dl[:, 24650:] = float("nan")                           # the K tail must be masked, not multiplied by zero
    Wo = randn(24650, 256, seed=10)
    out = ops.gemm(dl[:, :24650], Wo, transB=False, dtype="bf16")
    assert rel_err(out, bf(dl[:, :24650]) @ bf(Wo)) < 6e-6


@pytest.mark.parametrize("M,N,K", [(1920, 256, 256), (24000, 256, 256), (130, 768, 256), (1920, 256, 1024), (700, 24650, 256),
                                   (1000, 512, 3072)])
def test_weight_shadows_forward_and_dgrad(M, N, K):
    """Forward Y = X W^T reads the stored shadow, the data gradient dX = dY W the transposed one -- both through the
    k-contiguous bf16 B operand; row slices of a stacked weight (the per-layer K|V blocks) use the parent's pitch."""
    from fira_icse_amd import ops
    X, W, bias = randn(M, K, seed=1), randn(N, K, seed=2), randn(N, seed=3)
    wb, wbt = ops.weight_shadow(W)
    assert torch.equal(wb.view(torch.bfloat16).float(), W.to(torch.bfloat16).float())
    assert torch.equal(wbt.view(torch.bfloat16).float(), W.t().to(torch.bfloat16).float())
    ref = bf(X) @ bf(W).t() + bias.double()
    assert rel_err(ops.gemm_wb(X, wb, bias=bias), ref) < 2e-6
    assert rel_err(ops.gemm_wb(X, wb, bias=bias, relu=True), ref.clamp_min(0)) < 2e-6
    if N % 8 == 0:
        dY = randn(M, N, seed=4)
        C0 = randn(M, K, seed=5)
        dref = bf(dY) @ bf(W)
        assert rel_err(ops.gemm_wb(dY, wbt), dref) < 2e-6
        for sk in (1, 0, 4):
            assert rel_err(ops.gemm_wb(dY, wbt, out=C0.clone(), accumulate=True, splitk=sk), dref + C0.double()) < 3e-6, sk
        if N >= 512:                       # rows [256, 512) of the stacked weight: a column slice of the transposed shadow
            sl = slice(256, 512)
            got = ops.gemm_wb(dY[:, sl], wbt[:, sl])
            assert rel_err(got, bf(dY[:, sl]) @ bf(W[sl])) < 2e-6


@pytest.mark.parametrize("M,N", [(64, 64), (130, 96), (70, 40), (1000, 257), (8500, 3072), (1400, 24650), (24001, 512)])
def test_panel_kernel_k256_edges_accumulate_relu(M, N):
    """K = 256 products take the A-stationary panel kernel (gemm_bf16_panel.hip) unless they are decode-sized: ragged row
    panels, ragged 64-column tiles, both panel heights (BM 128 when N >= 512 and M*N >= 4 Mi), bias / ReLU / accumulate."""
    from fira_icse_amd import ops
    X, W, bias = randn(M, 256, seed=11), randn(N + (-N) % 8, 256, seed=12)[:N], randn(N, seed=13)
    wb, _ = ops.weight_shadow(W)
    ref = bf(X) @ bf(W).t()
    assert rel_err(ops.gemm_wb(X, wb), ref) < 2e-6
    assert rel_err(ops.gemm_wb(X, wb, bias=bias, relu=True), (ref + bias.double()).clamp_min(0)) < 2e-6
    C0 = randn(M, N, seed=14)
    got = ops.gemm_wb(X, wb, bias=bias, out=C0.clone(), accumulate=True)
    assert rel_err(got, ref + bias.double() + C0.double()) < 3e-6
    # a strided A (column slice of a wider activation) and a strided C
    Xw = randn(M, 512, seed=15)
    Cw = torch.zeros(M, N + 8, device=DEV)
    ops.gemm_wb(Xw[:, 256:], wb, out=Cw[:, :N])
    assert rel_err(Cw[:, :N], bf(Xw[:, 256:]) @ bf(W).t()) < 2e-6 and float(Cw[:, N:].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(5000, 256, 512), (4100, 320, 768), (3000, 512, 1024)])
def test_panel_kernel_k_slices(M, N, K):
    """K = 512 / 768 / 1024 run as a chain of K = 256 panel launches, the later slices accumulating atomically (one owner
    per element: deterministic); with a ReLU the tiled kernel takes the call instead -- both against the same reference."""
    from fira_icse_amd import ops
    X, W, bias, C0 = randn(M, K, seed=21), randn(N, K, seed=22), randn(N, seed=23), randn(M, N, seed=24)
    wb, _ = ops.weight_shadow(W)
    ref = bf(X) @ bf(W).t() + bias.double()
    a, b = ops.gemm_wb(X, wb, bias=bias), ops.gemm_wb(X, wb, bias=bias)
    assert rel_err(a, ref) < 2e-6 and torch.equal(a, b)
    assert rel_err(ops.gemm_wb(X, wb, bias=bias, out=C0.clone(), accumulate=True), ref + C0.double()) < 3e-6
    assert rel_err(ops.gemm_wb(X, wb, bias=bias, relu=True), ref.clamp_min(0)) < 2e-6


@pytest.mark.parametrize("M", [64, 1920, 10001, 24000])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_linear_layernorm_bf16_fused_equals_the_two_kernels(M, p):
    """The panel kernel with a LayerNorm epilogue against gemm_wb + add_layernorm_fwd: the products are the same MFMA
    sequence and the dropout indices are the same, so the pre-norm rows are IDENTICAL; statistics / outputs differ by the
    reduction order only (1e-5)."""
    from fira_icse_amd import ops
    x = randn(M, 264, seed=31)[:, :256]                      # strided rows
    w, b = randn(256, 256, seed=32) / 16, randn(256, seed=33)
    res, gamma, beta = randn(M, 256, seed=34), 1 + 0.1 * randn(256, seed=35), 0.1 * randn(256, seed=36)
    wb, _ = ops.weight_shadow(w)
    y, s, st = ops.linear_layernorm_bf16_fwd(x, wb, b, res, gamma, beta, p, seed=17, site=9)
    lin = ops.gemm_wb(x, wb, bias=b)
    y0, s0, st0 = ops.add_layernorm_fwd(lin.clone(), res, gamma, beta, p, seed=17, site=9)
    if M >= 4096:                                            # gemm_wb takes the panel kernel too: the same MFMA sequence
        assert torch.equal(s, s0)
    else:                                                    # gemm_wb = latency kernel (K split over four waves)
        assert rel_err(s, s0) < 2e-6 and torch.equal((s - res) == 0, (s0 - res) == 0)
    assert rel_err(y, y0) < 1e-5 and rel_err(st, st0) < 1e-5
    y1, _, _ = ops.linear_layernorm_bf16_fwd(x, wb, None, None, gamma, beta)
    ref = torch.nn.functional.layer_norm(bf(x) @ bf(w).t(), (256,), gamma.double(), beta.double())
    assert rel_err(y1, ref) < 5e-6


@pytest.fixture(scope="module")
def small():
    from fira_icse_amd.model import TransModel, DeviceBatch, reference_init_state_dict
    cfg = FiraConfig()
    store = data.process_raw(cfg, util.load_golden_raw())
    idx = data.split_index(*util.GOLDEN_SPLIT, seed=0)
    hb = store.batch(idx["train"][:util.GOLDEN_B])
    torch.manual_seed(0)
    sd = util.perturb_state_dict(reference_init_state_dict(cfg), seed=1)
    model = TransModel(cfg, init=False)
    model.load_state_dict(sd)
    model.eval()
    return cfg, model, DeviceBatch(hb, cfg), sd


def test_model_bf16_loss_grad_and_ids_vs_fp32(small):
    cfg, model, db, sd = small
    g = util.golden_npz("model_ref.npz")
    model.compute_dtype = "f32"
    l32, n32 = model.train_fwd_bwd(db)
    l32, g32 = float(l32), model.gbuf.clone()
    ids32 = model.forward_dev(db)
    model.compute_dtype = "bf16"
    try:
        l16, n16 = model.train_fwd_bwd(db)
        l16, g16 = float(l16), model.gbuf.clone()
        ids16 = model.forward_dev(db)
    finally:
        model.compute_dtype = "f32"
    assert int(n16) == int(n32) == int(g["n_tok"])
    assert abs(l16 - l32) / l32 < 1e-2 and abs(l16 - float(g["loss_sum"])) / float(g["loss_sum"]) < 1e-2
    assert l16 != l32                                         # the bf16 kernels really ran
    # teacher-forced argmax on the 71 labelled positions of this 4-commit batch: with freshly initialised weights the
    # 25 020-way distribution is nearly flat (top-2 margins ~1e-3 relative), so a position or two may flip under bf16
    # rounding; the >= 99 % gate is applied where it is meaningful, on trained weights (the curve test below)
    valid = db.tar_label.view(ids16.shape)[:, 1:] != 0
    agree = (ids16[:, :-1] == ids32[:, :-1])[valid].float().mean()
    assert float(agree) >= 0.95, float(agree)
    # gradient: direction and size agree with fp32 (bf16 input rounding: ~1e-2 relative on the whole vector)
    live = model.layout.live
    cos = torch.nn.functional.cosine_similarity(g16[:live].double(), g32[:live].double(), dim=0)
    assert float(cos) > 0.999, float(cos)
    assert abs(float(g16[:live].norm() / g32[:live].norm()) - 1) < 2e-2


def test_bf16_loss_curve_batch32_tracks_reference_fp32_curve():
    """20 Adam steps at batch 32, dropout off, against the reference's own fp32 curve."""
    from fira_icse_amd.model import TransModel, DeviceBatch, reference_init_state_dict
    from fira_icse_amd.train import Trainer
    gold = json.load(open(os.path.join(util.GOLDEN, "large_ref.json")))
    cfg = FiraConfig()
    store = data.process_raw(cfg, synth.generate_dataset(util.LARGE_N, seed=util.LARGE_SEED))
    idx = data.split_index(*util.LARGE_SPLIT, seed=0)["train"]
    torch.manual_seed(0)
    model = TransModel(cfg, init=False)
    model.load_state_dict(util.perturb_state_dict(reference_init_state_dict(cfg), seed=1))
    model.eval()
    model.compute_dtype = "bf16"
    trainer = Trainer(model)
    batches = [DeviceBatch(store.batch(idx[i * 32:(i + 1) * 32]), cfg) for i in range(4)]
    curve = []
    for it in range(20):
        trainer.step(batches[it % 4])
        curve.append(trainer.last_loss())
    ref = np.array(gold["loss_curve"])
    dev = np.abs(np.array(curve) - ref) / ref
    assert dev.max() < 2e-2, (dev.max(), curve, gold["loss_curve"])
    assert curve[-1] < curve[0] - 0.5                         # it learns
    # teacher-forced argmax ids of the trained weights: bf16 forward vs fp32 forward, >= 99 % identical
    model.eval()
    agree, total = 0, 0
    for db in batches:
        model.compute_dtype = "bf16"
        ids16 = model.forward_dev(db)
        model.compute_dtype = "f32"
        ids32 = model.forward_dev(db)
        valid = db.tar_label.view(ids16.shape)[:, 1:] != 0
        agree += int((ids16[:, :-1] == ids32[:, :-1])[valid].sum())
        total += int(valid.sum())
    assert total > 1000 and agree / total >= 0.99, (agree, total)


def test_bf16_loss_and_gradient_at_batch64_configs2():
    """BASELINE configs[2]'s per-GPU workload (batch 64, bf16): loss of one step against the CPU oracle's fp32 value on the
    same 64 commits (1e-2, the bf16 budget of SURVEY.md §8c) and against the engine's own fp32 step; the fp32 step itself
    must sit on the oracle (1e-5), so the comparison is anchored at this batch size and not only at 4 / 32."""
    from fira_icse_amd.model import TransModel, DeviceBatch, reference_init_state_dict
    from oracle import fira_oracle as O
    cfg = FiraConfig()
    store = data.process_raw(cfg, synth.generate_dataset(util.LARGE_N, seed=util.LARGE_SEED))
    idx = data.split_index(*util.LARGE_SPLIT, seed=0)["train"][:64]
    hb = store.batch(idx)
    torch.manual_seed(0)
    sd = util.perturb_state_dict(reference_init_state_dict(cfg), seed=1)
    model = TransModel(cfg, init=False)
    model.load_state_dict(sd)
    model.eval()
    db = DeviceBatch(hb, cfg)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    with torch.no_grad():
        ls, nt = O.forward(sd, cfg, t(hb.sou), t(hb.tar), t(hb.mark), t(hb.ast_change), t(hb.dense_edge(cfg.graph_len)),
                           t(hb.tar_label), t(hb.sub_token), "train")
    ls, nt = float(ls), int(nt)
    model.compute_dtype = "f32"
    l32, n32 = model.train_fwd_bwd(db)
    l32, n32, g32 = float(l32), int(n32), model.gbuf.clone()
    model.compute_dtype = "bf16"
    try:
        l16, n16 = model.train_fwd_bwd(db)
        l16, n16, g16 = float(l16), int(n16), model.gbuf.clone()
    finally:
        model.compute_dtype = "f32"
    assert n32 == n16 == nt
    assert abs(l32 - ls) / ls < 1e-5, (l32, ls)
    assert abs(l16 - ls) / ls < 1e-2 and l16 != l32, (l16, ls)
    live = model.layout.live
    cos = torch.nn.functional.cosine_similarity(g16[:live].double(), g32[:live].double(), dim=0)
    assert float(cos) > 0.999, float(cos)
    assert abs(float(g16[:live].norm() / g32[:live].norm()) - 1) < 2e-2


def test_bf16_small_vocabulary_transposed_shadows_do_not_overlap():
    """A user vocabulary with V % 8 != 0 and V small (1003 words): the transposed bf16 shadow of out_fc.weight has rows
    padded to 1008 and is larger than the tensor itself.  The shadows are packed at their own offsets, so it must not
    run into the shadow of the cross-attention K|V weights that follows it in the layout: the bf16 gradients
    of exactly those tensors are compared with the fp32 engine's."""
    from fira_icse_amd.model import TransModel, DeviceBatch, reference_init_state_dict
    V = 1003
    cfg = FiraConfig(vocab_size=V)
    store = data.process_raw(cfg, synth.generate_dataset(8, seed=5, vocab_size=V))
    hb = store.batch(range(8))
    torch.manual_seed(0)
    model = TransModel(cfg, init=False)
    model.load_state_dict(util.perturb_state_dict(reference_init_state_dict(cfg), seed=1))
    model.eval()
    db = DeviceBatch(hb, cfg)
    model.compute_dtype = "f32"
    l32, _ = model.train_fwd_bwd(db)
    l32 = float(l32)
    g32 = {k: v.clone() for k, v in model.grad_views().items()}
    model.compute_dtype = "bf16"
    try:
        l16, _ = model.train_fwd_bwd(db)
        l16 = float(l16)
        g16 = {k: v.clone() for k, v in model.grad_views().items()}
    finally:
        model.compute_dtype = "f32"
    assert abs(l16 - l32) / l32 < 1e-2
    keys = ["out_fc.weight", "copy_net.LinearSource.weight", "copy_net.LinearTarget.weight"] + \
        ["decoder.cross_attention_list.%d.fc_%s.weight" % (l, c) for l in range(6) for c in "kv"] + \
        ["decoder.feed_forward_list.5.fc2.weight", "encoder.gcn_list.0.fc1.weight"]
    for k in keys:
        a, b = g16[k].double().flatten(), g32[k].double().flatten()
        cos = float(torch.nn.functional.cosine_similarity(a, b, dim=0))
        assert cos > 0.995, (k, cos)
