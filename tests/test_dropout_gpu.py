"""Training-mode parity: the engine's counter-based dropout masks are exported (fira_dropout_mask) and applied by the
CPU oracle at the reference's six dropout sites, so the loss AND every gradient of a dropout-on step are compared
with the oracle at the fp32 gates (loss 1e-5, gradient rel-L2 1e-4 per tensor) -- a wrong keep probability, scale or
site in any of the ~40 dropout applications of a step fails here.  Also: the keep rate and the
1/(1-p) scale of the mask function itself, and the (seed, rank, step) derivation of the stream."""
import numpy as np
import pytest
import torch

import util
from fira_icse_amd import data, ops
from fira_icse_amd.config import FiraConfig

pytestmark = pytest.mark.gpu


def test_mask_statistics_and_determinism():
    for p in (0.1, 0.2):
        m = ops.dropout_mask(1234567, 7, 1 << 20, p)
        keep = float((m > 0).float().mean())
        assert abs(keep - (1 - p)) < 3e-3, keep
        assert torch.allclose(m[m > 0], torch.full_like(m[m > 0], 1 / (1 - p)))
        assert torch.equal(m, ops.dropout_mask(1234567, 7, 1 << 20, p))
        assert not torch.equal(m, ops.dropout_mask(1234567, 8, 1 << 20, p))        # another site
        assert not torch.equal(m, ops.dropout_mask(1234568, 7, 1 << 20, p))        # another seed
    assert float(ops.dropout_mask(1, 1, 1000, 0.0).min()) == 1.0


def test_dropout_stream_depends_on_seed_rank_and_step():
    from fira_icse_amd.model import TransModel
    model = TransModel(FiraConfig(), init=False)
    seen = set()
    for seed in (0, 1):
        for rank in (0, 1):
            model.set_dropout_stream(seed, rank)
            for step in (1, 2):
                model.dropout_step = step
                seen.add(model.dropout_seed)
    assert len(seen) == 8


def test_training_mode_loss_and_gradients_vs_oracle_with_the_same_masks():
    from oracle import fira_oracle as O
    from fira_icse_amd.model import TransModel, DeviceBatch, reference_init_state_dict
    cfg = FiraConfig()
    store = data.process_raw(cfg, util.load_golden_raw())
    idx = data.split_index(*util.GOLDEN_SPLIT, seed=0)
    hb = store.batch(idx["train"][:util.GOLDEN_B])
    torch.manual_seed(0)
    sd = util.perturb_state_dict(reference_init_state_dict(cfg), seed=1)
    model = TransModel(cfg, init=False)
    model.load_state_dict(sd)
    model.train()
    model.set_dropout_stream(5, 1)
    model.dropout_step = 41
    db = DeviceBatch(hb, cfg)
    loss, ntok = model.train_fwd_bwd(db)                       # dropout 0.1 / 0.2 (the reference's rates)
    loss, ntok = float(loss), int(ntok)
    seed = model.dropout_seed                                  # the seed that step used
    B, N, L, T = len(hb), cfg.graph_len, cfg.sou_len, cfg.tar_len
    node_rows = db.node_rows.cpu().long()
    code_nodes = node_rows[db.code_rows.cpu().long()]          # global node index b*N + local of every computed code row

    def drop(kind, layer, x):
        p = 0.2 if kind == O.GCN else cfg.dropout_rate
        site = layer * 8 + kind + 1                            # FIRA_SITE(layer, kind)
        if kind in (O.GATE, O.COMB_OUT):                       # x: [B, L, 256]; rows = computed code rows
            m = ops.dropout_mask(seed, site, db.n_code * 256, p).cpu().view(-1, 256)
            full = torch.ones(B * L, 256)
            full[(code_nodes // N) * L + code_nodes % N] = m
        elif kind == O.GCN:                                    # x: [B, N, 256]; rows = computed nodes
            m = ops.dropout_mask(seed, site, db.n_nodes * 256, p).cpu().view(-1, 256)
            full = torch.ones(B * N, 256)
            full[node_rows] = m
        else:                                                  # decoder sites: x [B, T, 256]; rows = computed target rows
            m = ops.dropout_mask(seed, site, db.n_dec_rows * 256, p).cpu().view(-1, 256)
            full = torch.ones(B * T, 256)                      # (the padded tail is not computed: its mask is irrelevant)
            full[torch.from_numpy(db.dec_rows_host).long()] = m
        return x * full.view(x.shape)

    tb = util.to_torch_batch(hb, cfg)
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ls, nt = O.forward(P, cfg, tb["sou"], tb["tar"], tb["mark"], tb["ast_change"], tb["edge"], tb["tar_label"],
                       tb["sub_token"], "train", drop=drop)
    ls.backward()
    assert int(nt) == ntok
    assert abs(loss - float(ls)) / float(ls) < 1e-5, (loss, float(ls))
    model.eval()
    l_eval, _ = model.train_fwd_bwd(db)
    assert abs(float(l_eval) - loss) / loss > 1e-4            # the masks really were applied
    model.train(); model.dropout_step = 41
    model.train_fwd_bwd(db)
    gv = model.grad_views()
    gmax = max(float(p.grad.norm()) for p in P.values() if p.grad is not None)
    worst = ("", 0.0)
    for k, p in P.items():
        if p.grad is None:
            assert float(gv[k].abs().max()) == 0.0, k
            continue
        ref = p.grad.double()
        err = float((gv[k].cpu().double() - ref).norm())
        tol = 1e-4 * float(ref.norm()) + 1e-6 * gmax          # floor: tensors whose true gradient is zero (fc_k.bias)
        if "feed_forward_list" in k and "fc1" in k:
            tol *= 3                                          # one ReLU sign flip at fp32 rounding moves 2e-4 of the norm
        if err / max(tol, 1e-30) > worst[1]:
            worst = (k, err / max(tol, 1e-30))
        assert err <= tol, (k, err, float(ref.norm()))
