"""CPU oracle for the FIRA hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain fp32 PyTorch-CPU restatement of the reference's model and
decode algorithm, written from the behaviour of the reference (file:line cited
per function).  It exists so that the HIP engine can be checked on a GPU box
where ``/root/reference`` is absent.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``fira_icse_amd`` never does, and fails loudly when its HIP library is missing.

Pinning: ``tests/golden/make_golden.py`` runs the *reference's own modules*
(imported from ``/root/reference`` in the build container) on seeded inputs and
weights and commits the results under ``tests/golden/``; ``tests/test_oracle.py``
checks this restatement against those fixtures (and, when the reference tree is
present, against the live reference).

Parameters are a flat ``dict[str, Tensor]`` with the reference's state-dict
keys.  Dropout: the reference draws its masks from torch's global RNG, which no other implementation
can reproduce; the functions below take an optional ``drop(kind, layer, x)`` callable applied at the reference's six
dropout sites (combination_layer.py:15-16; gnn_transformer.py:205, :83, :161 twice, :174) so that a test can feed the
engine's own counter-based masks (``fira_dropout_mask``) and compare training-mode losses and gradients exactly.
``drop=None`` = eval() / p = 0.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------- constants
def position_table(length: int, d: int) -> torch.Tensor:
    """pos[i,2j] = sin(i / 10000^(2j/d)), pos[i,2j+1] = cos(same); float64 maths, stored fp32
    (reference gnn_transformer.py:10-19)."""
    rows = []
    for i in range(length):
        row = []
        for j in range(d // 2):
            a = i / (10000 ** (2 * j / d))
            row += [math.sin(a), math.cos(a)]
        rows.append(row)
    return torch.tensor(rows)


def _lin(P: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def _ln(P: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], 1e-5)


# ----------------------------------------------------------------------------- encoder
GATE, COMB_OUT, GCN, SELF, CROSS, FFN = range(6)      # dropout site kinds (include/fira_hip.h FIRA_SITE_*)


def _drop(drop, kind, layer, x):
    return x if drop is None else drop(kind, layer, x)


def combination(P: Params, pre: str, x: torch.Tensor, mark_em: torch.Tensor, d_k: int, drop=None, layer=0) -> torch.Tensor:
    """Element-wise 2-way gated fusion of (q, k, v) + output projection + post-LN
    (reference gnn_transformer.py:192-205, combination_layer.py:7-17).
    The head split / transposes of the reference cancel: the op is element-wise."""
    q = _lin(P, pre + ".linear_layers.0", x)
    k = _lin(P, pre + ".linear_layers.1", x)
    v = _lin(P, pre + ".linear_layers.2", mark_em)
    s = math.sqrt(d_k)
    w = torch.softmax(torch.stack([q * k / s, q * v / s], -1), dim=-1)
    mixed = _drop(drop, GATE, layer, (w * torch.stack([k, v], -1)).sum(-1))
    return _ln(P, pre + ".layernorm", _drop(drop, COMB_OUT, layer, _lin(P, pre + ".output_linear", mixed)) + x)


def gcn(P: Params, pre: str, x: torch.Tensor, adj32: torch.Tensor, drop=None, layer=0) -> torch.Tensor:
    """LN(fc2(A_hat @ fc1(x)) + x) (reference gnn_transformer.py:74-86)."""
    h = _lin(P, pre + ".fc1", x)
    z = torch.bmm(adj32, h)
    return _ln(P, pre + ".layernorm", _drop(drop, GCN, layer, _lin(P, pre + ".fc2", z)) + x)


def encoder(P: Params, cfg, sou, mark, ast_change, edge, sub_token, drop=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reference gnn_transformer.py:45-62. ``edge`` is the [B,N,N] adjacency (any float dtype)."""
    L, S = cfg.sou_len, cfg.sub_token_len
    # padding_idx=0 on the three encoder tables (gnn_transformer.py:32-39): row 0 never receives a gradient, which is
    # observable when an over-long diff chains id-0 sub-token nodes into the graph
    emb = P["encoder.embedding.weight"]
    x = F.embedding(sou, emb, padding_idx=0) + position_table(L, cfg.embedding_dim)
    mark_em = F.embedding(mark, P["encoder.mark_embedding.weight"], padding_idx=0)
    ast_em = F.embedding(ast_change, P["encoder.ast_change_embedding.weight"], padding_idx=0)
    sub_em = F.embedding(sub_token, emb, padding_idx=0)
    adj32 = edge.float()
    for i in range(cfg.num_layers):
        x = combination(P, "encoder.combination_list2.%d" % i, x, mark_em, cfg.d_head, drop, i)
        g = gcn(P, "encoder.gcn_list.%d" % i, torch.cat([x, sub_em, ast_em], 1), adj32, drop, i)
        x, sub_em, ast_em = g[:, :L], g[:, L:L + S], g[:, L + S:]
    return x, sub_em


# ----------------------------------------------------------------------------- decoder
def attention(P: Params, pre: str, q_in, kv_in, mask, heads: int, drop=None, kind=SELF, layer=0) -> torch.Tensor:
    """Multi-head attention, -1e9 masking, post-LN residual (reference gnn_transformer.py:137-161).
    ``mask``: [B,Tk] (key padding) or [B,1,Tq,Tk]."""
    B, Tq, d = q_in.shape
    Tk = kv_in.shape[1]
    dh = d // heads
    q = _lin(P, pre + ".fc_q", q_in).view(B, Tq, heads, dh).transpose(1, 2)
    k = _lin(P, pre + ".fc_k", kv_in).view(B, Tk, heads, dh).transpose(1, 2)
    v = _lin(P, pre + ".fc_v", kv_in).view(B, Tk, heads, dh).transpose(1, 2)
    w = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dh)
    if mask.dim() < 4:
        mask = mask[:, None, None, :]
    w = torch.softmax(w.masked_fill(mask == 0, -1e9), dim=-1)
    o = torch.matmul(w, v).transpose(1, 2).contiguous().view(B, Tq, d)
    return _ln(P, pre + ".layernorm", _drop(drop, kind, layer, _lin(P, pre + ".fc_o", o)) + q_in)


def feed_forward(P: Params, pre: str, x, drop=None, layer=0) -> torch.Tensor:
    """Reference gnn_transformer.py:170-174."""
    h = _lin(P, pre + ".fc2", torch.relu(_lin(P, pre + ".fc1", x)))
    return _ln(P, pre + ".layernorm", _drop(drop, FFN, layer, h) + x)


def decoder(P: Params, cfg, tar, memory, mem_mask, tar_pad_mask, drop=None) -> torch.Tensor:
    """Reference gnn_transformer.py:108-122: key-pad AND causal mask on self-attention."""
    T = cfg.tar_len
    x = P["decoder.embedding.weight"][tar] + position_table(T, cfg.embedding_dim)
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
    self_mask = tar_pad_mask[:, None, None, :] & causal[None, None]
    for i in range(cfg.num_layers):
        x = attention(P, "decoder.attention_list.%d" % i, x, x, self_mask, cfg.num_head, drop, SELF, i)
        x = attention(P, "decoder.cross_attention_list.%d" % i, x, memory, mem_mask, cfg.num_head, drop, CROSS, i)
        x = feed_forward(P, "decoder.feed_forward_list.%d" % i, x, drop, i)
    return x


# ----------------------------------------------------------------------------- head
def copy_net(P: Params, memory, dec) -> Tuple[torch.Tensor, torch.Tensor]:
    """Additive copy attention + 2-way gate (reference Model.py:15-20)."""
    src = F.linear(memory, P["copy_net.LinearSource.weight"])
    tgt = F.linear(dec, P["copy_net.LinearTarget.weight"])
    score = _lin(P, "copy_net.LinearRes", torch.tanh(src[:, None] + tgt[:, :, None])).squeeze(-1)
    gate = torch.softmax(_lin(P, "copy_net.LinearProb", dec), dim=-1)
    return score, gate


def output_distribution(P: Params, memory, mem_mask, dec) -> torch.Tensor:
    """[gate0*softmax(out_fc) ; gate1*softmax(masked copy scores)] (reference Model.py:54-64)."""
    p_gen = torch.softmax(_lin(P, "out_fc", dec), dim=-1)
    score, gate = copy_net(P, memory, dec)
    p_copy = torch.softmax(score.masked_fill(mem_mask[:, None, :] == 0, -1e9), dim=-1)
    return torch.cat([gate[..., 0:1] * p_gen, gate[..., 1:2] * p_copy], -1)


def encode_memory(P: Params, cfg, sou, mark, ast_change, edge, sub_token, drop=None):
    code, sub = encoder(P, cfg, sou, mark, ast_change, edge, sub_token, drop)
    return torch.cat([code, sub], 1), torch.cat([sou != 0, sub_token != 0], 1)


def forward(P: Params, cfg, sou, tar, mark, ast_change, edge, tar_label, sub_token, stage="train", drop=None):
    """Whole model (reference Model.py:38-86): 'train' -> (loss_sum, n_tok); 'dev'/'test' -> argmax ids [B,T]."""
    memory, mem_mask = encode_memory(P, cfg, sou, mark, ast_change, edge, sub_token, drop)
    dec = decoder(P, cfg, tar, memory, mem_mask, tar != 0, drop)
    logp = torch.log(output_distribution(P, memory, mem_mask, dec).clamp(min=1e-10, max=1))
    label = torch.cat([tar_label[:, 1:], torch.zeros_like(tar_label[:, :1])], 1)     # shift left, pad with 0
    keep = label != 0
    nll = F.nll_loss(logp.reshape(-1, logp.shape[-1]), label.reshape(-1), reduction="none")
    nll = nll.masked_fill(~keep.reshape(-1), 0)
    if stage == "train":
        return nll.sum(), keep.sum()
    return torch.argmax(logp, dim=-1)


# ----------------------------------------------------------------------------- decode loop
def beam_decode(P: Params, cfg, sou, mark, ast_change, edge, sub_token, beam: int,
                start_id: int = 2, eos_id: int = 1, pad_id: int = 0, trace=None,
                max_steps: int = None) -> Tuple[List[List[List[int]]], List[List[float]]]:
    """The reference's test-time search (run_model.py:202-340), full recompute per step.

    Returns (hypotheses [B][beam] -> vocab-id list starting with <start>, probabilities [B][beam]).
    ``beam == 1`` is the reference's "greedy" mode.  Scores are products of probabilities in fp32,
    finished rows are forced to -1, and the top-``beam`` are taken from a descending sort, exactly as
    the reference does.
    """
    with torch.no_grad():
        memory, mem_mask = encode_memory(P, cfg, sou, mark, ast_change, edge, sub_token)
        B, T, V, L = sou.shape[0], cfg.tar_len, cfg.vocab_size, cfg.sou_len
        W = cfg.out_len
        hyp = [[[start_id] for _ in range(beam)] for _ in range(B)]
        prob = [[1.0 if j == 0 else 0.0 for j in range(beam)] for _ in range(B)]
        for step in range(T - 1 if max_steps is None else min(T - 1, max_steps)):   # max_steps: bounded timing runs
            blocks, active = [], []
            for j in range(beam):
                alive = [hyp[i][j][-1] != eos_id for i in range(B)]
                if not any(alive):
                    continue
                active.append(j)
                ids = torch.tensor([hyp[i][j] + [pad_id] * (T - len(hyp[i][j])) for i in range(B)])
                dec = decoder(P, cfg, ids, memory, mem_mask, ids != pad_id)
                dist = output_distribution(P, memory, mem_mask, dec)[:, step, :]
                dist = dist * torch.tensor([prob[i][j] for i in range(B)], dtype=torch.float32)[:, None]
                dist = dist.masked_fill(~torch.tensor(alive)[:, None], -1)
                blocks.append(dist)
            if not active:
                break
            done = [[j for j in range(beam) if hyp[i][j][-1] == eos_id] for i in range(B)]
            carried = torch.tensor([[prob[i][j] for j in done[i]] + [-1.0] * (beam - len(done[i])) for i in range(B)],
                                   dtype=torch.float32)
            allv = torch.cat(blocks + [carried], -1)
            top_p, top_i = torch.sort(allv, descending=True, dim=-1)
            top_p, top_i = top_p[:, :beam], top_i[:, :beam]
            if trace is not None:          # raw (pre copy-resolution) choices and their scores, per step
                trace.append((top_i.clone(), top_p.clone(), list(active)))
            new_hyp = []
            for i in range(B):
                row = []
                for j in range(beam):
                    which, tok = int(top_i[i, j]) // W, int(top_i[i, j]) % W
                    if which == len(active):
                        row.append(hyp[i][done[i][tok]])
                    else:
                        if tok >= V + L:
                            tok = int(sub_token[i, tok - V - L])
                        elif tok >= V:
                            tok = int(sou[i, tok - V])
                        row.append(hyp[i][active[which]] + [tok])
                new_hyp.append(row)
            hyp = new_hyp
            prob = top_p.numpy().tolist()
        return hyp, prob


def best_hypothesis(hyp, prob) -> List[List[int]]:
    """argmax-probability hypothesis per item, first on ties (reference run_model.py:351-352)."""
    out = []
    for h, p in zip(hyp, prob):
        best = max(range(len(p)), key=lambda j: (p[j], -j))
        out.append(h[best])
    return out
