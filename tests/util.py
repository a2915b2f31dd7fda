"""Shared helpers for the test-suite (seeded weights, fixtures, tolerances)."""
from __future__ import annotations

import gzip
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
REFERENCE = "/root/reference"
for p in (REPO,):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_N = 24                 # commits in the golden synthetic DataSet
GOLDEN_SPLIT = (16, 4, 4)     # train / valid / test
GOLDEN_B = 4                  # batch the golden model run uses (first commits of the split)
# fixtures at BASELINE batch sizes (tests/golden/make_golden_large.py): 128 train / 4 valid / 64 test commits
LARGE_SEED = 11
LARGE_SPLIT = (128, 4, 64)
LARGE_N = sum(LARGE_SPLIT)
# BASELINE configs[0] (tests/golden/make_golden_epoch.py): one epoch of the reference's train() at batch 4
EPOCH_SEED = 21
EPOCH_SPLIT = (96, 16, 16)
EPOCH_N = sum(EPOCH_SPLIT)
EPOCH_B = 4

GRAD_SAMPLE_KEYS = [
    "encoder.embedding.weight", "encoder.mark_embedding.weight", "encoder.ast_change_embedding.weight",
    "encoder.combination_list2.0.linear_layers.2.weight", "encoder.combination_list2.5.output_linear.bias",
    "encoder.gcn_list.0.fc1.weight", "encoder.gcn_list.5.layernorm.weight",
    "decoder.embedding.weight", "decoder.attention_list.0.fc_k.weight", "decoder.cross_attention_list.3.fc_v.bias",
    "decoder.feed_forward_list.5.fc1.weight", "out_fc.weight", "out_fc.bias",
    "copy_net.LinearSource.weight", "copy_net.LinearRes.weight", "copy_net.LinearProb.bias",
]


def has_reference() -> bool:
    return os.path.isdir(REFERENCE) and os.path.exists(os.path.join(REFERENCE, "Model.py"))


def perturb_state_dict(sd: dict, seed: int = 1) -> dict:
    """Make LayerNorm affine parameters non-trivial so parity checks are sensitive to them.

    Deterministic (own torch.Generator); applied identically to the reference model and to the engine.
    """
    g = torch.Generator().manual_seed(seed)
    for k in sorted(sd):
        if ".layernorm." in k:
            sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
    return sd


def peaked_state_dict(sd: dict, seed: int = 2) -> dict:
    """Weights whose output distributions are peaked, for decode parity.

    With random-init weights the reference's running probability *product* underflows fp32 to zero after
    ~13-16 steps and every candidate ties (SURVEY.md §7 "hard parts"); sharpening the generator / copy
    heads keeps the per-step maximum well separated so that token choices are decided by the model, not
    by the sort's tie-breaking.  The <eos> bias makes hypotheses end at varied lengths.
    """
    sd = perturb_state_dict(sd, seed)
    sd["out_fc.weight"] = sd["out_fc.weight"] * 24.0
    sd["out_fc.bias"] = sd["out_fc.bias"].clone()
    sd["out_fc.bias"][1] += 52.0
    sd["copy_net.LinearRes.weight"] = sd["copy_net.LinearRes.weight"] * 24.0
    sd["copy_net.LinearProb.bias"] = sd["copy_net.LinearProb.bias"].clone()
    sd["copy_net.LinearProb.bias"][1] += 1.0
    return sd


TIE_TWINS = (4, 24650)      # token ids [lo, hi): every odd id in the range is an exact copy of the even id below it


def tie_state_dict(sd: dict, seed: int = 2) -> dict:
    """peaked_state_dict + twin vocabulary entries: for every even id e in TIE_TWINS the generator row, its bias and the
    decoder embedding row of e + 1 are bit-copies of e's.  Hypotheses that differ only in a twin have EXACTLY equal
    probabilities at every later step, so beam search must break ties -- the reference does it with
    ``torch.sort(descending=True)`` over the flattened candidates (run_model.py:305)."""
    sd = peaked_state_dict(sd, seed)
    lo, hi = TIE_TWINS
    for k in ("out_fc.weight", "out_fc.bias", "decoder.embedding.weight"):
        t = sd[k].clone()
        t[lo + 1:hi:2] = t[lo:hi:2]
        sd[k] = t
    return sd


def load_golden_raw():
    """The golden synthetic raw DataSet (regenerated deterministically from the committed generator)."""
    from fira_icse_amd import synth
    return synth.generate_dataset(GOLDEN_N, seed=0, overlong_every=6)


def golden_npz(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def to_torch_batch(hb, cfg, dense_edge: bool = True):
    """HostBatch -> the reference's 8 tensors (attr is a zero placeholder: the model ignores it)."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    edge = t(hb.dense_edge(cfg.graph_len)) if dense_edge else None
    return dict(sou=t(hb.sou), tar=t(hb.tar), mark=t(hb.mark), ast_change=t(hb.ast_change), edge=edge,
                tar_label=t(hb.tar_label), sub_token=t(hb.sub_token))


def edge_case_raw():
    """4 hand-made commits in the raw schema."""
    from fira_icse_amd import synth
    ds = synth.generate_dataset(4, seed=7)
    words = [w for w in ds["word_vocab"] if w.startswith("w0")][:40]
    # commit 0: no identifiers (no sub-token nodes), no AST, no edit operations
    ds["difftoken"][0] = words[:12]
    ds["diffatt"][0] = [[] for _ in range(12)]
    ds["diffmark"][0] = [2] * 12
    ds["ast"][0], ds["change"][0] = [], []
    for k in ("edge_ast", "edge_ast_code", "edge_change_ast", "edge_change_code"):
        ds[k][0] = []
    ds["msg"][0] = ["fix", words[20]]
    ds["variable"][0] = {}
    # commit 1: every message token is copied from the diff -> no target row needs the vocabulary GEMM
    ds["msg"][1] = [t for t in ds["difftoken"][1][:6]]
    # commit 2: maximum-length message (28 tokens + <start>/<eos> = 30 positions)
    ds["msg"][2] = [words[i % 40] for i in range(28)]
    # commit 3: over-long diff (truncated at 208 tokens; unguarded sequential edges)
    ds["difftoken"][3] = [words[i % 40] for i in range(230)]
    ds["diffatt"][3] = [[] for _ in range(230)]
    ds["diffmark"][3] = [2] * 230
    ds["edge_ast_code"][3] = [[0, j] for j in range(0, 230, 7)] if ds["ast"][3] else []
    ds["edge_change_code"][3] = [e for e in ds["edge_change_code"][3] if e[1] < 230]
    return ds
