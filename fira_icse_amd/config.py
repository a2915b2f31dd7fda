"""Hyper-parameters of the FIRA hot path.

Mirrors the single hard-coded ``args`` dictionary of the reference driver
(reference run_model.py:27-46): the lengths, widths and optimiser settings are
the reference defaults; everything the BASELINE configs vary (batch, beam,
data dir, split sizes, world size) is an explicit field here instead of a
module-level constant.
"""
from __future__ import annotations

import dataclasses


@dataclasses.dataclass
class FiraConfig:
    # sequence / graph geometry (reference run_model.py:31-35)
    sou_len: int = 210          # code-token nodes (incl. <start>/<eos>)
    tar_len: int = 30           # message positions
    att_len: int = 25           # per-token sub-token list width (produced, unused by the model)
    ast_change_len: int = 280   # AST + edit-operation nodes
    sub_token_len: int = 160    # sub-token nodes
    # optimiser / model (reference run_model.py:36-39)
    lr: float = 1e-4
    dropout_rate: float = 0.1   # the GCN hard-codes 0.2 (reference gnn_transformer.py:43)
    num_head: int = 8
    embedding_dim: int = 256
    num_layers: int = 6         # hard-coded "range(6)" in the reference
    # run control (reference run_model.py:40-43)
    batch_size: int = 170
    test_batch_size: int = 20
    epoches: int = 150
    beam_size: int = 3
    # filled from the vocab files (reference run_model.py:53,56)
    vocab_size: int = 24650
    ast_change_vocab_size: int = 71

    # ---- derived ----
    @property
    def graph_len(self) -> int:
        return self.sou_len + self.sub_token_len + self.ast_change_len

    @property
    def mem_len(self) -> int:
        return self.sou_len + self.sub_token_len

    @property
    def out_len(self) -> int:
        return self.vocab_size + self.mem_len

    @property
    def d_head(self) -> int:
        return self.embedding_dim // self.num_head

    # the reference reads its settings through attribute *and* item access
    def __getitem__(self, k):
        return getattr(self, k)

    def replace(self, **kw) -> "FiraConfig":
        return dataclasses.replace(self, **kw)


PAD, EOS, START, UNK = 0, 1, 2, 3   # ids 0-3 of word_vocab.json
