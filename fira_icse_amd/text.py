"""Token-id -> message text, exactly as the reference driver writes ``OUTPUT/output_fira``
(reference run_model.py:342-372; dev variant run_model.py:138-177)."""
from __future__ import annotations

from typing import Dict, List, Sequence


def detokenize(ids: Sequence[int], r_vocab: Dict[int, str], var_map: Dict[str, str]) -> str:
    """ids (vocab ids, copy choices already resolved) -> output line.

    <start>/<eos>/<pad> are dropped by *string replacement* on the joined sentence, <unkm> becomes the emoji
    the reference uses, and placeholders are mapped back to the original identifiers (``var_map`` is the
    commit's ``variable.json`` entry: original -> placeholder).
    """
    s = " ".join(r_vocab[int(i)] for i in ids)
    s = s.replace("<start>", "").replace("<eos>", "").replace("<pad>", "").replace("<unkm>", "\U0001F605").strip()
    toks = s.split()
    back = {v: k for k, v in var_map.items()}
    return " ".join(back.get(t, t) for t in toks)


def resolve_copy(tok: int, sou_row: Sequence[int], sub_row: Sequence[int], V: int, L: int) -> int:
    """Output index -> vocabulary id (reference run_model.py:334-338)."""
    if tok >= V + L:
        return int(sub_row[tok - V - L])
    if tok >= V:
        return int(sou_row[tok - V])
    return int(tok)


def dev_sentence(ids: List[int], sou_row, sub_row, V: int, L: int, eos: int = 1) -> List[int]:
    """Teacher-forced output row -> vocab ids up to the first <eos> (reference run_model.py:149-158)."""
    if eos in ids:
        ids = ids[:ids.index(eos)]
    return [resolve_copy(t, sou_row, sub_row, V, L) for t in ids]
