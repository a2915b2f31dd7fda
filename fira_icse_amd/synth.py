"""Synthetic FIRA DataSet generator (raw JSON schema of the reference).

The per-commit data of the reference (``DataSet/*.json``) is not redistributed,
so benchmarks and parity tests run on commits synthesised in the same raw
schema the reference data layer consumes (reference Dataset.py:30-44; schema
recovered from Dataset.py:96-266).  Every file is a JSON list with one entry
per commit; see SURVEY.md §8(d) for the constraints each entry obeys.

Nothing here is on the timed path; it only manufactures inputs.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List

import numpy as np

SPECIALS = ["<pad>", "<eos>", "<start>", "<unkm>"]
CHANGE_OPS = ["update", "delete", "add", "move", "match"]
LEMMA_SOURCES = ["added", "fixed", "removed", "adding", "fixing", "removing"]


def make_vocab(vocab_size: int = 24650) -> Dict[str, int]:
    """A word vocabulary of the reference's size: ids 0-3 are the specials."""
    vocab = {t: i for i, t in enumerate(SPECIALS)}
    for w in ["fix", "to", "add", "remove", "the", "get", "set", "user", "name", "value",
              "<nb>", "<nl>", "NAMESPACE", "STRING0", "STRING1", "NUMBER0", "NUMBER1", "COMMENT"]:
        vocab[w] = len(vocab)
    i = 0
    while len(vocab) < vocab_size:
        vocab["w%05d" % i] = len(vocab)
        i += 1
    return vocab


def make_ast_change_vocab(size: int = 71) -> Dict[str, int]:
    vocab = {"<pad>": 0}
    for op in CHANGE_OPS:
        vocab[op] = len(vocab)
    i = 0
    while len(vocab) < size:
        vocab["asttype%02d" % i] = len(vocab)
        i += 1
    return vocab


def make_upper_case() -> List[str]:
    return ["NAMESPACE", "SINGLE", "COMMENT"] + ["STRING%d" % i for i in range(8)] + \
           ["NUMBER%d" % i for i in range(8)] + ["FLOAT%d" % i for i in range(4)]


def _camel(parts: List[str]) -> str:
    return parts[0] + "".join(p.capitalize() for p in parts[1:])


def generate_commit(rng: np.random.Generator, words: List[str], ast_types: List[str], upper: List[str],
                    n_d: int, n_msg: int, n_a: int, n_c: int) -> dict:
    """One commit in the raw schema. ``n_d`` may exceed 208 to exercise truncation."""
    # --- identifiers: camelCase tokens with 2-3 lower-case sub-tokens each ---
    n_ident = max(1, n_d // 12)
    idents = []
    for _ in range(n_ident):
        k = int(rng.integers(2, 4))
        parts = [words[int(rng.integers(0, len(words)))] for _ in range(k)]
        idents.append((_camel(parts), parts))
    ident_att = {}
    for name, parts in idents:
        ident_att.setdefault(name.lower(), (name, parts))
    idents = list(ident_att.values())

    # --- a couple of renamed variables (variable.json: original -> placeholder) ---
    var_map = {}
    n_var = int(rng.integers(0, 3))
    for v in range(n_var):
        var_map["origVar%d_%d" % (v, int(rng.integers(0, 1000)))] = "n%d" % v

    difftoken, diffmark, diffatt = [], [], []
    var_names = list(var_map.keys())
    for j in range(n_d):
        r = rng.random()
        if r < 0.20 and j < 200:   # (j+1, k+210) must never be a self-pair (reference Dataset.py:275)
            name, parts = idents[int(rng.integers(0, len(idents)))]
            difftoken.append(name)
            diffatt.append(list(parts))
        elif r < 0.25:
            difftoken.append(upper[int(rng.integers(0, len(upper)))])
            diffatt.append([])
        elif r < 0.28 and var_names:
            difftoken.append(var_names[int(rng.integers(0, len(var_names)))])
            diffatt.append([])
        elif r < 0.31:
            difftoken.append("<nl>" if rng.random() < 0.5 else "<nb>")
            diffatt.append([])
        elif r < 0.34:
            difftoken.append("Zz%dQ" % int(rng.integers(0, 50)))   # out-of-vocabulary -> <unkm>
            diffatt.append([])
        else:
            difftoken.append(words[int(rng.integers(0, len(words)))])
            diffatt.append([])
        diffmark.append(int(rng.choice([1, 2, 3], p=[0.25, 0.5, 0.25])))

    # --- message: some fresh words, some copied tokens, some copied sub-tokens ---
    msg = []
    all_sub = [p for _, parts in idents for p in parts]
    for k in range(n_msg):
        r = rng.random()
        if r < 0.25:
            msg.append(difftoken[int(rng.integers(0, min(n_d, 208)))])
        elif r < 0.40 and all_sub:
            msg.append(all_sub[int(rng.integers(0, len(all_sub)))])
        elif r < 0.47:
            msg.append(LEMMA_SOURCES[int(rng.integers(0, len(LEMMA_SOURCES)))])
        else:
            msg.append(words[int(rng.integers(0, len(words)))])

    ast = [ast_types[int(rng.integers(0, len(ast_types)))] for _ in range(n_a)]
    # type labels are case-folded by the data layer: emit some CamelCase
    ast = [a.capitalize() if rng.random() < 0.3 else a for a in ast]
    change = [CHANGE_OPS[int(rng.integers(0, len(CHANGE_OPS)))] for _ in range(n_c)]

    edge_ast = []
    for child in range(1, n_a):
        edge_ast.append([int(rng.integers(0, child)), child])            # random tree
    edge_ast_code = []
    for j in range(0, n_d, 2):
        if n_a:
            edge_ast_code.append([int(rng.integers(0, n_a)), j])
    edge_change_ast, edge_change_code = [], []
    for c in range(n_c):
        if n_a and rng.random() < 0.7:
            edge_change_ast.append([c, int(rng.integers(0, n_a))])
        if rng.random() < 0.7:
            edge_change_code.append([c, int(rng.integers(0, n_d))])
    return dict(difftoken=difftoken, diffmark=diffmark, diffatt=diffatt, msg=msg, variable=var_map,
                ast=ast, change=change, edge_ast=edge_ast, edge_ast_code=edge_ast_code,
                edge_change_ast=edge_change_ast, edge_change_code=edge_change_code)


RAW_FILES = ["difftoken", "diffatt", "diffmark", "msg", "variable", "change", "ast",
             "edge_change_code", "edge_change_ast", "edge_ast_code", "edge_ast"]


def generate_dataset(n_commits: int, seed: int = 0, vocab_size: int = 24650, ast_vocab_size: int = 71,
                     overlong_every: int = 0) -> dict:
    """Raw DataSet as a dict of python lists (+ vocabularies).

    ``overlong_every`` > 0 makes every k-th commit exceed the un-truncated
    limits (n_d > 208, many sub-tokens) to pin the reference's truncation and
    unguarded-index behaviour (SURVEY.md §8a note N2).
    """
    rng = np.random.default_rng(seed)
    vocab = make_vocab(vocab_size)
    ast_vocab = make_ast_change_vocab(ast_vocab_size)
    upper = make_upper_case()
    words = [w for w in vocab if w.startswith("w") or w in ("fix", "to", "add", "remove", "the", "get", "set",
                                                            "user", "name", "value")]
    ast_types = [a for a in ast_vocab if a.startswith("asttype")]
    out = {k: [] for k in RAW_FILES}
    for i in range(n_commits):
        if overlong_every and i % overlong_every == overlong_every - 1:
            n_d = int(rng.integers(215, 240))
        else:
            n_d = int(rng.integers(20, 208))
        n_msg = int(rng.integers(3, 28))
        n_a = int(rng.integers(10, 250))
        n_c = int(rng.integers(0, 30))
        c = generate_commit(rng, words, ast_types, upper, n_d, n_msg, n_a, n_c)
        for k in RAW_FILES:
            out[k].append(c[k])
    out["word_vocab"] = vocab
    out["ast_change_vocab"] = ast_vocab
    out["VOCAB_UPPER_CASE"] = upper
    return out


def write_dataset(root: str, ds: dict) -> None:
    """Write ``ds`` under ``root`` in the reference's cwd-relative layout."""
    os.makedirs(os.path.join(root, "DataSet"), exist_ok=True)
    for k in RAW_FILES:
        with open(os.path.join(root, "DataSet", k + ".json"), "w") as f:
            json.dump(ds[k], f)
    with open(os.path.join(root, "DataSet", "word_vocab.json"), "w") as f:
        json.dump(ds["word_vocab"], f)
    with open(os.path.join(root, "DataSet", "ast_change_vocab.json"), "w") as f:
        json.dump(ds["ast_change_vocab"], f)
    with open(os.path.join(root, "VOCAB_UPPER_CASE"), "w") as f:
        json.dump(ds["VOCAB_UPPER_CASE"], f)
