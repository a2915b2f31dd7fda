"""Host data layer: raw FIRA JSON graphs -> padded id arrays + batched CSR adjacency.

Drop-in for the reference's ``Dataset.TransDataset`` (reference Dataset.py:17-343)
on the input side: same cwd-relative ``DataSet/*.json`` files, same split rule,
same 8 per-commit arrays.  The one deliberate difference is the adjacency
layout: the reference keeps a ``scipy.sparse.coo_matrix`` per commit and
densifies it to a 650x650 float64 array on every ``__getitem__``
(Dataset.py:336-343); here the normalised adjacency is built once as CSR
(int32 rowptr/col, fp32 val rounded from the float64 value the reference
computes, Dataset.py:277-291 + gnn_transformer.py:80) and is never densified on
the training path.  ``dense_edge()`` reproduces the reference's float64 matrix
for parity tests.

Behavioural spec followed: SURVEY.md Appendix A (derived from Dataset.py:96-334).
"""
from __future__ import annotations

import json
import math
import os
import random
from typing import Dict, List, Optional, Sequence

import numpy as np

from .config import FiraConfig

LEMMATIZATION = {"added": "add", "fixed": "fix", "removed": "remove",
                 "adding": "add", "fixing": "fix", "removing": "remove"}   # Dataset.py:15

RAW_FILES = ["difftoken", "diffatt", "diffmark", "msg", "variable", "change", "ast",
             "edge_change_code", "edge_change_ast", "edge_ast_code", "edge_ast"]


def _ids(tokens: Sequence[str], vocab: Dict[str, int], upper: set) -> List[int]:
    """Token -> id with the case rule and <unkm> fallback (Dataset.py:69-78)."""
    out = []
    for t in tokens:
        if t not in upper:
            t = t.lower()
        out.append(vocab[t] if t in vocab else vocab["<unkm>"])
    return out


def _fit(seq: List[int], n: int, pad: int = 0) -> List[int]:
    """Pad with ``pad`` or truncate to exactly n entries (Dataset.py:80-86)."""
    return seq + [pad] * (n - len(seq)) if len(seq) < n else seq[:n]


class CommitGraph:
    """Processed form of one commit: id arrays + symmetric-normalised adjacency in CSR."""
    __slots__ = ("sou", "tar", "attr", "mark", "ast_change", "tar_label", "sub_token",
                 "rowptr", "col", "val64")


def process_commit(cfg: FiraConfig, vocab: Dict[str, int], ast_vocab: Dict[str, int], upper: set,
                   raw_diff: List[str], raw_att: List[List[str]], raw_mark: List[int], raw_msg: List[str],
                   var_map: Dict[str, str], raw_change: List[str], raw_ast: List[str],
                   e_change_code, e_change_ast, e_ast_code, e_ast) -> CommitGraph:
    V = len(vocab)
    L, S, A = cfg.sou_len, cfg.sub_token_len, cfg.ast_change_len
    N = L + S + A

    # 1. token normalisation: variable placeholders, case rule, message lemmatisation (Dataset.py:125-137)
    diff = []
    for t in raw_diff:
        t = var_map.get(t, t)
        diff.append(t if t in upper else t.lower())
    msg = []
    for t in raw_msg:
        t = var_map.get(t, t)
        t = t if t in upper else t.lower()
        msg.append(LEMMATIZATION.get(t, t))

    g = CommitGraph()
    # 2. code-token ids, marks, per-token sub-token ids (Dataset.py:139-166)
    g.sou = _fit([vocab["<start>"]] + _ids(diff, vocab, upper) + [vocab["<eos>"]], L)
    msg_ids = _ids(msg, vocab, upper)
    g.tar = _fit([vocab["<start>"]] + msg_ids + [vocab["<eos>"]], cfg.tar_len)
    for a in raw_att:
        for s in a:
            if not s.islower():
                raise AssertionError("sub-token %r is not lower-case" % (s,))          # Dataset.py:150
    att = [[]] + [_ids(a, vocab, upper) for a in raw_att] + [[]]
    mark = [2] + list(raw_mark) + [2]
    if not (len(diff) + 2 == len(att) == len(mark)):
        raise AssertionError("difftoken / diffatt / diffmark length mismatch")        # Dataset.py:159
    att = [_fit(a, cfg.att_len) for a in att]
    att = att + [[0] * cfg.att_len] * (L - len(att)) if len(att) < L else att[:L]
    g.attr = att
    g.mark = _fit(mark, L)

    # 4. AST + edit-operation node labels (Dataset.py:168-171); no <unkm> in that vocab -> KeyError like the reference
    g.ast_change = _fit(_ids(list(raw_ast) + list(raw_change), ast_vocab, upper), A)

    # 5. sub-token nodes and code<->sub-token edges (Dataset.py:173-196)
    sub_tokens: List[str] = []
    slots: Dict[str, List[int]] = {}
    e_sub = []
    for j, a in enumerate(raw_att):
        if not a:
            continue
        tok = diff[j]
        if tok in slots:
            if [sub_tokens[k] for k in slots[tok]] != a:
                raise AssertionError("token %r has two different sub-token lists" % (tok,))   # Dataset.py:184
            e_sub.extend((j, k) for k in slots[tok])
        else:
            base = len(sub_tokens)
            sub_tokens.extend(a)
            slots[tok] = list(range(base, base + len(a)))
            e_sub.extend((j, k) for k in slots[tok])
    g.sub_token = _fit(_ids(sub_tokens, vocab, upper), S)

    # 6. copy labels: whole-token copy wins over sub-token copy, first occurrence wins (Dataset.py:199-217)
    label = list(msg_ids)
    first_diff: Dict[str, int] = {}
    for j, t in enumerate(diff):
        first_diff.setdefault(t, j)
    first_sub: Dict[str, int] = {}
    for k, t in enumerate(sub_tokens):
        first_sub.setdefault(t, k)
    for k, t in enumerate(msg):
        if t in first_diff:
            label[k] = first_diff[t] + V + 1
    for k, t in enumerate(msg):
        if t in first_sub and label[k] < V:
            label[k] = first_sub[t] + V + L
    g.tar_label = _fit([vocab["<start>"]] + label + [vocab["<eos>"]], cfg.tar_len)

    # 7. merged undirected edge set in the reference's node index space (Dataset.py:220-266)
    n_ast = len(raw_ast)
    pairs = set()

    def link(p, q):
        pairs.add((p, q))
        pairs.add((q, p))

    for c, j in e_change_code:
        if j + 1 < L:
            link(c + L + S + n_ast, j + 1)
    for c, a in e_change_ast:
        link(c + L + S + n_ast, a + L + S)
    for a, j in e_ast_code:
        if j + 1 < L:
            link(a + L + S, j + 1)
    for a, b in e_ast:
        link(a + L + S, b + L + S)
    for j, k in e_sub:                      # unguarded in the reference (note N2)
        link(j + 1, k + L)
    for j in range(len(diff) + 1):          # sequential edges, unguarded (note N2)
        link(j, j + 1)
    for i in range(N):
        if (i, i) in pairs:
            raise AssertionError("raw edge (%d,%d) is a self-pair" % (i, i))          # Dataset.py:275
        pairs.add((i, i))
    rows = np.fromiter((p for p, _ in pairs), dtype=np.int64, count=len(pairs))
    cols = np.fromiter((q for _, q in pairs), dtype=np.int64, count=len(pairs))
    if rows.size and (rows.min() < 0 or cols.min() < 0 or rows.max() >= N or cols.max() >= N):
        raise ValueError("edge index out of the %d-node graph" % N)                  # scipy raises here too
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    deg = np.bincount(rows, minlength=N)     # the set is symmetric: row degree == column degree
    # value = 1/sqrt(deg_r)/sqrt(deg_c), evaluated in float64 exactly as Dataset.py:291 does
    val = np.array([1.0 / math.sqrt(deg[r]) / math.sqrt(deg[c]) for r, c in zip(rows.tolist(), cols.tolist())],
                   dtype=np.float64)
    rowptr = np.zeros(N + 1, dtype=np.int32)
    np.cumsum(deg, out=rowptr[1:])
    g.rowptr, g.col, g.val64 = rowptr, cols.astype(np.int32), val
    return g


class GraphStore:
    """All commits of one split, stored as flat arrays (ids) + one concatenated CSR."""

    def __init__(self, cfg: FiraConfig, graphs: List[CommitGraph]):
        self.cfg = cfg
        n = len(graphs)
        self.sou = np.array([g.sou for g in graphs], dtype=np.int64).reshape(n, cfg.sou_len)
        self.tar = np.array([g.tar for g in graphs], dtype=np.int64).reshape(n, cfg.tar_len)
        self.attr = np.array([g.attr for g in graphs], dtype=np.int64).reshape(n, cfg.sou_len, cfg.att_len)
        self.mark = np.array([g.mark for g in graphs], dtype=np.int64).reshape(n, cfg.sou_len)
        self.ast_change = np.array([g.ast_change for g in graphs], dtype=np.int64).reshape(n, cfg.ast_change_len)
        self.tar_label = np.array([g.tar_label for g in graphs], dtype=np.int64).reshape(n, cfg.tar_len)
        self.sub_token = np.array([g.sub_token for g in graphs], dtype=np.int64).reshape(n, cfg.sub_token_len)
        N = cfg.graph_len
        self.rowptr = np.stack([g.rowptr for g in graphs]).astype(np.int32) if n else np.zeros((0, N + 1), np.int32)
        self.nnz = self.rowptr[:, -1].astype(np.int64) if n else np.zeros(0, np.int64)
        self.offset = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(self.nnz, out=self.offset[1:])
        self.col = np.concatenate([g.col for g in graphs]) if n else np.zeros(0, np.int32)
        self.val64 = np.concatenate([g.val64 for g in graphs]) if n else np.zeros(0, np.float64)
        self.val = self.val64.astype(np.float32)       # == edge.float() of the reference (gnn_transformer.py:80)

    def __len__(self):
        return self.sou.shape[0]

    def dense_edge(self, i: int) -> np.ndarray:
        """The reference's per-item float64 adjacency (Dataset.py:340); for parity checks only."""
        N = self.cfg.graph_len
        out = np.zeros((N, N), dtype=np.float64)
        lo, hi = self.offset[i], self.offset[i + 1]
        rows = np.repeat(np.arange(N), np.diff(self.rowptr[i]))
        out[rows, self.col[lo:hi]] = self.val64[lo:hi]
        return out

    def item(self, i: int):
        """The reference's 8-tuple for commit i (Dataset.py:336-343)."""
        return [self.sou[i], self.tar[i], self.attr[i], self.mark[i], self.ast_change[i],
                self.dense_edge(i), self.tar_label[i], self.sub_token[i]]

    def batch(self, idx: Sequence[int]) -> "HostBatch":
        """Collate commits ``idx`` into one block-diagonal CSR batch (replaces the reference's dense
        ``.toarray()`` + default collate, Dataset.py:336-343).  Vectorised: no per-commit Python loop."""
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        B, N = len(idx), self.cfg.graph_len
        from . import _lib
        if B and os.environ.get("FIRA_HOST_LISTS", "native") != "numpy" and _lib.host_lib() is not None:
            # one C++ pass (csrc/hostlists.cpp: fira_host_collate_csr); the numpy statement below is its specification
            # (tests/test_host_lists.py compares them) and what runs when the library cannot be loaded on a CPU-only box
            total = int(self.nnz[idx].sum())
            rowptr = np.empty(B * N + 1, dtype=np.int32)
            col = np.empty(max(total, 1), dtype=np.int32)
            val = np.empty(max(total, 1), dtype=np.float32)
            ap = lambda a: a.ctypes.data
            _lib.check(_lib.lib().fira_host_collate_csr(B, N, ap(idx), len(self), ap(self._rowptr_c()), ap(self.offset),
                                                        ap(self._col_c()), ap(self._val_c()), ap(rowptr), ap(col), ap(val)),
                       "fira_host_collate_csr")
            return HostBatch(self.sou[idx], self.tar[idx], self.mark[idx], self.ast_change[idx], self.tar_label[idx],
                             self.sub_token[idx], rowptr, col[:total], val[:total], attr=lambda: self.attr[idx])
        nnz = self.nnz[idx]
        base = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(nnz, out=base[1:])
        total = int(base[-1])
        rowptr = np.empty(B * N + 1, dtype=np.int32)
        rowptr[:B * N] = (self.rowptr[idx, :-1].astype(np.int64) + base[:-1, None]).reshape(-1)
        rowptr[B * N] = total
        # entry k of the batch comes from store entry src[k]; its column moves into graph b's node block
        src = np.repeat(self.offset[idx] - base[:-1], nnz) + np.arange(total, dtype=np.int64)
        col = (self.col[src].astype(np.int64) + np.repeat(np.arange(B, dtype=np.int64) * N, nnz)).astype(np.int32)
        val = self.val[src]
        return HostBatch(self.sou[idx], self.tar[idx], self.mark[idx], self.ast_change[idx],
                         self.tar_label[idx], self.sub_token[idx], rowptr, col, val, self.attr[idx])

    # contiguous views of the store arrays in the dtypes the C helper reads (made once)
    def _rowptr_c(self):
        if getattr(self, "_rp_c", None) is None:
            self._rp_c = np.ascontiguousarray(self.rowptr, dtype=np.int32)
        return self._rp_c

    def _col_c(self):
        if getattr(self, "_cl_c", None) is None:
            self._cl_c = np.ascontiguousarray(self.col, dtype=np.int32)
        return self._cl_c

    def _val_c(self):
        if getattr(self, "_vl_c", None) is None:
            self._vl_c = np.ascontiguousarray(self.val, dtype=np.float32)
        return self._vl_c

    # ---- cache ----
    def save(self, path: str) -> None:
        np.savez_compressed(path, sou=self.sou, tar=self.tar, attr=self.attr.astype(np.int32), mark=self.mark,
                            ast_change=self.ast_change, tar_label=self.tar_label, sub_token=self.sub_token,
                            rowptr=self.rowptr, col=self.col, val64=self.val64)

    @classmethod
    def load(cls, cfg: FiraConfig, path: str) -> "GraphStore":
        z = np.load(path)
        self = cls.__new__(cls)
        self.cfg = cfg
        for k in ("sou", "tar", "mark", "ast_change", "tar_label", "sub_token", "rowptr", "col", "val64"):
            setattr(self, k, z[k])
        self.attr = z["attr"].astype(np.int64)
        self.nnz = self.rowptr[:, -1].astype(np.int64)
        self.offset = np.zeros(len(self.nnz) + 1, dtype=np.int64)
        np.cumsum(self.nnz, out=self.offset[1:])
        self.val = self.val64.astype(np.float32)
        return self


class HostBatch:
    """One collated batch on the host: int64 id arrays (reference dtypes) + block-diagonal CSR."""

    def __init__(self, sou, tar, mark, ast_change, tar_label, sub_token, rowptr, col, val, attr=None):
        self.sou, self.tar, self.mark, self.ast_change = sou, tar, mark, ast_change
        self.tar_label, self.sub_token = tar_label, sub_token
        self.rowptr, self.col, self.val = rowptr, col, val
        self._attr = attr                      # array, or a callable that gathers it on first use (the model never reads it)

    @property
    def attr(self):
        if callable(self._attr):
            self._attr = self._attr()
        return self._attr

    def __len__(self):
        return self.sou.shape[0]

    def dense_edge(self, graph_len: int) -> np.ndarray:
        """[B,N,N] float64 adjacency as the reference's collate would produce it (parity checks only)."""
        B = len(self)
        out = np.zeros((B, graph_len, graph_len), dtype=np.float64)
        rows = np.repeat(np.arange(B * graph_len), np.diff(self.rowptr))
        b = rows // graph_len
        out[b, rows % graph_len, self.col - b * graph_len] = self.val.astype(np.float64)
        return out


def load_raw(root: str) -> dict:
    raw = {}
    for k in RAW_FILES:
        with open(os.path.join(root, "DataSet", k + ".json")) as f:
            raw[k] = json.load(f)
    n = len(raw["difftoken"])
    if any(len(raw[k]) != n for k in RAW_FILES):
        raise AssertionError("raw DataSet files differ in length")                    # Dataset.py:42
    with open(os.path.join(root, "DataSet", "word_vocab.json")) as f:
        raw["word_vocab"] = json.load(f)
    with open(os.path.join(root, "DataSet", "ast_change_vocab.json")) as f:
        raw["ast_change_vocab"] = json.load(f)
    with open(os.path.join(root, "VOCAB_UPPER_CASE")) as f:
        raw["VOCAB_UPPER_CASE"] = json.load(f)
    return raw


def process_raw(cfg: FiraConfig, raw: dict, progress: bool = False) -> GraphStore:
    vocab, ast_vocab, upper = raw["word_vocab"], raw["ast_change_vocab"], set(raw["VOCAB_UPPER_CASE"])
    n = len(raw["difftoken"])
    it = range(n)
    if progress:
        from tqdm import tqdm
        it = tqdm(it)
    graphs = [process_commit(cfg, vocab, ast_vocab, upper, raw["difftoken"][i], raw["diffatt"][i],
                             raw["diffmark"][i], raw["msg"][i], raw["variable"][i], raw["change"][i],
                             raw["ast"][i], raw["edge_change_code"][i], raw["edge_change_ast"][i],
                             raw["edge_ast_code"][i], raw["edge_ast"][i]) for i in it]
    return GraphStore(cfg, graphs)


def split_index(n_train: int, n_valid: int, n_test: int, seed: int = 0) -> Dict[str, List[int]]:
    """The reference's split: python ``random`` seeded by the CLI, one shuffle (Dataset.py:306-313)."""
    index = list(range(n_train + n_valid + n_test))
    random.Random(seed).shuffle(index)
    return {"train": index[:n_train], "valid": index[n_train:n_train + n_valid],
            "test": index[n_train + n_valid:]}


class TransDataset:
    """Drop-in for reference ``TransDataset(args, split)`` (Dataset.py:17-67).

    Reads ``<root>/DataSet/*.json``, ``<root>/VOCAB_UPPER_CASE``; writes
    ``<root>/all_index`` and an ``.npz`` CSR cache per split on first use (the
    reference writes ``processed_<split>.pkl``).  Split sizes default to the
    reference's hard-coded 75000/8000/7661 (Dataset.py:10-12) when the data has
    exactly that many commits, otherwise they must be given.
    """

    def __init__(self, cfg: FiraConfig, data_name: str, root: str = ".",
                 splits: Optional[Sequence[int]] = None, seed: int = 0, progress: bool = False,
                 build: bool = True, wait_s: float = 24 * 3600.0):
        """``build=False`` (ranks other than 0 of a multi-process run): never build the cache, wait for the rank that
        does -- by polling the file system, so no collective with a timeout is involved (the first-run build is a pure
        Python pass over ~90 k commits and can take many minutes)."""
        self.cfg, self.data_name, self.root = cfg, data_name, root
        cache = os.path.join(root, "fira_cache_%s.npz" % data_name)
        # what the cache depends on besides the raw files: a cache built with other splits / seed / lengths is stale
        meta = json.dumps({"splits": list(splits) if splits is not None else None, "seed": int(seed),
                           "lens": [cfg.sou_len, cfg.tar_len, cfg.att_len, cfg.ast_change_len, cfg.sub_token_len],
                           "vocab": [cfg.vocab_size, cfg.ast_change_vocab_size]}, sort_keys=True)
        meta_path = os.path.join(root, "fira_cache_meta.json")

        def fresh():
            if not (os.path.exists(cache) and os.path.exists(meta_path)):
                return False
            try:
                with open(meta_path) as f:
                    have = json.load(f)
            except (OSError, ValueError):
                return False
            want = json.loads(meta)
            if splits is None:                           # split sizes not specified: any cached split serves
                want["splits"] = have.get("splits")
            return have == want

        if not build:
            import time
            t0 = time.time()
            while not fresh():
                if time.time() - t0 > wait_s:
                    raise TimeoutError("no fresh %s after %.0f s" % (cache, wait_s))
                time.sleep(0.5)
        elif not fresh():
            raw = load_raw(root)
            n = len(raw["difftoken"])
            if splits is None:
                splits = (75000, 8000, 7661)
            if sum(splits) != n:
                raise ValueError("split sizes %r do not add up to the %d commits in DataSet/" % (tuple(splits), n))
            full = process_raw(cfg, raw, progress)
            all_index = split_index(*splits, seed=seed)
            with open(os.path.join(root, "all_index"), "w") as f:
                json.dump(all_index, f)
            if os.path.exists(meta_path):
                os.remove(meta_path)                     # the marker goes first and comes back last
            for name in ("train", "valid", "test"):
                sub = _subset(full, all_index[name])
                sub.save(os.path.join(root, "fira_cache_%s.npz" % name))
            with open(meta_path + ".tmp", "w") as f:
                f.write(meta)
            os.replace(meta_path + ".tmp", meta_path)
        self.store = GraphStore.load(cfg, cache)

    def __len__(self):
        return len(self.store)

    def __getitem__(self, i):
        return self.store.item(i)


def _subset(full: GraphStore, idx: Sequence[int]) -> GraphStore:
    idx = np.asarray(idx, dtype=np.int64)
    sub = GraphStore.__new__(GraphStore)
    sub.cfg = full.cfg
    for k in ("sou", "tar", "attr", "mark", "ast_change", "tar_label", "sub_token", "rowptr"):
        setattr(sub, k, getattr(full, k)[idx])
    sub.nnz = sub.rowptr[:, -1].astype(np.int64)
    sub.offset = np.zeros(len(idx) + 1, dtype=np.int64)
    np.cumsum(sub.nnz, out=sub.offset[1:])
    if len(idx):
        sub.col = np.concatenate([full.col[full.offset[i]:full.offset[i + 1]] for i in idx])
        sub.val64 = np.concatenate([full.val64[full.offset[i]:full.offset[i + 1]] for i in idx])
    else:
        sub.col, sub.val64 = np.zeros(0, np.int32), np.zeros(0, np.float64)
    sub.val = sub.val64.astype(np.float32)
    return sub


def iterate_batches(n: int, batch_size: int, shuffle: bool):
    """Index batches in the order ``torch.utils.data.DataLoader`` would visit them.

    The reference builds ``DataLoader(train_set, batch_size, shuffle=True)``
    (run_model.py:387): its permutation comes from the global torch RNG at the
    moment iteration starts.  Driving a real DataLoader over bare indices keeps
    that RNG consumption identical, so a seeded run visits commits in the same
    order as the reference would.
    """
    import torch
    from torch.utils.data import DataLoader
    for idx in DataLoader(range(n), batch_size=batch_size, shuffle=shuffle):
        yield idx.tolist() if isinstance(idx, torch.Tensor) else list(idx)
