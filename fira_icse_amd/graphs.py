"""Synthetic adjacency batches in the engine's CSR layout (inputs for benchmarks and full-size property tests)."""
from __future__ import annotations

import numpy as np


def normalised_csr(n: int, pairs: np.ndarray):
    """Symmetrise + dedupe + self-loops + D^-1/2 (A+I) D^-1/2, the merge rule of reference Dataset.py:220-291."""
    a = np.concatenate([pairs, pairs[:, ::-1], np.stack([np.arange(n), np.arange(n)], 1)])
    key = np.unique(a[:, 0].astype(np.int64) * n + a[:, 1])
    rows, cols = key // n, key % n
    deg = np.bincount(rows, minlength=n).astype(np.float64)
    val = (1.0 / np.sqrt(deg[rows]) / np.sqrt(deg[cols])).astype(np.float32)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=rowptr[1:])
    return rowptr, cols.astype(np.int32), val


def dense_stress_batch(B: int, n: int, n_types: int = 4, edges_per_type: int = 8192, seed: int = 0):
    """BASELINE.json config 5: B graphs of n nodes, n_types x edges_per_type random distinct non-self pairs each."""
    rng = np.random.default_rng(seed)
    rp, cs, vs, base = [np.zeros(1, np.int64)], [], [], 0
    for b in range(B):
        pairs = []
        for _ in range(n_types):
            k = rng.choice(n * (n - 1), size=edges_per_type, replace=False)
            i, j = k // (n - 1), k % (n - 1)
            j = j + (j >= i)                       # skip the diagonal
            pairs.append(np.stack([i, j], 1))
        rowptr, col, val = normalised_csr(n, np.concatenate(pairs))
        rp.append(rowptr[1:] + base)
        cs.append(col + b * n)
        vs.append(val)
        base += int(rowptr[-1])
    return np.concatenate(rp).astype(np.int32), np.concatenate(cs).astype(np.int32), np.concatenate(vs)
