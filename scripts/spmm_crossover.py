"""Crossover of the aggregation kernels over graph density (SURVEY.md §7 "record the crossover as a
measured experiment"): 128 graphs x 512 nodes, adjacency density 0.3 .. 25 %, every variant of fira_csr_spmm:

    1  spmm_rowwave_kernel      CSR, wave per row, neighbour rows from L2
    2  spmm_lds_kernel          CSR, 64-column slab of the graph's features staged in LDS
    3  spmm_dense_f32_kernel    block-dense, fp32 MFMA (the reference's bmm arithmetic)
    4  spmm_dense_bf16_kernel   block-dense, bf16 MFMA (bf16 operands, fp32 accumulate: configs[2]'s dtype)

Feature buffers rotate over 3 (X, Y) sets = 402 MB > the 256 MiB Infinity Cache.  Algorithmic bytes per launch =
4(N+1) + 8 nnz + 2 N 256 4.  Prints a markdown table; python scripts/spmm_crossover.py > profiles/...md
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import ops  # noqa: E402
from fira_icse_amd.graphs import normalised_csr  # noqa: E402


def batch(B, n, density, seed):
    rng = np.random.default_rng(seed)
    n_pairs = max(1, int(density * n * n / 2))
    rp, cs, vs, base = [np.zeros(1, np.int64)], [], [], 0
    for b in range(B):
        k = rng.choice(n * (n - 1), size=n_pairs, replace=False)
        i, j = k // (n - 1), k % (n - 1)
        j = j + (j >= i)
        rowptr, col, val = normalised_csr(n, np.stack([i, j], 1))
        rp.append(rowptr[1:] + base); cs.append(col + b * n); vs.append(val)
        base += int(rowptr[-1])
    return np.concatenate(rp).astype(np.int32), np.concatenate(cs).astype(np.int32), np.concatenate(vs)


def time_variant(rp, c, v, Xs, Ys, N, variant, iters=12):
    k = [0]

    def fn():
        i = k[0] % len(Xs); k[0] += 1
        ops.csr_spmm(rp, c, v, Xs[i], graph_rows=N, variant=variant, out=Ys[i])
    for _ in range(len(Xs)):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    B, N = 128, 512
    Xs = [torch.randn(B * N, 256, device="cuda") for _ in range(3)]
    Ys = [torch.empty(B * N, 256, device="cuda") for _ in range(3)]
    names = {1: "CSR row-wave", 2: "CSR LDS slab", 3: "dense fp32 MFMA", 4: "dense bf16 MFMA"}
    print("| density | nnz / row | MB per launch | " + " | ".join("%s us (TB/s)" % names[v] for v in (1, 2, 3, 4)) +
          " | best fp32 | best any |")
    print("|---|---|---|---|---|---|---|---|---|")
    for dens in (0.003, 0.01, 0.03, 0.06, 0.12, 0.25):
        rp, c, v = (torch.from_numpy(a).cuda() for a in batch(B, N, dens, seed=int(dens * 1000)))
        nnz = int(c.numel())
        by = 4 * (B * N + 1) + 8 * nnz + 2 * B * N * 256 * 4
        ref = ops.csr_spmm(rp, c, v, Xs[0], graph_rows=N, variant=1)
        t = {}
        for var in (1, 2, 3, 4):
            t[var] = time_variant(rp, c, v, Xs, Ys, N, var)
            err = float((ops.csr_spmm(rp, c, v, Xs[0], graph_rows=N, variant=var) - ref).norm() / ref.norm())
            assert err < (1e-5 if var != 4 else 1e-2), (var, err)
        cells = " | ".join("%.1f (%.2f)" % (t[var] * 1e6, by / t[var] / 1e12) for var in (1, 2, 3, 4))
        best32 = min((1, 2, 3), key=lambda q: t[q])
        best = min((1, 2, 3, 4), key=lambda q: t[q])
        print("| %.1f %% | %.1f | %.1f | %s | %s | %s |" % (100.0 * nnz / (B * N * N), nnz / (B * N), by / 1e6, cells,
                                                           names[best32], names[best]), flush=True)


if __name__ == "__main__":
    main()
