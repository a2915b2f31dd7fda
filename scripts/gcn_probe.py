"""Stand-alone timing of one GCN layer at the engine's shapes: the fused launch (gcn_fused.hip) against the three / two
separate launches it replaces (CSR SpMM + [Nc,256]x[256,256] product + add-LayerNorm; product + SpMM).

    python scripts/gcn_probe.py [batch] [dtype: 0 fp32 | 1 bf16]

Uses a synthetic batch of FIRA-shaped graphs (fira_icse_amd.synth), the compact CSR the engine builds for it, and rotates
over enough feature buffers to leave the L2s.  Prints microseconds per launch (HIP events over 100 launches)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import data, ops, synth  # noqa: E402
from fira_icse_amd.config import FiraConfig  # noqa: E402
from fira_icse_amd.model import DeviceBatch  # noqa: E402


def timeit(fn, n=100):
    for _ in range(5):
        fn(0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dtype = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cfg = FiraConfig()
    store = data.process_raw(cfg, synth.generate_dataset(B, seed=7))
    db = DeviceBatch(store.batch(range(B)), cfg)
    db.wait_ready() if hasattr(db, "wait_ready") else None
    rowptr, col, val = db.rowptr[:db.n_nodes + 1], db.col[:db.nnz], db.val[:db.nnz]
    n = db.n_nodes
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(0)
    nbuf = 8
    Xs = [torch.randn(n, 256, generator=g).to(dev) for _ in range(nbuf)]
    W = (torch.randn(256, 256, generator=g) * 0.06).to(dev)
    Wt = W.t().contiguous()
    b2, c21 = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
    gamma, beta = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    dX = torch.zeros(n, 256, device=dev)
    print("batch %d: %d computed nodes, %d entries (%.2f per row), dtype %s" % (B, n, db.nnz, db.nnz / n, "bf16" if dtype else "fp32"))
    us = timeit(lambda i: ops.gcn_layer_fwd(rowptr, col, val, Xs[i % nbuf], Wt, b2, c21, gamma, beta, dropout=0.2, seed=1, site=3,
                                            dtype=dtype, want_rowsum=False))
    flop = 2.0 * n * 256 * 256
    print("fused forward : %6.1f us per launch  (%.1f TF/s on the product, %.2f TB/s on 3 rows in/out)" %
          (us, flop / us / 1e6, 3.0 * n * 1024 / us / 1e6))
    us = timeit(lambda i: ops.gcn_layer_bwd(rowptr, col, val, Xs[i % nbuf], W, dX, dtype=dtype))
    print("fused backward: %6.1f us per launch  (%.1f TF/s)" % (us, flop / us / 1e6))
    us = timeit(lambda i: ops.csr_spmm(rowptr, col, val, Xs[i % nbuf], variant=1))
    print("csr spmm      : %6.1f us per launch" % us)
    out = torch.empty(n, 256, device=dev)
    us = timeit(lambda i: ops.gemm(Xs[i % nbuf], W, out=out, dtype="bf16" if dtype else "f32"))
    print("product       : %6.1f us per launch  (%.1f TF/s)" % (us, flop / us / 1e6))


if __name__ == "__main__":
    main()
