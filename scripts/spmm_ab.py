"""Stand-alone timing of the row-gather aggregation kernels on realistic graph batches, over rotating buffer sets larger than the
Infinity Cache (as bench.py's spmm_b64 / spmm_b64_compact): variant 1 (four rows per wave, batched requests; round 5) against
variant 5 (one row per wave; rounds 1-4).  python scripts/spmm_ab.py [batch]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import ops, data, synth
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import DeviceBatch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = FiraConfig()
store = data.process_raw(cfg, synth.generate_dataset(B, seed=1000))
hb = store.batch(range(B))
rp, c, v = (torch.from_numpy(x).cuda() for x in (hb.rowptr, hb.col, hb.val))
db = DeviceBatch(hb, cfg, "cuda")


def rotating(n_rows, rp_, c_, v_, graph_rows, variant):
    per_set = 2 * n_rows * 256 * 4
    n_sets = max(4, -(-300 * (1 << 20) // per_set))
    Xs = [torch.randn(n_rows, 256, device="cuda") for _ in range(n_sets)]
    Ys = [torch.empty(n_rows, 256, device="cuda") for _ in range(n_sets)]
    k = [0]

    def fn():
        i = k[0] % n_sets
        k[0] += 1
        ops.csr_spmm(rp_, c_, v_, Xs[i], graph_rows=graph_rows, variant=variant, out=Ys[i])
    return bench.time_gpu(fn, iters=3 * n_sets, warmup=n_sets)


for name, n_rows, a, gr in (("dense layout", B * cfg.graph_len, (rp, c, v), cfg.graph_len), ("computed rows", db.n_nodes, (db.rowptr, db.col, db.val), 0)):
    nnz = int(a[1].numel())
    by = bench.spmm_bytes(n_rows, nnz)
    for rep in range(2):
        for variant in (5, 1):
            t = rotating(n_rows, *a, gr, variant)
            print("batch %d %-13s rows %6d nnz %6d variant %d: %6.2f us  %6.0f GB/s  %.3f of 8 TB/s" %
                  (B, name, n_rows, nnz, variant, t * 1e6, by / t / 1e9, by / t / 1e9 / 8000.0), flush=True)

# floor: a plain row copy of the same feature matrix over the same rotating sets (torch's copy kernel)
n_rows = B * cfg.graph_len
per_set = 2 * n_rows * 256 * 4
n_sets = max(4, -(-300 * (1 << 20) // per_set))
Xs = [torch.randn(n_rows, 256, device="cuda") for _ in range(n_sets)]
Ys = [torch.empty(n_rows, 256, device="cuda") for _ in range(n_sets)]
k = [0]


def cp():
    i = k[0] % n_sets
    k[0] += 1
    Ys[i].copy_(Xs[i])


t = bench.time_gpu(cp, iters=3 * n_sets, warmup=n_sets)
print("batch %d row copy of %d rows (2 x %.1f MB): %.2f us = %.0f GB/s (%.3f of 8 TB/s)" % (B, n_rows, n_rows * 1024 / 1e6, t * 1e6, 2 * n_rows * 1024 / t / 1e9, 2 * n_rows * 1024 / t / 1e9 / 8000))
