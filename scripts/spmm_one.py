"""Run the GCN aggregation on a realistic batch-64 graph batch (padding skipped and not) for rocprofv3 --pmc passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import ops, data, synth
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import computed_nodes
cfg = FiraConfig()
store = data.process_raw(cfg, synth.generate_dataset(64, seed=0))
hb = store.batch(range(64))
for skip in (True, False):
    node_rows, rowptr, col, val, *_ = computed_nodes(hb, cfg, skip)
    rp, c, v = (torch.from_numpy(a).cuda() for a in (rowptr, col, val))
    X = torch.randn(len(node_rows), 256, device="cuda")
    for _ in range(5):
        ops.csr_spmm(rp, c, v, X)
    torch.cuda.synchronize()
    print("rows", len(node_rows), "nnz", len(col), "alg bytes", 4 * (len(node_rows) + 1) + 8 * len(col) + 2 * len(node_rows) * 1024)
