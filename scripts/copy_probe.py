"""Stand-alone timing of the copy head's backward kernel (copyhead.hip: copy_score_bwd_kernel) at the training step's shape:
B commits x 370 memory slots, a few copy-labelled target rows per commit (only those carry a non-zero dscore row).

    python scripts/copy_probe.py [batch] [active rows per commit]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import ops  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    act = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    T, S = 30, 370
    g = torch.Generator().manual_seed(0)
    src = torch.randn(B, S, 256, generator=g).cuda()
    tgt = torch.randn(B, T, 256, generator=g).cuda()
    w = torch.randn(256, generator=g).cuda()
    ds = torch.zeros(B, T, S)
    for b in range(B):
        rows = torch.randperm(T, generator=g)[:act]
        ds[b, rows] = torch.randn(act, S, generator=g) * 0.01
    ds = ds.cuda()
    for _ in range(5):
        ops.copy_score_bwd(src, tgt, w, ds)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    n = 50
    for _ in range(n):
        ops.copy_score_bwd(src, tgt, w, ds)
    e.record()
    torch.cuda.synchronize()
    print("copy_score_bwd batch %d, %d active rows per commit: %.1f us per call (incl. its output allocations)" % (B, act, a.elapsed_time(e) * 1e3 / n))


if __name__ == "__main__":
    main()
