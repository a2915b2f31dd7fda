"""Config 5 (128 graphs x 512 nodes x 4 x 8192 edges): the aggregation variants timed over rotating buffers, checked against a
dense fp64 product on graph 0 and 127.  FIRA_SPMM_DENSE_PROBE (bits: 1 no densify, 2 one chunk of H, 4 no stores) is read by the
library when a probe build carries it."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
from fira_icse_amd import graphs, ops

B, N = 128, 512
rp_h, c_h, v_h = graphs.dense_stress_batch(B, N)
rp, c, v = (torch.from_numpy(x).cuda() for x in (rp_h, c_h, v_h))
Xs = [torch.randn(B * N, 256, device="cuda") for _ in range(3)]
Ys = [torch.empty(B * N, 256, device="cuda") for _ in range(3)]
by = 4 * (B * N + 1) + 8 * c.numel() + 2 * B * N * 1024
variants = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2,3,4").split(",")]
for variant in variants:
    k = [0]
    def fn():
        i = k[0] % 3; k[0] += 1
        ops.csr_spmm(rp, c, v, Xs[i], graph_rows=N, variant=variant, out=Ys[i])
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(12): fn()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 12 * 1e-3
    # check graphs 0 and B-1 of buffer set (k-1)%3
    i = (k[0] - 1) % 3
    err = 0.0
    for g in (0, B - 1):
        A = np.zeros((N, N))
        for r in range(N):
            lo, hi = rp_h[g * N + r], rp_h[g * N + r + 1]
            np.add.at(A[r], c_h[lo:hi] - g * N, v_h[lo:hi])
        ref = A @ Xs[i][g * N:(g + 1) * N].double().cpu().numpy()
        got = Ys[i][g * N:(g + 1) * N].double().cpu().numpy()
        err = max(err, float(np.abs(ref - got).max() / np.abs(ref).max()))
    print("variant %d  %.1f us  frac %.3f  max rel err %.2e" % (variant, t * 1e6, by / t / 8e12, err), flush=True)
