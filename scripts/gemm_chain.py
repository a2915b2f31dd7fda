"""Cost of the decoder-sized GEMMs INSIDE a dependent chain (what the training step and the decode loop pay): every product
reads the previous product's output, 40 of them are captured into a hipGraph (no host launch cost) and replayed; the
figure is microseconds per product including the kernel boundary.  Checked against an fp64 product first.

    python scripts/gemm_chain.py [f32|bf16]        (FIRA_SMALL_GEMM=1 selects the round-2 fragment-load kernel)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import ops  # noqa: E402


def chain_us(fn, n_inner=40, reps=20):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n_inner):
                fn()
        for _ in range(3):
            g.replay()
        s.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(reps):
            g.replay()
        b.record(s)
        s.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * n_inner)


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
    torch.manual_seed(0)
    rows = []
    # (M, widths of the chain): x[M,w0] -> w1 -> w0 ...; forward = weights [N,K] (transB), dgrad = weights [K,N]
    cases = [(960, (256, 256)), (960, (256, 768)), (960, (256, 1024)), (1920, (256, 256)), (1920, (256, 768)),
             (1920, (256, 1024)), (5100, (256, 256)), (5100, (256, 1024)), (64, (256, 256)), (64, (256, 1024)),
             (192, (256, 256)), (60, (256, 256))]
    if len(sys.argv) > 2:                       # a tile-count override case: M values only
        cases = [(int(m), (256, 256)) for m in sys.argv[2].split(",")] + [(int(m), (256, 1024)) for m in sys.argv[2].split(",")]
    for M, (w0, w1) in cases:
        for transB in ((True,) if dtype == "bf16" else (True, False)):
            x0 = torch.randn(M, w0, device="cuda")
            x1 = torch.empty(M, w1, device="cuda")
            # W01: w0 -> w1, W10: w1 -> w0, stored [N,K] (forward) or [K,N] (data gradient)
            W01 = torch.randn((w1, w0) if transB else (w0, w1), device="cuda") / w0 ** 0.5
            W10 = torch.randn((w0, w1) if transB else (w1, w0), device="cuda") / w1 ** 0.5
            bias = torch.randn(w1, device="cuda")
            # correctness of one hop against fp64 (bf16: against the bf16-rounded operands)
            if dtype == "bf16":                 # the engine's path: fp32 activations x bf16 weight shadow [N,K]
                wb01, _ = ops.weight_shadow(W01)
                wb10, _ = ops.weight_shadow(W10)
                got = ops.gemm_wb(x0, wb01, bias=bias)
            else:
                got = ops.gemm(x0, W01, transB=transB, bias=bias, dtype=dtype)
            rnd = (lambda t: t.to(torch.bfloat16).double()) if dtype == "bf16" else (lambda t: t.double())
            ref = rnd(x0) @ (rnd(W01).t() if transB else rnd(W01)) + bias.double()
            err = float((got.double() - ref).norm() / ref.norm())
            assert err < (3e-6 if dtype == "f32" else 1e-5), (M, w0, w1, transB, err)

            def hop():
                if dtype == "bf16":
                    ops.gemm_wb(x0, wb01, out=x1)
                    ops.gemm_wb(x1, wb10, out=x0)
                else:
                    ops.gemm(x0, W01, transB=transB, out=x1, dtype=dtype)
                    ops.gemm(x1, W10, transB=transB, out=x0, dtype=dtype)
            us = chain_us(hop, n_inner=20) / 2          # two products per hop
            fl = 2.0 * M * w0 * w1
            rows.append((M, w0, w1, "NT" if transB else "NN", us, fl / us / 1e6, err))
    print("| M | K<->N | layout | us per product (in chain) | TFLOP/s | rel err |")
    print("|---|---|---|---|---|---|")
    for M, w0, w1, lay, us, tf, err in rows:
        print("| %d | %d<->%d | %s | %.2f | %.1f | %.1e |" % (M, w0, w1, lay, us, tf, err))


if __name__ == "__main__":
    main()
