"""Stand-alone timing of the generator projection [R, 24650] x K = 256: fira_head_logits_x3 (three bf16 terms) against the fp32
tiled kernel behind ops.gemm; error of both against fp64."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fira_icse_amd import ops


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    V = 24650
    g = torch.Generator().manual_seed(0)
    W = (torch.randn(V, 256, generator=g) * 0.06).cuda()
    b = (torch.randn(V, generator=g) * 0.1).cuda()
    for R in (64, 416, 530, 1000, 2700):
        x = torch.randn(R, 256, generator=g).cuda()
        ref = x.double() @ W.double().t() + b.double()
        got = ops.head_logits_x3(x, W, b)
        f32 = ops.gemm(x, W, bias=b)
        err = lambda a: float((a.double() - ref).norm() / ref.norm())
        t3 = timed(lambda: ops.head_logits_x3(x, W, b))
        t32 = timed(lambda: ops.gemm(x, W, bias=b))
        print("R %5d: x3 %.1f us (incl. the split launch and the output allocation)  fp32 %.1f us   rel err x3 %.2e  fp32 %.2e"
              % (R, t3, t32, err(got), err(f32)), flush=True)


if __name__ == "__main__":
    main()
