"""LayerNorm-backward prologue of the data-gradient tile kernel (fira_ln_bwd_linear_f32) against the plain product of the same
shape: where do the extra microseconds go?  M = 500 / 1000 rows (batch 32 / 64), N = 256 / 768 / 1024, dropout 0 / 0.1."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fira_icse_amd import ops


def timed(fn, reps=50):
    for _ in range(5):
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
    g = torch.Generator().manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()
    for M in (500, 1000):
        for N in (256, 768, 1024):
            dy, summ = rn(M, 256), rn(M, 256)
            stats = torch.stack([summ.mean(1), 1.0 / torch.sqrt(summ.var(1, unbiased=False) + 1e-5)], 1).contiguous()
            gamma = rn(256)
            Wt = rn(256, N) * 0.06
            out = torch.empty(M, N, device="cuda")
            t_plain = timed(lambda: ops.gemm(dy, Wt, transB=False, out=out))
            t0 = timed(lambda: ops.ln_bwd_linear(dy, Wt, summ, stats, gamma, dropout=0.0))
            t1 = timed(lambda: ops.ln_bwd_linear(dy, Wt, summ, stats, gamma, dropout=0.1, seed=3, site=5))
            t_alloc = timed(lambda: (torch.empty((M, N), device="cuda"), torch.empty_like(dy), torch.empty_like(dy), torch.zeros(256, device="cuda"),
                                     torch.zeros(256, device="cuda"), torch.empty(((M + 31) // 32) * 512, device="cuda")))
            print("M %4d N %4d: plain product %.1f us   LN-bwd prologue p=0 %.1f us   p=0.1 %.1f us  (of which the wrapper's allocations + two fills + the closing reduce launch: %.1f us + a launch)"
                  % (M, N, t_plain, t0, t1, t_alloc), flush=True)


if __name__ == "__main__":
    main()
