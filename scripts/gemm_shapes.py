"""Time every GEMM shape of one training step (batch 32) in isolation, with the library's automatic tile/split choice
going through the same code path as the engine (fira_train_fwd_bwd uses gemm_f32_ex(splitk=0)); here splitk is given
explicitly so several choices can be compared."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import ops
from scripts.bench_kernels import timeit

def main():
    shapes = {  # name: (M, N, K, tA, tB, accumulate)
        "enc_fc fwd NT": (20800, 256, 256, 0, 1, 0), "enc_qk fwd NT": (6720, 512, 256, 0, 1, 0),
        "enc_fc dgrad NN": (20800, 256, 256, 0, 0, 0), "enc_fc wgrad TN": (256, 256, 20800, 1, 0, 1),
        "enc_qk wgrad TN": (512, 256, 6720, 1, 0, 1), "enc_o wgrad TN": (256, 256, 6720, 1, 0, 1),
        "kv_all fwd NT": (11840, 3072, 256, 0, 1, 0), "kv_all dgrad NN": (11840, 256, 3072, 0, 0, 1),
        "kv_all wgrad TN": (3072, 256, 11840, 1, 0, 1), "dec small fwd NT": (960, 256, 256, 0, 1, 0),
        "dec qkv fwd NT": (960, 768, 256, 0, 1, 0), "dec ffn2 fwd NT": (960, 256, 1024, 0, 1, 0),
        "dec wgrad TN": (256, 256, 960, 1, 0, 1), "out_fc fwd NT": (250, 24650, 256, 0, 1, 0),
        "out_fc wgrad TN": (24650, 256, 250, 1, 0, 1), "out_fc dgrad NN": (250, 256, 24650, 0, 0, 1),
    }
    out = {}
    for name, (M, N, K, tA, tB, acc) in shapes.items():
        A = torch.randn((K, M) if tA else (M, K), device="cuda")
        B = torch.randn((N, K) if tB else (K, N), device="cuda")
        C = torch.zeros(M, N, device="cuda")
        res = {}
        for tile in (1, 2, 3):
            if acc:
                continue
            t = timeit(lambda: ops.gemm(A, B, transA=bool(tA), transB=bool(tB), out=C, tile=tile))
            res["tile%d" % tile] = "%.1fus %.1fTF" % (t * 1e6, 2.0 * M * N * K / t / 1e12)
        for sk in ([1] if not acc else [16, 64, 128]):
            if sk > max(1, K // 64):
                continue
            t = timeit(lambda: ops.gemm(A, B, transA=bool(tA), transB=bool(tB), out=C, accumulate=bool(acc), splitk=sk))
            res["sk%d" % sk] = "%.1fus %.1fTF" % (t * 1e6, 2.0 * M * N * K / t / 1e12)
        out[name] = res
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
