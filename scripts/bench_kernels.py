"""Micro-benchmarks of the two roofline kernels (run on the GPU box): CSR SpMM (HBM) and the fp32 MFMA GEMM."""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import ops, graphs, data, synth  # noqa: E402
from fira_icse_amd.config import FiraConfig  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def spmm_bytes(n_rows, nnz, d=256):
    return 4 * (n_rows + 1) + 8 * nnz + 2 * n_rows * d * 4


def main():
    out = {}
    # config 5: dense stress graphs
    B, N = 128, 512
    rp, c, v = (torch.from_numpy(a).cuda() for a in graphs.dense_stress_batch(B, N))
    X = torch.randn(B * N, 256, device="cuda")
    for variant in (1, 2):
        t = timeit(lambda: ops.csr_spmm(rp, c, v, X, graph_rows=N, variant=variant))
        by = spmm_bytes(B * N, c.numel())
        out["spmm_cfg5_v%d" % variant] = dict(us=t * 1e6, alg_GBs=by / t / 1e9, nnz=int(c.numel()),
                                              gather_GBs=(8 * c.numel() + c.numel() * 1024 + B * N * 1024) / t / 1e9)
    # realistic graphs at batch 64
    cfg = FiraConfig()
    store = data.process_raw(cfg, synth.generate_dataset(64, seed=0))
    hb = store.batch(range(64))
    rp, c, v = (torch.from_numpy(a).cuda() for a in (hb.rowptr, hb.col, hb.val))
    X = torch.randn(64 * 650, 256, device="cuda")
    for variant in (1,):
        t = timeit(lambda: ops.csr_spmm(rp, c, v, X, graph_rows=650, variant=variant))
        out["spmm_real_b64_v%d" % variant] = dict(us=t * 1e6, alg_GBs=spmm_bytes(64 * 650, c.numel()) / t / 1e9,
                                                  nnz=int(c.numel()))
    # GEMMs
    for name, (M, Nn, K, tA, tB) in {
        "enc_fc_b32": (20800, 256, 256, False, True), "enc_fc_b64": (41600, 256, 256, False, True),
        "out_fc_b64": (1920, 24650, 256, False, True), "dec_qkv_b64": (1920, 768, 256, False, True),
        "kv_all_b64": (23680, 3072, 256, False, True), "square4k": (4096, 4096, 4096, False, True),
        "wgrad_gcn_b32": (256, 256, 20800, True, False),
    }.items():
        A = torch.randn((K, M) if tA else (M, K), device="cuda")
        Bm = torch.randn((Nn, K) if tB else (K, Nn), device="cuda")
        C = torch.zeros(M, Nn, device="cuda")
        sk = 24 if name.startswith("wgrad") else 1
        t = timeit(lambda: ops.gemm(A, Bm, transA=tA, transB=tB, out=C, accumulate=sk > 1, splitk=sk))
        out["gemm_" + name] = dict(us=t * 1e6, TFs=2.0 * M * Nn * K / t / 1e12)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
