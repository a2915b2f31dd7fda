#!/usr/bin/env python
"""Benchmark of the FIRA hot path on MI355X: training commits/s (+ greedy-decode tokens/s), BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f32|bf16]

A "step" is one optimisation step of run_model.py:101-112 (forward + backward + [RCCL all-reduce] + Adam) over one
batch of synthetic commits in the reference's raw schema (there is no network for the real DataSet; seed-0 generator,
SURVEY.md §8d), with the reference's dropout (0.1 / 0.2) ON.  The default workload is BASELINE configs[1]
("1xMI355X, batch=32, fp32"); ``--dtype bf16`` is configs[2]'s per-GPU workload (batch 64 per GPU, nn.Linear products on
the bf16 MFMA with fp32 accumulation).  Every rank keeps its batch as N grows (weak scaling).  Inputs are resident in
HBM before the timed region (pre-collated device batches, cycled).  One JSON line is printed by rank 0.

``--gpus N`` with N > 1: when the process was not started by torch.distributed.run (no WORLD_SIZE in the environment)
bench.py re-launches itself under ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1`` (one process per GPU, RCCL) and passes the child's line through -- the reference's multi-GPU entry is one
command too (run_model.py:392-394).  ``--dp-same-device`` puts every rank on cuda:0 with the gloo transport, which is
how the N > 1 path is exercised on a one-GPU box (tests/test_dp_gpu.py).

Extra objects on the line (all measured outside the timed region):
  b64 / b170    (default command only) the same step at batch 64 per GPU in fp32 AND bf16 (north_star's "at batch 64";
                bf16 = BASELINE configs[2]'s per-GPU workload: `b64.bf16`, also `configs2` when N > 1), each with its
                own `roofline`, `decoder_gemm` (FLOP / summed launch time of the decoder's M = B*30 products only) and
                `host_inclusive`; and at the reference's per-GPU batch of 170 (run_model.py:40).
  roofline      the dominant kernel class of the step by GPU time (the MFMA GEMMs): algorithmic FLOP / summed launch
                time, measured with HIP events on the launch stream inside the library during extra profiled steps
                (fira_prof_*); peak = 157.3 TFLOP/s fp32 MFMA or 2500 TFLOP/s dense bf16 (MI355X_MICROARCH.md).  In
                bf16 the same launches are also priced against HBM (`roofline_hbm`: algorithmic operand bytes / time):
                at K = 256 the products sit at 64 FLOP/B, far left of the 400 FLOP/B ridge.
  spmm / spmm_b64 / spmm_cfg5   the GCN aggregation against HBM (algorithmic bytes of SURVEY.md §8d): inside the step,
                stand-alone on a realistic batch-64 graph batch, and on BASELINE config 5 (128 x 512-node dense graphs).
  host_inclusive  the same step fed by the data path (vectorised collate -> pinned arena -> one async H2D copy per
                batch, two batches ahead on a worker thread): fresh batches every step.
  decode        greedy (batch 64) and beam-3 (batch 20) search on a checkpoint trained for a few hundred steps inside
                the setup (untimed) so that message lengths are realistic; `roofline` = HBM bytes a step must move.
  cpu_baseline  the CPU oracle (a port of the reference's PyTorch path, oracle/fira_oracle.py, checked equal to the
                reference by tests/test_oracle.py) timed on this host on a bounded sample: train steps at batch 4 and 32
                and a few greedy / beam-3 search steps; rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

# The search keeps several batches in flight, each on its own HIP stream (decode.Searcher.greedy_many); HIP multiplexes streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two lanes that land on one queue run one after the other: four lanes
# took 0.38 ms per batch-step on 4 queues and 0.14 on 8 (profiles/r6_probes.md).  Read by the HIP runtime when it initialises,
# so it is set before torch is imported; training is unaffected (same-box triple).  An exported value wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

FP32_MFMA_PEAK_TF = 157.3
BF16_MFMA_PEAK_TF = 2500.0
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--batch", type=int, default=0, help="commits per GPU per step (default 32 for f32 = BASELINE "
                                                         "configs[1], 64 for bf16 = configs[2])")
    ap.add_argument("--decode-batch", type=int, default=64)
    ap.add_argument("--decode-train-steps", type=int, default=300)
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the stand-alone SpMM / host-inclusive legs")
    ap.add_argument("--pool", type=int, default=4, help="distinct resident batches cycled through")
    ap.add_argument("--dp-same-device", action="store_true", help="all ranks on cuda:0 over gloo (one-GPU testing)")
    ap.add_argument("--detail", default="", help="where the full-precision record of every measured object goes "
                    "(default gpurun_out/bench_detail_<dtype>.json); stdout carries ONE short JSON line")
    ap.add_argument("--grad-wire", choices=["auto", "f32", "bf16"], default="auto", help="N > 1, all-reduce path: wire format of "
                    "the two gradient buckets; auto = bf16 with --dtype bf16 (BASELINE configs[2]: 55.6 MB per step), else f32")
    ap.add_argument("--zero1", action="store_true", help="N > 1: reduce-scatter + sharded Adam + all-gather instead of "
                    "all-reduce + replicated Adam")
    return ap.parse_args()


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(a) -> int:
    """One process per GPU under torch.distributed.run; the child (rank 0) prints the JSON line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def prof_report():
    from fira_icse_amd import _lib
    n = 11
    ms, work, byts, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_int64 * n)()
    _lib.lib().fira_prof_report(n, ms, work, byts, cnt)
    # gemm_dec: the decoder's M = B*30 row products (forward + data gradients), split out of the GEMM family by the library
    # gcn: the fused GCN-layer launches (gather + product + LayerNorm / accumulate in one kernel): work = FLOP, bytes = bytes
    # comb: the fused Combination-block launches (round 5: q|k products + gate + output product + LayerNorm in one kernel)
    # dec_region (round 6): WALL time on the caller's stream of the decoder's layer loops (forward + backward; with two
    # commit-lanes the lanes' launches overlap inside it) -- not a kernel class: taken out of the per-class sums below
    names = ["gemm", "spmm", "attention", "rowops", "copy", "head", "adam", "gemm_dec", "gcn", "comb", "dec_region"]
    out = {k: dict(ms=ms[i], work=work[i], bytes=byts[i], count=int(cnt[i])) for i, k in enumerate(names)}
    region = out.pop("dec_region")
    dec = out.pop("gemm_dec")
    dec["region_ms"], dec["region_count"] = region["ms"], region["count"]
    out["gemm"] = {k: out["gemm"][k] + dec[k] for k in ("ms", "work", "bytes", "count")}   # the family = every GEMM launch of the step
    out["gemm"]["decoder"] = dec
    return out


def time_gpu(fn, iters=30, warmup=3):
    """Average seconds per call, HIP events on torch's current stream (the stream the op wrappers launch on)."""
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
    """Compulsory traffic of Z = A_hat H: rowptr + (col, val) + features in + features out."""
    return 4 * (n_rows + 1) + 8 * nnz + 2 * n_rows * d * 4


def spmm_standalone(cfg, store):
    from fira_icse_amd import graphs, ops
    out = {}
    # realistic graphs, batch 64, every node listed (the reference's dense 650-node layout: 41 600 rows)
    hb = store.batch(range(64))
    rp, c, v = (torch.from_numpy(x).cuda() for x in (hb.rowptr, hb.col, hb.val))

    def rotating(n_rows, rp_, c_, v_, graph_rows):
        """Launch over ROTATING (X, Y) buffer sets whose total exceeds the 256 MiB Infinity Cache, so that the features are
        streamed from HBM and not re-read from the last-level cache (MI355X_MICROARCH.md: scale past L3)."""
        per_set = 2 * n_rows * 256 * 4
        n_sets = max(4, -(-300 * (1 << 20) // per_set))
        Xs = [torch.randn(n_rows, 256, device="cuda") for _ in range(n_sets)]
        Ys = [torch.empty(n_rows, 256, device="cuda") for _ in range(n_sets)]
        k = [0]

        def fn():
            i = k[0] % n_sets
            k[0] += 1
            ops.csr_spmm(rp_, c_, v_, Xs[i], graph_rows=graph_rows, variant=1, out=Ys[i])
        t = time_gpu(fn, iters=3 * n_sets, warmup=n_sets)
        return t, n_sets, n_sets * per_set

    X_rows = 64 * cfg.graph_len
    t, n_sets, tot = rotating(X_rows, rp, c, v, cfg.graph_len)
    by = spmm_bytes(X_rows, c.numel())
    out["spmm_b64"] = {"bound": "hbm", "kernel": "spmm_rowwave_kernel", "rows": int(X_rows), "nnz": int(c.numel()),
                       "avg_launch_us": t * 1e6, "bytes_per_launch": by, "achieved": by / t / 1e9, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": by / t / 1e9 / HBM_PEAK_GBS, "buffer_sets": n_sets,
                       "rotated_bytes": tot, "note": "dense 650-node layout (padded nodes carry only a self-loop); "
                       "feature buffers rotated over %d sets = %.0f MiB > the 256 MiB Infinity Cache" % (n_sets, tot / 2**20)}
    # the launch the ENGINE issues: compact rows (computed nodes only), same graphs
    from fira_icse_amd.model import DeviceBatch
    db = DeviceBatch(hb, cfg, "cuda")
    t, n_sets, tot = rotating(db.n_nodes, db.rowptr, db.col, db.val, 0)
    by = spmm_bytes(db.n_nodes, db.nnz)
    out["spmm_b64_compact"] = {"bound": "hbm", "kernel": "spmm_rowwave_kernel", "rows": db.n_nodes, "nnz": db.nnz,
                               "avg_launch_us": t * 1e6, "bytes_per_launch": by, "achieved": by / t / 1e9,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / t / 1e9 / HBM_PEAK_GBS,
                               "buffer_sets": n_sets, "rotated_bytes": tot,
                               "note": "computed-node rows only (what encoder_forward launches), batch 64, stand-alone"}
    # what the CUs actually pull: every listed neighbour row is 1 KiB through the L2 -> CU path even when it hits (a row is
    # gathered ~3.4 times), plus the stored rows.  The chip's L2s deliver ~8.4 TB/s to 256 CUs (13.6 B / clk / CU, measured on
    # config 5: profiles/r6_probes.md) -- the bound this launch sits on, whatever its HBM fraction says.
    for key in ("spmm_b64", "spmm_b64_compact"):
        o = out[key]
        moved = 8 * o["nnz"] + o["nnz"] * 1024 + o["rows"] * 1024
        o["cu_side_bytes"] = moved
        o["cu_side_GBs"] = moved / (o["avg_launch_us"] * 1e-6) / 1e9
        o["cu_side_frac_of_8400GBs"] = o["cu_side_GBs"] / 8400.0
    # BASELINE config 5: 128 graphs x 512 nodes x 4 edge types x 8192 edges
    B, N = 128, 512
    rp, c, v = (torch.from_numpy(x).cuda() for x in graphs.dense_stress_batch(B, N))
    X = torch.randn(B * N, 256, device="cuda")
    by = spmm_bytes(B * N, c.numel())
    cfg5 = {"bound": "hbm", "rows": B * N, "nnz": int(c.numel()), "bytes_per_launch": by, "peak": HBM_PEAK_GBS,
            "unit": "GB/s"}
    # three rotating (X, Y) sets = 402 MB > the Infinity Cache; variants 3 / 4 = block-dense MFMA (spmm_dense.hip)
    Xs = [X] + [torch.randn(B * N, 256, device="cuda") for _ in range(2)]
    Ys = [torch.empty(B * N, 256, device="cuda") for _ in range(3)]
    for variant, name, dt in ((1, "spmm_rowwave_kernel", "f32"), (2, "spmm_lds_kernel", "f32"),
                              (3, "spmm_dense_f32_kernel", "f32"), (4, "spmm_dense_bf16_kernel", "bf16")):
        k = [0]

        def fn():
            i = k[0] % 3
            k[0] += 1
            ops.csr_spmm(rp, c, v, Xs[i], graph_rows=N, variant=variant, out=Ys[i])
        t = time_gpu(fn, iters=12, warmup=3)
        cfg5["v%d" % variant] = {"kernel": name, "dtype": dt, "avg_launch_us": t * 1e6, "achieved": by / t / 1e9,
                                 "frac": by / t / 1e9 / HBM_PEAK_GBS}
        if variant <= 2:
            cfg5["v%d" % variant]["gathered_GBs"] = (8 * c.numel() + c.numel() * 1024 + B * N * 1024) / t / 1e9
        else:
            flop = 2.0 * B * N * N * 256
            cfg5["v%d" % variant]["mfma_TFLOPs"] = flop / t / 1e12
    cfg5["best_f32"] = min(("v1", "v2", "v3"), key=lambda q: cfg5[q]["avg_launch_us"])
    cfg5["best_bf16"] = min(("v1", "v2", "v3", "v4"), key=lambda q: cfg5[q]["avg_launch_us"])
    cfg5["frac"] = cfg5[cfg5["best_f32"]]["frac"]
    cfg5["frac_bf16"] = cfg5[cfg5["best_bf16"]]["frac"]
    cfg5["note"] = ("22 % dense adjacency: the CSR gathers are bound by the L2 / LDS gather rate, the block-dense kernels "
                    "by the MFMA pipe (fp32: 17.2 GFLOP = 110 us at peak) or HBM (bf16 operands, fp32 accumulate)")
    out["spmm_cfg5"] = cfg5
    # SURVEY.md 8(d) row 5, second unit: one full GCN layer (folded form) forward + backward on the same graphs.
    # fused = the launches the engine issues per layer: gcn_fused fwd | LayerNorm backward + gcn_fused bwd + the weight
    # gradient dW21 = V^T X; unfused = aggregation kernel of the density crossover + product + row kernel (forward only).
    n = B * N
    W21 = torch.randn(256, 256, device="cuda") * 0.06
    W21t = W21.t().contiguous()
    b2, c21 = torch.randn(256, device="cuda") * 0.1, torch.randn(256, device="cuda") * 0.1
    gamma, beta = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
    dW = torch.zeros(256, 256, device="cuda")
    layer_bytes_fwd = 4 * (n + 1) + 8 * c.numel() + 3 * n * 1024          # (col, val) + rows in, pre-norm + normalised rows out
    layer_bytes_bwd = 4 * (n + 1) + 8 * c.numel() + 4 * n * 1024 + 4 * n * 1024   # LN bwd: dy, sum in, ds, dx out; fused: dY in, V out, acc in + out
    g5 = {"rows": n, "nnz": int(c.numel()), "flop_fwd": 2.0 * n * 256 * 256, "bytes_fwd": layer_bytes_fwd, "bytes_bwd": layer_bytes_bwd}
    for dt, name in ((0, "f32"), (1, "bf16")):
        k = [0]

        def fwd():
            i = k[0] % 3
            k[0] += 1
            return ops.gcn_layer_fwd(rp, c, v, Xs[i], W21t, b2, c21, gamma, beta, dropout=0.2, seed=1, site=3, dtype=dt,
                                     want_rowsum=False)
        summ, y, stats, _ = fwd()
        t_f = time_gpu(fwd, iters=9, warmup=3)

        def bwd():
            i = k[0] % 3
            k[0] += 1
            ds, dxd, _, _ = ops.add_layernorm_bwd(Ys[i], summ, stats, gamma, dropout=0.2, seed=1, site=3, want_dx_drop=True)
            V = ops.gcn_layer_bwd(rp, c, v, dxd, W21, ds, dtype=dt)
            ops.gemm_wgrad_panel(V, Xs[i], dW, dtype=dt)       # the engine's weight-gradient kernel
        t_b = time_gpu(bwd, iters=9, warmup=3)

        def unfused():
            i = k[0] % 3
            k[0] += 1
            Z = ops.csr_spmm(rp, c, v, Xs[i], graph_rows=N, variant=0, out=Ys[i], dtype=dt, auto=True)
            lin = ops.gemm(Z, W21, bias=b2)
            ops.add_layernorm_fwd(lin, Xs[i], gamma, beta, dropout=0.2, seed=1, site=3)
        def unfused_bwd():                            # the engine's backward on a dense batch (gcn_fused_on() false)
            i = k[0] % 3
            k[0] += 1
            ds, dxd, _, _ = ops.add_layernorm_bwd(Ys[i], summ, stats, gamma, dropout=0.2, seed=1, site=3, want_dx_drop=True)
            V = ops.csr_spmm(rp, c, v, dxd, graph_rows=N, variant=0, dtype=dt, auto=True)
            ops.gemm(V, W21, transB=False, out=ds, accumulate=True, dtype="bf16" if dt else "f32")
            ops.gemm_wgrad_panel(V, Xs[i], dW, dtype=dt)       # the engine's weight-gradient kernel
        try:
            t_u = time_gpu(unfused, iters=9, warmup=3)
            t_ub = time_gpu(unfused_bwd, iters=9, warmup=3)
        except Exception as ex:                       # noqa: BLE001 -- the comparison leg must not take the object down
            t_u = t_ub = float("nan")
            g5["unfused_error_%s" % name] = str(ex)[:120]
        g5[name] = {"fwd_us": t_f * 1e6, "bwd_us": t_b * 1e6, "unfused_fwd_us": t_u * 1e6, "unfused_bwd_us": t_ub * 1e6,
                    "routed_frac_hbm_fwd": layer_bytes_fwd / t_u / 1e9 / HBM_PEAK_GBS,
                    "routed_frac_hbm_bwd": layer_bytes_bwd / t_ub / 1e9 / HBM_PEAK_GBS,
                    "frac_hbm_fwd": layer_bytes_fwd / t_f / 1e9 / HBM_PEAK_GBS, "frac_hbm_bwd": layer_bytes_bwd / t_b / 1e9 / HBM_PEAK_GBS,
                    "frac_mfma_fwd": 2.0 * n * 256 * 256 / t_f / 1e12 / (FP32_MFMA_PEAK_TF if dt == 0 else BF16_MFMA_PEAK_TF)}
    g5["note"] = ("one GCN layer on config 5 (116 entries per row: the fused kernel's gather is its tail path throughout); "
                  "bwd = LayerNorm backward + fused V = A dY, dX += V W21 + the weight gradient V^T X; unfused_* = the launches "
                  "the ENGINE issues on a batch this dense (aggregation kernel of the density crossover + product + row kernel), "
                  "routed_frac_* = the layer's algorithmic bytes over those times")
    out["gcn_cfg5"] = g5
    return out


def host_inclusive(cfg, store, trainer, B, steps):
    """Steps fed by the real data path: collate + pinned arena + one H2D copy per batch on a worker thread."""
    from fira_icse_amd.model import DeviceBatch
    from fira_icse_amd.prefetch import prefetch
    n = len(store)
    order = [[(i * B + k) % n for k in range(B)] for i in range(steps + 4)]
    dev = trainer.model.device_
    it = prefetch(order, lambda idx: DeviceBatch(store.batch(idx), cfg, dev), depth=2, device=dev)
    for _ in range(4):                                  # untimed: also page-locks the worker's three staging slots
        trainer.step(next(it))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for db in it:
        trainer.step(db)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for idx in order[:8]:
        store.batch(idx)
    collate_ms = (time.perf_counter() - t1) / 8 * 1e3
    return {"commits_per_s": steps * B / dt, "ms_per_step": dt / steps * 1e3, "collate_ms_per_batch": collate_ms,
            "note": "fresh batch every step: vectorised CSR collate -> pinned arena -> one async H2D copy, prepared two "
                    "batches ahead on a worker thread"}


def cpu_baseline(cfg, store, seconds_budget=30.0):
    """Reference-equivalent CPU path (oracle port): train steps at batch 4 and 32, a few search steps."""
    from oracle import fira_oracle as O
    from fira_icse_amd.model import reference_init_state_dict
    # more threads than ~32 only add contention on these small-matrix ops (measured: 256 threads are >100x slower)
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    sd = reference_init_state_dict(cfg)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    t_all = time.time()

    def train_leg(B, max_steps, budget):
        P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        opt = torch.optim.Adam(list(P.values()), cfg.lr)
        hb = store.batch(range(B))
        args = (cfg, t(hb.sou), t(hb.tar), t(hb.mark), t(hb.ast_change), t(hb.dense_edge(cfg.graph_len)),
                t(hb.tar_label), t(hb.sub_token))
        times, t_leg = [], time.time()
        for i in range(max_steps):
            t0 = time.time()
            ls, nt = O.forward(P, *args, "train")
            opt.zero_grad(set_to_none=True)
            (ls / nt).backward()
            opt.step()
            times.append(time.time() - t0)
            if time.time() - t_leg > budget and len(times) >= 2:
                break
        med = float(np.median(times[1:])) if len(times) > 1 else times[0]
        return B / med, len(times)

    v4, n4 = train_leg(4, 12, 0.2 * seconds_budget)
    v32, n32 = train_leg(32, 10, 0.5 * seconds_budget)           # ~1.2 s per step on 32 threads: 10 steps ~ 12-15 s

    def search_leg(B, beam, steps):
        hb = store.batch(range(B))
        t0 = time.time()
        O.beam_decode(sd, cfg, t(hb.sou), t(hb.mark), t(hb.ast_change), t(hb.dense_edge(cfg.graph_len)),
                      t(hb.sub_token), beam, max_steps=steps)
        return B * steps / (time.time() - t0)

    g20 = search_leg(20, 1, 4)
    b20 = search_leg(20, 3, 3)
    return {"value": v32, "unit": "commits/s", "cores": threads, "kind": "port",
            "sample": "train steps (fwd+bwd+Adam, dropout off, fp32, dense f64->f32 adjacency as the reference feeds "
                      "it): %d at batch 32 (value = batch / median step time), %d at batch 4; search: 4 greedy / 3 beam-3 steps at batch 20 with "
                      "the encoder pass (full recompute per step as the reference does); %.0f s in total" %
                      (n32, n4, time.time() - t_all),
            "train_batch4_commits_per_s": v4, "train_batch32_commits_per_s": v32,
            "greedy_batch20_step_tokens_per_s": g20, "beam3_batch20_step_tokens_per_s": b20,
            "port_checked_equal_to_reference": "tests/test_oracle.py (fixtures generated by the reference + live import)"}


def decode_leg(cfg, store, model, trainer, batches, a, world, barrier):
    """Greedy (batch 64) and beam-3 (batch 20) search, on weights trained for a few hundred steps (untimed) so that the
    hypotheses have realistic lengths (random-init weights copy <eos> after 1-2 tokens)."""
    from fira_icse_amd.model import DeviceBatch
    from fira_icse_amd.decode import Searcher
    for i in range(a.decode_train_steps):
        trainer.step(batches[i % len(batches)])
    torch.cuda.synchronize()
    model.eval()
    search = Searcher(model)
    dbd = DeviceBatch(store.batch(range(a.decode_batch)), cfg, model.device_)
    for _ in range(2):
        out, length, p = search.greedy(dbd)
    barrier()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        out, length, p = search.greedy(dbd)
    barrier()
    ddt = (time.perf_counter() - t0) / reps
    toks = int((length - 1).sum().item())          # emitted tokens up to and including <eos>, cap 29
    steps_run = int(length.max().item()) - 1
    # HBM bytes one KV-cached step must move: every decoder / head weight once + the cross-attention K|V of the batch's
    # memory rows + the copy head's source projection + the logits row per hypothesis (write + read)
    lay = model.layout
    w_bytes = 4 * sum(int(np.prod(shp)) for k, (off, shp) in lay.entries.items()
                      if k.startswith(("decoder.", "out_fc", "copy_net")) and "embedding" not in k)
    Bd, Sm = a.decode_batch, cfg.mem_len
    # the cross K|V of the VALID memory rows only (the step's attention kernel streams nothing else: the engine stores the
    # computed memory rows compactly), + LinearSource's rows of the valid copy slots
    kv_bytes = 4 * dbd.n_mem * 256 * (2 * cfg.num_layers + 1)
    kv_bytes_dense = 4 * Bd * Sm * 256 * (2 * cfg.num_layers + 1)
    logit_bytes = 2 * 4 * Bd * cfg.out_len
    step_bytes = w_bytes + kv_bytes + logit_bytes
    step_s = ddt / max(steps_run, 1)
    decode = {"dtype": "f32 (the search always runs the reference's fp32 arithmetic, whatever --dtype says: engine.hip fira_decode_*)",
              "tokens_per_s": toks * world / ddt, "commits_per_s": a.decode_batch * world / ddt,
              "step_tokens_per_s": a.decode_batch * steps_run * world / ddt, "steps_run": steps_run,
              "mean_tokens_per_commit": toks / a.decode_batch, "batch": a.decode_batch, "beam": 1,
              "ms_per_batch": ddt * 1e3, "ms_per_step": step_s * 1e3, "tokens_per_batch": toks,
              "trained_steps": a.decode_train_steps,
              "roofline": {"bound": "hbm", "bytes_per_step": step_bytes, "achieved": step_bytes / step_s / 1e9,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                           "memory_rows": int(dbd.n_mem), "memory_rows_dense": Bd * Sm,
                           "bytes_per_step_if_dense_kv": w_bytes + kv_bytes_dense + logit_bytes,
                           "note": "encoder pass included in the step time; the loop is launch-bound (one small "
                                   "kernel per layer op), not bandwidth-bound"},
              "note": "tokens = emitted tokens up to and including <eos> (SURVEY 8d); step_tokens = batch x decoder "
                      "steps executed, the unit of BASELINE.md's CPU figure.  Weights: %d training steps on the "
                      "synthetic commits (untimed), fp32 search" % a.decode_train_steps}
    # several batches of the same size in flight, each on its own stream (Searcher.greedy_many: what run_model.py test does
    # with the test set's consecutive batches); the batch size per search call stays BASELINE configs[3]'s 64
    try:
        n_store = len(store)
        n_fl = int(os.environ.get("FIRA_DECODE_IN_FLIGHT", "4"))
        group = [dbd] + [DeviceBatch(store.batch([(j * a.decode_batch + k) % n_store for k in range(a.decode_batch)]), cfg,
                                     model.device_) for j in range(1, n_fl)]
        for _ in range(2):
            res = search.greedy_many(group, in_flight=n_fl)
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = search.greedy_many(group, in_flight=n_fl)
        barrier()
        d2 = (time.perf_counter() - t0) / reps
        toks2 = sum(int((r[1] - 1).sum().item()) for r in res)
        steps2 = sum(int(r[1].max().item()) - 1 for r in res)
        same = bool(torch.equal(res[0][0], out) and torch.equal(res[0][1], length))
        decode["in_flight"] = {"tokens_per_s": toks2 * world / d2, "commits_per_s": n_fl * a.decode_batch * world / d2,
                               "step_tokens_per_s": a.decode_batch * steps2 * world / d2,
                               "ms_per_group": d2 * 1e3, "ms_per_batch_step": d2 * 1e3 / max(steps2, 1),
                               "speedup_over_one_at_a_time": (a.decode_batch * steps2 / d2) / (a.decode_batch * steps_run / ddt),
                               "batch": a.decode_batch, "batches_in_flight": n_fl, "ids_equal_to_single": same,
                               "note": "%d independent batches of %d, each on its own stream / workspace / captured graphs "
                                       "(one decode step is 41 dependent launches of 16-64 workgroups: one chain leaves most "
                                       "of the chip idle; GPU_MAX_HW_QUEUES=%s); per-batch arithmetic and ids unchanged" % (n_fl, a.decode_batch, os.environ.get("GPU_MAX_HW_QUEUES", "default"))}
    except Exception as e:
        decode["in_flight"] = {"error": repr(e)}
    # the same greedy search streaming a bf16 copy of the cross K|V (FIRA_DECODE_KV_BF16; not the default: the ids are no
    # longer bit-identical to the fp32 search) -- reported beside the fp32 figure, with the token agreement
    try:
        s16 = Searcher(model, kv_bf16=True)
        for _ in range(2):
            out16, length16, p16 = s16.greedy(dbd)
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            out16, length16, p16 = s16.greedy(dbd)
        barrier()
        d16 = (time.perf_counter() - t0) / reps
        steps16 = int(length16.max().item()) - 1
        toks16 = int((length16 - 1).sum().item())
        valid = torch.arange(out.shape[1], device=out.device)[None, :] < torch.minimum(length, length16)[:, None]
        kv16_bytes = w_bytes + kv_bytes // 2 + 4 * dbd.n_mem * 256 // 2 + logit_bytes      # K|V halved, LinearSource rows fp32
        decode["bf16_kv"] = {"ms_per_step": d16 / max(steps16, 1) * 1e3, "tokens_per_s": toks16 * world / d16,
                             "steps_run": steps16, "bytes_per_step": kv16_bytes,
                             "token_agreement_with_fp32": float((out == out16)[valid].float().mean().item()),
                             "same_length": float((length == length16).float().mean().item())}
    except Exception as e:
        decode["bf16_kv"] = {"error": repr(e)}
    # the reference's own test configuration: beam 3, batch 20 (run_model.py:401-415)
    dbb = DeviceBatch(store.batch(range(20)), cfg, model.device_)
    for _ in range(2):
        gen, blen, bp = search.beam(dbb, 3)
    barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        gen, blen, bp = search.beam(dbb, 3)
    barrier()
    bdt = (time.perf_counter() - t0) / reps
    best = search.best(gen, blen, bp)
    decode["beam3"] = {"batch": 20, "ms_per_batch": bdt * 1e3, "commits_per_s": 20 * world / bdt,
                       "tokens_per_s": sum(len(h) - 1 for h in best) * world / bdt,
                       "steps_run": int(blen.max().item()) - 1}
    model.train()
    return decode


def gemm_objects(prof, dtype, prof_steps, traffic, dec_rows=None):
    """`roofline` (MFMA), `roofline_hbm` and the decoder-GEMM sub-object of one profiled leg (fira_prof_* HIP events).
    dec_rows = (computed target rows, B*30 rows) of the profiled batches: the decoder runs on the computed rows only
    (fira_batch.dec_off), so its EXECUTED FLOP are below the reference's dense B*30-row products; both rates are reported
    -- `achieved` / `frac` on the executed FLOP, `*_dense_equiv` on the FLOP of the dense shapes over the
    same time."""
    gemm = prof["gemm"]
    dec = gemm["decoder"]
    g_s = max(gemm["ms"], 1e-9) * 1e-3
    total_ms = sum(v["ms"] for v in prof.values()) or 1.0
    peak_tf = BF16_MFMA_PEAK_TF if dtype == "bf16" else FP32_MFMA_PEAK_TF
    kern = "gemm_bf16_k256_kernel / gemm_bf16_small_kernel / gemm_bf16_kernel (v_mfma_f32_32x32x16_bf16) + the one-plane linear_x3 / dgrad_x3_splitk / wgrad_panel kernels (v_mfma_f32_16x16x32_bf16)" if dtype == "bf16" else \
        "gemm_f32_kernel / gemm_tile32_kernel (v_mfma_f32_32x32x2_f32) + wgrad_panel_kernel (weight gradients), head_logits_x3_kernel (generator projection), linear_x3_kernel / linear_x3_kacc_kernel (cross-attention K|V projection of the memory rows and its data gradient): fp32-accurate three-term bf16 split on v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16; algorithmic fp32 FLOP priced against the fp32 MFMA peak"
    roofline = {"bound": "mfma", "kernel": kern, "achieved": gemm["work"] / g_s / 1e12, "peak": peak_tf,
                "unit": "TFLOP/s", "frac": gemm["work"] / g_s / 1e12 / peak_tf,
                "traffic": traffic.get("gemm", {}).get("hbm_bytes_per_launch"),
                "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_traffic.sh (counters cannot "
                                  "be read inside this process): profiles/traffic.json, batch 32 (f32) / 64 (bf16)",
                "algorithmic_flop_per_launch": gemm["work"] / max(gemm["count"], 1),
                "launches_per_step": gemm["count"] // prof_steps, "avg_launch_us": 1e3 * gemm["ms"] / max(gemm["count"], 1),
                "share_of_kernel_time": gemm["ms"] / total_ms}
    roofline_hbm = {"bound": "hbm", "kernel": kern, "achieved": gemm["bytes"] / g_s / 1e9, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": gemm["bytes"] / g_s / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": gemm["bytes"] / max(gemm["count"], 1),
                    "flop_per_byte": gemm["work"] / max(gemm["bytes"], 1.0),
                    "note": "operands and result counted once in fp32 storage: 4*(M*K + N*K + M*N) per product"}
    d_s = max(dec["ms"], 1e-9) * 1e-3
    decoder_gemm = {"bound": "mfma", "achieved": dec["work"] / d_s / 1e12, "peak": peak_tf, "unit": "TFLOP/s",
                    "frac": dec["work"] / d_s / 1e12 / peak_tf, "launches_per_step": dec["count"] // prof_steps,
                    "avg_launch_us": 1e3 * dec["ms"] / max(dec["count"], 1),
                    "algorithmic_flop_per_launch": dec["work"] / max(dec["count"], 1),
                    "hbm_GBs": dec["bytes"] / d_s / 1e9,
                    "note": "the decoder's M = B*30 row products only (forward + data gradients), FLOP / summed launch "
                            "time: the quantity north_star's >= 40 % MFMA target is set on"}
    if dec.get("region_ms", 0) > 0:
        # FLOP of the decoder's products over the WALL time of the decoder's layer loops on the caller's stream (attention,
        # LayerNorm prologues, waits and -- with two commit-lanes -- the overlap of the lanes included): occupancy of the
        # region, not per-launch latency
        r_s = dec["region_ms"] * 1e-3
        decoder_gemm["region_ms_per_step"] = dec["region_ms"] / prof_steps
        decoder_gemm["achieved_region"] = dec["work"] / r_s / 1e12
        decoder_gemm["frac_region"] = dec["work"] / r_s / 1e12 / peak_tf
    if dec_rows and dec_rows[0] > 0:
        ratio = dec_rows[1] / dec_rows[0]
        decoder_gemm["rows_computed_over_dense"] = dec_rows[0] / dec_rows[1]
        decoder_gemm["achieved_dense_equiv"] = decoder_gemm["achieved"] * ratio
        decoder_gemm["frac_dense_equiv"] = decoder_gemm["frac"] * ratio
        dense_work = gemm["work"] + dec["work"] * (ratio - 1.0)
        roofline["achieved_dense_equiv"] = dense_work / g_s / 1e12
        roofline["frac_dense_equiv"] = dense_work / g_s / 1e12 / peak_tf
    return roofline, roofline_hbm, decoder_gemm


def attention_work(metas, n_layers, d=256):
    """Algorithmic FLOP and bytes of the attention launches of ONE training step over the given batches' commits
    (gnn_transformer.py:137-158; 2 matmuls forward, 5 backward; causal self-attention counted at half of Tq x Tq):
    metas = [(tq[B], tk[B])] = computed target rows and valid memory keys per commit."""
    flop = byts = 0.0
    for tq, tk in metas:
        tq, tk = tq.astype(np.float64), tk.astype(np.float64)
        cross, self_ = float((tq * tk).sum()), float((tq * tq).sum()) / 2
        flop += n_layers * (2 + 5) * 2.0 * d * (cross + self_)
        row = 4.0 * d
        # forward: K, V rows in, Q in, O out; backward: K, V in + dK, dV out, Q, O, dO in, dQ out
        byts += n_layers * row * float(((2 + 4) * tk + (2 + 4) * tq).sum())            # cross
        byts += n_layers * row * float(((3 + 1) * tq + (5 + 3) * tq).sum())            # self (q|k|v rows, o; + gradients)
    return flop, byts


def rocprof_reference(dtype):
    """Per-class kernel time of the SAME command from the committed rocprofv3 kernel trace (profiles/r6_kernel_classes.json,
    written by scripts/rocpd_stats.py from `rocprofv3 --kernel-trace --stats -- python bench.py --dtype <dtype> ...`):
    kernel begin-to-end durations, i.e. without the launch gaps and the cross-stream event overlap that the in-process
    HIP-event sums include.  Counters and traces cannot be collected inside this process."""
    for name in ("r6_kernel_classes.json", "r5_kernel_classes.json"):        # the newest committed trace
        try:
            with open(os.path.join(HERE, "profiles", name)) as f:
                ref = json.load(f).get(dtype)
            if ref:
                return ref
        except Exception:
            pass
    return None


def lib_sha16():
    """First 16 hex digits of the SHA-256 of the libfira_hip.so this process loaded (which build produced the line)."""
    import hashlib
    from fira_icse_amd import _lib
    try:
        with open(_lib.LIB_PATH, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def box_fingerprint():
    """The pool's boxes fall into a slow and a fast class ~5 % apart on the training line; a device-to-device copy of 1 GiB
    (read + write, 20 repetitions) separates them the same way and costs 10 ms: GB/s of that copy, with the host name."""
    try:
        n = 1 << 28
        src = torch.empty(n, dtype=torch.float32, device="cuda")
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            dst.copy_(src)
        b.record()
        torch.cuda.synchronize()
        gbs = 20 * 2.0 * n * 4 / (a.elapsed_time(b) * 1e-3) / 1e9
        del src, dst
        return {"host": socket.gethostname(), "copy_GBs": round(gbs, 1)}
    except Exception as e:      # noqa: BLE001
        return {"host": socket.gethostname(), "error": repr(e)[:80]}


def _sig(x, n=6):
    """Floats at n significant digits (the line is read by a parser with a size limit; the full-precision record is the
    detail file)."""
    if isinstance(x, float):
        return float("%.*g" % (n, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def _pick(obj, keys):
    return {k: obj[k] for k in keys if isinstance(obj, dict) and k in obj and obj[k] is not None}


def write_detail(line, a):
    """Every object this run measured, at full precision, as ONE JSON file beside the (short) stdout line: the driver's
    parser dropped round 4's 23 KB line, so the stdout line now carries the contract's keys + a flat summary and names this
    file.  `--detail PATH` overrides the default gpurun_out/bench_detail_<dtype>[_b<batch>].json."""
    path = a.detail or os.path.join(HERE, "gpurun_out", "bench_detail_%s%s.json" % (a.dtype, "_b%d" % a.batch if a.batch else ""))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(line, f, indent=1)
        return os.path.relpath(path, HERE)
    except Exception:
        return None


def compact_line(line, detail_path):
    """The ONE stdout line: BASELINE.json's metric + `roofline` + `cpu_baseline` as the contract names them, a flat `summary`
    of the other measured legs (each a scalar; the objects they come from are in the detail file), under ~4 KB."""
    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype", "data", "host_enqueue_ms_per_step", "config")}
    r = line["roofline"]
    out["roofline"] = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step",
                                "avg_launch_us", "algorithmic_flop_per_launch", "share_of_kernel_time", "frac_dense_equiv"))
    if isinstance(r.get("rocprof"), dict):
        out["roofline"]["rocprof_frac"] = r["rocprof"].get("frac")
        out["roofline"]["rocprof_commit"] = r["rocprof"].get("commit")
    out["roofline"]["method"] = ("algorithmic FLOP / sum of per-launch HIP-event times on the launch streams (in-process, extra "
                                 "steps); rocprof_frac = same FLOP / kernel durations of the committed rocprofv3 trace; "
                                 "traffic = PMC bytes per launch (profiles/traffic.json)")
    cpu = line.get("cpu_baseline")
    if isinstance(cpu, dict):
        out["cpu_baseline"] = _pick(cpu, ("value", "unit", "cores", "kind", "sample", "error"))
        if "sample" in out["cpu_baseline"]:
            out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"][:160]
    else:
        out["cpu_baseline"] = cpu
    if "roofline_hbm" in line:
        out["roofline_hbm"] = _pick(line["roofline_hbm"], ("bound", "achieved", "peak", "unit", "frac"))
    if "process_group" in line:
        pg = dict(line["process_group"])
        pg.pop("note", None)
        out["process_group"] = pg
    s = {}

    def put(key, obj, *path):
        try:
            for p in path:
                obj = obj[p]
            if obj is not None:
                s[key] = obj
        except Exception:
            pass
    put("decoder_gemm_frac", line, "decoder_gemm", "frac")
    put("decoder_gemm_frac_dense_equiv", line, "decoder_gemm", "frac_dense_equiv")
    put("decoder_region_ms", line, "decoder_gemm", "region_ms_per_step")
    for k in ("avg_launch_us", "frac_mfma", "frac_hbm", "traffic"):
        put("gcn_" + k, line, "gcn", k)
        put("attention_" + k, line, "attention", k)
    put("comb_avg_launch_us", line, "comb", "avg_launch_us")
    put("comb_frac_mfma", line, "comb", "frac_mfma")
    put("spmm_in_step_frac", line, "spmm", "frac")
    put("spmm_b64_frac", line, "spmm_b64", "frac")
    put("spmm_b64_us", line, "spmm_b64", "avg_launch_us")
    put("spmm_b64_compact_frac", line, "spmm_b64_compact", "frac")
    put("spmm_b64_compact_cu_side_GBs", line, "spmm_b64_compact", "cu_side_GBs")
    put("spmm_cfg5_frac_f32", line, "spmm_cfg5", "frac")
    put("spmm_cfg5_frac_bf16", line, "spmm_cfg5", "frac_bf16")
    put("gcn_cfg5_f32_fwd_us", line, "gcn_cfg5", "f32", "fwd_us")
    put("gcn_cfg5_f32_bwd_us", line, "gcn_cfg5", "f32", "bwd_us")
    put("gcn_cfg5_bf16_fwd_us", line, "gcn_cfg5", "bf16", "fwd_us")
    put("gcn_cfg5_bf16_bwd_us", line, "gcn_cfg5", "bf16", "bwd_us")
    put("gcn_cfg5_f32_routed_fwd_us", line, "gcn_cfg5", "f32", "unfused_fwd_us")
    put("gcn_cfg5_f32_routed_bwd_us", line, "gcn_cfg5", "f32", "unfused_bwd_us")
    put("gcn_cfg5_bf16_routed_fwd_us", line, "gcn_cfg5", "bf16", "unfused_fwd_us")
    put("gcn_cfg5_bf16_routed_bwd_us", line, "gcn_cfg5", "bf16", "unfused_bwd_us")
    put("host_inclusive_commits_per_s", line, "host_inclusive", "commits_per_s")
    for leg in ("b64", "b170"):
        for dt in ("f32", "bf16"):
            put("%s_%s_commits_per_s" % (leg, dt), line, leg, dt, "commits_per_s")
            put("%s_%s_ms_per_step" % (leg, dt), line, leg, dt, "ms_per_step")
            put("%s_%s_gemm_frac" % (leg, dt), line, leg, dt, "roofline", "frac")
            put("%s_%s_decoder_gemm_frac" % (leg, dt), line, leg, dt, "decoder_gemm", "frac")
            put("%s_%s_decoder_gemm_frac_dense_equiv" % (leg, dt), line, leg, dt, "decoder_gemm", "frac_dense_equiv")
            put("%s_%s_gcn_frac_hbm" % (leg, dt), line, leg, dt, "gcn", "frac_hbm")
    put("b64_bf16_over_f32", line, "b64", "bf16_over_f32")
    put("b64_bf16_host_inclusive_commits_per_s", line, "b64", "bf16", "host_inclusive", "commits_per_s")
    put("decode_tokens_per_s", line, "decode", "tokens_per_s")
    put("decode_step_tokens_per_s", line, "decode", "step_tokens_per_s")
    put("decode_ms_per_step", line, "decode", "ms_per_step")
    put("decode_batch", line, "decode", "batch")
    put("decode_hbm_frac", line, "decode", "roofline", "frac")
    put("decode_in_flight_tokens_per_s", line, "decode", "in_flight", "tokens_per_s")
    put("decode_in_flight_batches", line, "decode", "in_flight", "batches_in_flight")
    put("decode_beam3_b20_commits_per_s", line, "decode", "beam3", "commits_per_s")
    put("decode_error", line, "decode", "error")
    for k in ("legs_error", "extras_error"):
        put(k, line, k)
    if isinstance(cpu, dict):
        put("cpu_greedy_b20_step_tokens_per_s", cpu, "greedy_batch20_step_tokens_per_s")
        put("cpu_train_b4_commits_per_s", cpu, "train_batch4_commits_per_s")
    out["summary"] = s
    out["kernel_time_ms_per_step"] = line.get("kernel_time_ms_per_step")
    out["detail"] = detail_path
    return _sig(out)


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))

    from fira_icse_amd import _lib, data, synth
    from fira_icse_amd.config import FiraConfig
    from fira_icse_amd.parallel import init_from_env

    if a.dp_same_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = init_from_env("gloo" if a.dp_same_device else None)
    if world != a.gpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    torch.cuda.set_device(local)
    from fira_icse_amd.model import TransModel, DeviceBatch
    from fira_icse_amd.train import Trainer

    cfg = FiraConfig()
    B = a.batch or (64 if a.dtype == "bf16" else 32)
    default_line = a.batch == 0 and a.dtype == "f32"      # the driver's command: carries the batch-64 / 170 objects too
    want_legs = default_line and not a.no_extras
    n_commits = max(a.pool * B, a.decode_batch, 64, (a.pool * 64) if want_legs else 0, (2 * 170) if want_legs else 0)
    store = data.process_raw(cfg, synth.generate_dataset(n_commits, seed=1000 + rank))
    torch.manual_seed(0)
    model = TransModel(cfg, device="cuda:%d" % local)
    model.compute_dtype = a.dtype
    model.set_dropout_stream(0, rank)
    model.train()                                      # dropout on, as the reference trains
    wire = a.grad_wire if a.grad_wire != "auto" else ("bf16" if a.dtype == "bf16" else "f32")
    trainer = Trainer(model, distributed=world > 1, zero1=a.zero1, grad_wire=wire)
    lib = _lib.lib()
    try:        # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process)
        with open(os.path.join(HERE, "profiles", "traffic.json")) as f:
            traffic_all = json.load(f)
    except Exception:
        traffic_all = {}

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def make_batches(Bx, pool):
        n = len(store)
        out = []
        for i in range(pool):
            hb = store.batch([(i * Bx + k) % n for k in range(Bx)])
            db = DeviceBatch(hb, cfg, model.device_)
            # per commit: computed target rows (DeviceBatch's prefix rule) and valid memory keys -- the attention roofline
            used = np.asarray(hb.tar) != 0
            used[:, :-1] |= np.asarray(hb.tar_label)[:, 1:] != 0
            T = used.shape[1]
            tq = np.where(used.any(axis=1), T - np.argmax(used[:, ::-1], axis=1), 1)
            tk = (np.asarray(hb.sou) != 0).sum(1) + (np.asarray(hb.sub_token) != 0).sum(1)
            db.attn_meta = (tq if model.compact_dec else np.full_like(tq, T), tk)
            out.append(db)
        return out

    def timed_leg(batches, steps, warmup):
        """W untimed + K timed steps, barrier + synchronize on both sides, max over ranks."""
        for i in range(warmup):
            trainer.step(batches[i % len(batches)])
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            trainer.step(batches[i % len(batches)])
        t_enq = time.perf_counter() - t0              # host time to enqueue the steps (the stream is still draining)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=model.device_)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, t_enq

    def profiled(batches, prof_steps=3):
        """kernel-class timing with HIP events on the launch stream (extra steps, outside any timed region)"""
        lib.fira_prof_enable(1)
        for i in range(prof_steps):
            trainer.step(batches[i % len(batches)])
        torch.cuda.synchronize()
        prof = prof_report()
        lib.fira_prof_enable(0)
        return prof

    def dec_rows_of(bs, n):
        """(computed target rows, dense B*30 rows) of the first n profiled batches"""
        used = [bs[i % len(bs)] for i in range(n)]
        dense = sum(b.B * cfg.tar_len for b in used)
        return (sum(b.n_dec_rows for b in used) if model.compact_dec else dense, dense)

    def side_leg(dtype, Bx, steps, warmup, pool, with_host=False):
        """One more training configuration measured like the headline (same trainer, other dtype / batch)."""
        model.compute_dtype = dtype
        bs = make_batches(Bx, pool)
        dt, t_enq = timed_leg(bs, steps, warmup)
        prof = profiled(bs)
        tr = traffic_all.get(dtype) or {}
        roof, roof_hbm, dec = gemm_objects(prof, dtype, 3, tr, dec_rows_of(bs, 3))
        obj = {"commits_per_s": steps * Bx * world / dt, "ms_per_step": dt / steps * 1e3, "batch_per_gpu": Bx,
               "dtype": dtype, "steps": steps, "host_enqueue_ms_per_step": t_enq / steps * 1e3, "roofline": roof,
               "decoder_gemm": dec, "gcn": gcn_object(prof, dtype, 3), "comb": comb_object(prof, 3),
               "attention": attention_object(prof, bs, 3),
               "kernel_time_ms_per_step": {k: v["ms"] / 3 for k, v in prof.items()}}
        if dtype == "bf16":
            obj["roofline_hbm"] = roof_hbm
        if with_host and world == 1:
            obj["host_inclusive"] = host_inclusive(cfg, store, trainer, Bx, max(10, steps))
        del bs
        return obj

    def gcn_object(prof, dtype, n_steps):
        """The fused GCN-layer launches (one per layer and direction): MFMA-bound in fp32, gather / HBM-bound in bf16."""
        g = prof.get("gcn")
        if not g or g["count"] == 0:
            return None
        t_s = g["ms"] * 1e-3
        peak_tf = BF16_MFMA_PEAK_TF if dtype == "bf16" else FP32_MFMA_PEAK_TF
        return {"kernel": "gcn_fused_kernel (gather A_hat X -> bf16 planes in LDS -> v_mfma_f32_16x16x32_bf16: six terms per k step "
                          "in fp32 mode = fp32-accurate, one in bf16 mode -> LayerNorm / accumulate rows)",
                "bound": "hbm", "launches_per_step": g["count"] // n_steps,
                "avg_launch_us": 1e3 * g["ms"] / g["count"], "flop_per_launch": g["work"] / g["count"],
                "bytes_per_launch": g["bytes"] / g["count"], "achieved_TFLOPs": g["work"] / t_s / 1e12,
                "frac_mfma": g["work"] / t_s / 1e12 / peak_tf, "achieved_GBs": g["bytes"] / t_s / 1e9,
                "frac_hbm": g["bytes"] / t_s / 1e9 / HBM_PEAK_GBS,
                "note": "bytes = rowptr + gathered rows in + rows out (3 row streams forward, 4 backward); bench adds no "
                        "(col, val) bytes here; FLOP = the algorithmic [rows,256]x[256,256] product (frac_mfma prices it against "
                        "the mode's MFMA peak: fp32 157.3 TF although round 6 runs it as bf16 terms; the launch is bound by the "
                        "gather's dependent round trips and the row phase, not by either roof)"}

    def comb_object(prof, n_steps):
        """The fused Combination-block launches (comb_fused.hip): three [n_code,256]x[256,256] products per launch, fp32 MFMA."""
        g = prof.get("comb")
        if not g or g["count"] == 0:
            return None
        t_s = g["ms"] * 1e-3
        return {"kernel": "comb_fused_fwd_kernel / comb_fused_bwd_kernel (q|k products -> gate in registers -> output product -> LayerNorm "
                          "rows; round 6: the products on bf16 planes, six terms per k step in fp32 mode)",
                "bound": "mfma", "launches_per_step": g["count"] // n_steps, "avg_launch_us": 1e3 * g["ms"] / g["count"],
                "flop_per_launch": g["work"] / g["count"], "bytes_per_launch": g["bytes"] / g["count"],
                "achieved_TFLOPs": g["work"] / t_s / 1e12, "frac_mfma": g["work"] / t_s / 1e12 / FP32_MFMA_PEAK_TF,
                "achieved_GBs": g["bytes"] / t_s / 1e9, "frac_hbm": g["bytes"] / t_s / 1e9 / HBM_PEAK_GBS}

    def attention_object(prof, bs, n_steps):
        at = prof["attention"]
        if at["count"] == 0:
            return None
        flop, byts = attention_work([bs[i % len(bs)].attn_meta for i in range(n_steps)], cfg.num_layers)
        t_s = at["ms"] * 1e-3
        return {"kernel": "attention_fwd_kernel / attention_bwd_kernel (fp32 MFMA chains; ragged query and key rows)",
                "bound": "latency", "launches_per_step": at["count"] // n_steps, "avg_launch_us": 1e3 * at["ms"] / at["count"],
                "flop_per_step": flop / n_steps, "bytes_per_step": byts / n_steps,
                "achieved_TFLOPs": flop / t_s / 1e12, "frac_mfma": flop / t_s / 1e12 / FP32_MFMA_PEAK_TF,
                "achieved_GBs": byts / t_s / 1e9, "frac_hbm": byts / t_s / 1e9 / HBM_PEAK_GBS,
                "note": "algorithmic FLOP (2 matmuls forward, 5 backward over computed target rows x valid keys, causal half) "
                        "and bytes (valid K|V rows, Q / O / gradient rows once) over the summed launch time"}

    batches = make_batches(B, a.pool)
    nnz_mean = float(np.mean([b.nnz for b in batches]))
    dt, t_enq = timed_leg(batches, a.steps, a.warmup)
    loss = trainer.last_loss()
    commits_per_s = a.steps * B * world / dt

    prof_steps = 3
    prof = profiled(batches, prof_steps)
    total_ms = sum(v["ms"] for v in prof.values()) or 1.0
    traffic = traffic_all.get(a.dtype) or (traffic_all if a.dtype == "f32" and "gemm" in traffic_all else {})
    roofline, roofline_hbm, decoder_gemm = gemm_objects(prof, a.dtype, prof_steps, traffic, dec_rows_of(batches, prof_steps))
    ref = rocprof_reference(a.dtype)
    if ref and ref.get("batch") == B and ref.get("gemm_us_per_step"):
        # the same FLOP over the rocprofv3 kernel durations of the committed trace of this command: the number a reader of
        # profiles/r5_kernel_stats_<dtype>.md recomputes (kernel begin-to-end, no launch gaps)
        flop_step = prof["gemm"]["work"] / prof_steps
        tf = flop_step / (ref["gemm_us_per_step"] * 1e-6) / 1e12
        roofline["rocprof"] = {"gemm_us_per_step": ref["gemm_us_per_step"], "gemm_launches_per_step": ref.get("gemm_launches_per_step"),
                               "achieved": tf, "frac": tf / (BF16_MFMA_PEAK_TF if a.dtype == "bf16" else FP32_MFMA_PEAK_TF),
                               "source": ref.get("source"), "commit": ref.get("commit"), "box_commits_per_s": ref.get("commits_per_s")}
    roofline["method"] = ("frac = algorithmic FLOP / SUM of per-launch HIP-event times inside this process (includes launch gaps "
                          "and the overlap of the three streams: a lower bound); `rocprof` = the same FLOP over the kernel "
                          "durations of the committed rocprofv3 trace of this command")
    spmm = prof["spmm"]
    spmm_obj = None
    if spmm["count"] > 0:                           # (the separate aggregation kernel: FIRA_GCN_FUSED=0)
        spmm_bytes_step = spmm["work"] + 8.0 * nnz_mean * spmm["count"]         # + (col,val) of the batch's nnz
        spmm_obj = {"bound": "hbm", "kernel": "spmm_rowwave_kernel", "achieved": spmm_bytes_step / (spmm["ms"] * 1e-3) / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": spmm_bytes_step / (spmm["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "traffic": traffic.get("spmm", {}).get("hbm_bytes_per_launch"),
                    "avg_launch_us": 1e3 * spmm["ms"] / max(spmm["count"], 1),
                    "bytes_per_launch": spmm_bytes_step / max(spmm["count"], 1), "share_of_kernel_time": spmm["ms"] / total_ms}
    gcn_obj = gcn_object(prof, a.dtype, prof_steps)
    if gcn_obj:
        gcn_obj["traffic"] = traffic.get("gcn", {}).get("hbm_bytes_per_launch")
        gcn_obj["share_of_kernel_time"] = prof["gcn"]["ms"] / total_ms
    attn_obj = attention_object(prof, batches, prof_steps)
    if attn_obj:
        attn_obj["traffic"] = traffic.get("attention", {}).get("hbm_bytes_per_launch")
        attn_obj["share_of_kernel_time"] = prof["attention"]["ms"] / total_ms

    extras = {}
    single = rank == 0 and world == 1
    if want_legs:
        # BASELINE configs[2]'s per-GPU workload and north_star's "at batch 64" targets: fp32 AND bf16 at batch 64 per
        # GPU (same trainer; with N > 1 ranks these are data-parallel steps over RCCL, i.e. configs[2] itself), and the
        # reference's real operating point, 170 commits per GPU (run_model.py:40).  All outside the headline's timed region.
        try:
            b64 = {"f32": side_leg("f32", 64, a.steps, 3, a.pool, with_host=True),
                   "bf16": side_leg("bf16", 64, a.steps, 3, a.pool, with_host=True)}
            b64["decoder_gemm"] = {"f32": b64["f32"]["decoder_gemm"], "bf16": b64["bf16"]["decoder_gemm"],
                                   "frac": b64["bf16"]["decoder_gemm"]["frac"],
                                   "frac_f32": b64["f32"]["decoder_gemm"]["frac"],
                                   "note": "frac = bf16 (configs[2]'s arithmetic) against 2.5 PFLOP/s; frac_f32 against 157.3 TFLOP/s"}
            b64["bf16_over_f32"] = b64["bf16"]["commits_per_s"] / b64["f32"]["commits_per_s"]
            extras["b64"] = b64
            if world > 1:
                extras["configs2"] = dict(b64["bf16"], workload="BASELINE configs[2]: data-parallel over %d GPUs, batch "
                                          "64/GPU, bf16" % world, global_batch=64 * world)
            extras["b170"] = {"f32": side_leg("f32", 170, max(5, a.steps // 2), 2, 2),
                              "bf16": side_leg("bf16", 170, max(5, a.steps // 2), 2, 2),
                              "note": "the reference's per-GPU batch (run_model.py:40)"}
        except Exception as e:
            extras["legs_error"] = repr(e)
        model.compute_dtype = a.dtype
    if single and not a.no_extras:
        try:
            extras.update(spmm_standalone(cfg, store))
            extras["host_inclusive"] = host_inclusive(cfg, store, trainer, B, max(10, a.steps))
        except Exception as e:                          # an auxiliary leg must never take the bench line down
            extras["extras_error"] = repr(e)

    decode = None
    if not a.no_decode:
        try:
            decode = decode_leg(cfg, store, model, trainer, batches, a, world, barrier)
        except Exception as e:
            decode = {"error": repr(e)}

    cpu = None
    if single and not a.no_cpu_baseline:
        try:
            cpu = cpu_baseline(cfg, store)
        except Exception as e:
            cpu = {"error": repr(e)}

    comm = None
    if world > 1 and trainer.reducer is not None and not a.zero1:
        # per-bucket collective timing (extra steps, outside the timed region): HIP events around each all-reduce on the
        # stream that carries it, and around the caller-stream waits (= the communication that was NOT hidden)
        trainer.reducer.timing = True
        for i in range(6):
            trainer.step(batches[i % len(batches)])
        comm = trainer.reducer.timing_summary()
        trainer.reducer.timing = False
        barrier()
    if rank == 0:
        pg = None
        if world > 1:
            pg = {"rccl_world_size": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(),
                  "mode": "zero1 (reduce-scatter + sharded Adam + all-gather)" if a.zero1 else "all-reduce in 2 readiness buckets",
                  "gradient_bytes_per_rank": 4 * int(model.layout.live),
                  "fused_step": bool(getattr(trainer, "fused_dp", False)) and not a.zero1}
            if trainer.reducer is not None and not a.zero1:
                pg["wire"] = trainer.reducer.wire
                pg["bytes_per_step"] = trainer.reducer.bytes_per_step()
            if comm:
                pg.update(comm)
                pg["note"] = ("allreduce_ms_early = head+decoder bucket on the side stream from the mid-backward event "
                              "(overlaps the encoder backward); allreduce_ms_late = encoder bucket, launch to ready on the "
                              "caller's stream (overlaps Adam of the early slice); exposed_comm_ms = time the caller's "
                              "stream stood waiting for the two buckets")
        wl = ("BASELINE configs[1]: FIRA training step (fwd+bwd+Adam, dropout 0.1/0.2), batch %d commits/GPU, fp32"
              if a.dtype == "f32" else
              "BASELINE configs[2] per-GPU workload: FIRA training step (fwd+bwd+Adam, dropout 0.1/0.2), batch %d "
              "commits/GPU, bf16 MFMA products with fp32 accumulation / storage / LayerNorm / loss / Adam")
        line = {
            "metric": "training commits/sec (FIRA default config)", "value": commits_per_s, "unit": "commits/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "host_enqueue_ms_per_step": t_enq / a.steps * 1e3,
            "config": {"workload": (wl % B) + ", 650-node graphs (mean nnz %.0f/graph), vocab 24650" % (nnz_mean / B),
                       "global_batch": B * world, "parallelism": "dp%d%s" % (world, "+zero1" if (a.zero1 and world > 1) else ""), "loss": loss,
                       "lib_sha16": lib_sha16(), "box": box_fingerprint() if not a.no_extras else {"host": socket.gethostname()}},
            "roofline": roofline, "decoder_gemm": decoder_gemm, "gcn": gcn_obj, "comb": comb_object(prof, prof_steps),
            "attention": attn_obj, "spmm": spmm_obj,
            "decode": decode, "cpu_baseline": cpu,
            "kernel_time_ms_per_step": {k: v["ms"] / prof_steps for k, v in prof.items()},
        }
        if a.dtype == "bf16":
            line["roofline_hbm"] = roofline_hbm
        if pg:
            line["process_group"] = pg
        line.update(extras)
        # the reference's own per-GPU batch (170, run_model.py:40) and BASELINE configs[2]'s per-GPU workload (batch 64, bf16)
        # as first-class fields of `config`
        for key, path in (("b170_f32_commits_per_s", ("b170", "f32", "commits_per_s")),
                          ("b170_bf16_commits_per_s", ("b170", "bf16", "commits_per_s")),
                          ("b64_f32_commits_per_s", ("b64", "f32", "commits_per_s")),
                          ("b64_bf16_commits_per_s", ("b64", "bf16", "commits_per_s"))):
            try:
                obj = extras
                for p in path:
                    obj = obj[p]
                line["config"][key] = obj
            except Exception:
                pass
        detail_path = write_detail(line, a)
        print(json.dumps(compact_line(line, detail_path)), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
