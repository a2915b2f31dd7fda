#!/usr/bin/env python
"""Benchmark of the FIRA hot path on MI355X: training commits/s (+ greedy-decode tokens/s), BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run)

A "step" is one optimisation step of run_model.py:101-112 (forward + backward + [RCCL all-reduce] + Adam) over one
batch of synthetic commits in the reference's raw schema (there is no network for the real DataSet; seed-0 generator,
SURVEY.md §8d), with the reference's dropout (0.1 / 0.2) ON and fp32 arithmetic.  The N=1 workload is BASELINE
configs[1] ("1xMI355X, batch=32, fp32"); every rank keeps batch 32 as N grows (weak scaling).  Inputs are resident in
HBM before the timed region (pre-collated device batches, cycled).  One JSON line is printed by rank 0.

Extra objects on the line:
  roofline      dominant kernel of the step by GPU time (the fp32 MFMA GEMM): algorithmic FLOP / summed launch time,
                both measured with HIP events inside the library on the launch stream during extra profiled steps
                (fira_prof_*); peak = 157.3 TFLOP/s fp32 MFMA (MI355X_MICROARCH.md).  `spmm` carries the same for the
                GCN aggregation against HBM (algorithmic bytes).
  cpu_baseline  the CPU oracle (a port of the reference's PyTorch path, oracle/fira_oracle.py) timed on this host on a
                bounded sample (a few batch-4 steps), rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from fira_icse_amd import _lib, data, synth                      # noqa: E402
from fira_icse_amd.config import FiraConfig                     # noqa: E402
from fira_icse_amd.parallel import init_from_env                # noqa: E402

FP32_MFMA_PEAK_TF = 157.3
HBM_PEAK_GBS = 8000.0


def prof_report():
    n = 7
    ms, work, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_int64 * n)()
    _lib.lib().fira_prof_report(n, ms, work, cnt)
    names = ["gemm", "spmm", "attention", "rowops", "copy", "head", "adam"]
    return {k: dict(ms=ms[i], work=work[i], count=int(cnt[i])) for i, k in enumerate(names)}


def cpu_baseline(cfg, store, seconds_budget=25.0):
    """Reference-equivalent CPU path (oracle port): train steps at batch 4 on this host's cores."""
    from oracle import fira_oracle as O
    from fira_icse_amd.model import reference_init_state_dict
    # more threads than ~32 only add contention on these small-matrix ops (measured: 256 threads are >100x slower)
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    P = {k: v.clone().requires_grad_(True) for k, v in reference_init_state_dict(cfg).items()}
    opt = torch.optim.Adam(list(P.values()), cfg.lr)
    hb = store.batch(range(4))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    edge = t(hb.dense_edge(cfg.graph_len))
    args = (cfg, t(hb.sou), t(hb.tar), t(hb.mark), t(hb.ast_change), edge, t(hb.tar_label), t(hb.sub_token))
    times = []
    t_all = time.time()
    for i in range(12):
        t0 = time.time()
        ls, nt = O.forward(P, *args, "train")
        opt.zero_grad(set_to_none=True)
        (ls / nt).backward()
        opt.step()
        times.append(time.time() - t0)
        if time.time() - t_all > seconds_budget and len(times) >= 2:
            break
    med = float(np.median(times[1:])) if len(times) > 1 else times[0]
    return {"value": 4.0 / med, "unit": "commits/s", "cores": threads, "kind": "port",
            "sample": "%d train steps (fwd+bwd+Adam, dropout off) at batch 4, fp32, median; dense f64->f32 adjacency "
                      "as the reference feeds it" % len(times)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="commits per GPU per step (BASELINE configs[1])")
    ap.add_argument("--decode-batch", type=int, default=64)
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pool", type=int, default=4, help="distinct resident batches cycled through")
    a = ap.parse_args()

    rank, world, local = init_from_env()
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (a.gpus, a.gpus))
    torch.cuda.set_device(local)
    from fira_icse_amd.model import TransModel, DeviceBatch
    from fira_icse_amd.train import Trainer
    from fira_icse_amd.decode import Searcher

    cfg = FiraConfig()
    B = a.batch
    n_commits = max(a.pool * B, a.decode_batch)
    store = data.process_raw(cfg, synth.generate_dataset(n_commits, seed=1000 + rank))
    torch.manual_seed(0)
    model = TransModel(cfg, device="cuda:%d" % local)
    model.train()                                      # dropout on, as the reference trains
    trainer = Trainer(model, distributed=world > 1)
    batches = [DeviceBatch(store.batch(range(i * B, (i + 1) * B)), cfg, model.device_) for i in range(a.pool)]
    nnz_mean = float(np.mean([b.nnz for b in batches]))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        trainer.step(batches[i % a.pool])
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        trainer.step(batches[i % a.pool])
    t_enq = time.perf_counter() - t0                  # host time to enqueue the steps (the stream is still draining)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=model.device_)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    loss = trainer.last_loss()
    commits_per_s = a.steps * B * world / dt

    # ---- kernel-class timing with HIP events on the launch stream (extra steps, outside the timed region)
    lib = _lib.lib()
    lib.fira_prof_enable(1)
    prof_steps = 3
    for i in range(prof_steps):
        trainer.step(batches[i % a.pool])
    torch.cuda.synchronize()
    prof = prof_report()
    lib.fira_prof_enable(0)
    total_ms = sum(v["ms"] for v in prof.values()) or 1.0
    try:        # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process)
        with open(os.path.join(HERE, "profiles", "traffic.json")) as f:
            traffic = json.load(f)
    except Exception:
        traffic = {}
    gemm, spmm = prof["gemm"], prof["spmm"]
    spmm_bytes = spmm["work"] + 8.0 * nnz_mean * spmm["count"]              # + (col,val) of the batch's nnz
    roofline = {"bound": "mfma", "kernel": "gemm_f32_kernel (v_mfma_f32_32x32x2_f32)",
                "achieved": gemm["work"] / (gemm["ms"] * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": gemm["work"] / (gemm["ms"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
                "traffic": traffic.get("gemm", {}).get("hbm_bytes_per_launch"),
                "algorithmic_flop_per_launch": gemm["work"] / max(gemm["count"], 1),
                "launches_per_step": gemm["count"] // prof_steps, "avg_launch_us": 1e3 * gemm["ms"] / max(gemm["count"], 1),
                "share_of_kernel_time": gemm["ms"] / total_ms}
    spmm_obj = {"bound": "hbm", "kernel": "spmm_rowwave_kernel", "achieved": spmm_bytes / (spmm["ms"] * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": spmm_bytes / (spmm["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic.get("spmm", {}).get("hbm_bytes_per_launch"),
                "avg_launch_us": 1e3 * spmm["ms"] / max(spmm["count"], 1),
                "bytes_per_launch": spmm_bytes / max(spmm["count"], 1), "share_of_kernel_time": spmm["ms"] / total_ms}

    # ---- greedy decode (BASELINE configs[3]): tokens/s, batch 64, encoder included
    decode = None
    if not a.no_decode:
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
        decode = {"tokens_per_s": toks * world / ddt, "commits_per_s": a.decode_batch * world / ddt,
                  "step_tokens_per_s": a.decode_batch * steps_run * world / ddt, "steps_run": steps_run,
                  "batch": a.decode_batch, "beam": 1, "ms_per_batch": ddt * 1e3, "tokens_per_batch": toks,
                  "note": "tokens = emitted tokens up to and including <eos> (SURVEY 8d); step_tokens = batch x decoder "
                          "steps executed, the unit of BASELINE.md's CPU figure (109 step-tokens/s). Random-init weights: "
                          "most hypotheses copy <eos> within a few steps while a few run all 29"}
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

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            cpu = cpu_baseline(cfg, store)
        except Exception as e:                          # the baseline leg must never take the bench line down
            cpu = {"error": repr(e)}

    if rank == 0:
        line = {
            "metric": "training commits/sec (FIRA default config)", "value": commits_per_s, "unit": "commits/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "host_enqueue_ms_per_step": t_enq / a.steps * 1e3,
            "config": {"workload": "BASELINE configs[1]: FIRA training step (fwd+bwd+Adam, dropout 0.1/0.2), "
                                   "batch %d commits/GPU, fp32, 650-node graphs (mean nnz %.0f/graph), vocab 24650" %
                                   (B, nnz_mean / B),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "loss": loss},
            "roofline": roofline, "spmm": spmm_obj, "decode": decode, "cpu_baseline": cpu,
            "kernel_time_ms_per_step": {k: v["ms"] / prof_steps for k, v in prof.items()},
        }
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
