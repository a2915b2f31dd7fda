"""Probe: do two independent half-batch training steps on two HIP streams (two host threads) finish sooner than one
full-batch step on one stream?  Measures how much of the step is kernel-boundary latency that a second in-flight
chain can hide.  (Two full model replicas here; an in-library version would share the parameters.)"""
import os, sys, threading, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import data, synth
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import TransModel, DeviceBatch
from fira_icse_amd.train import Trainer


def run(n_streams, B, steps=30, warmup=5):
    cfg = FiraConfig()
    store = data.process_raw(cfg, synth.generate_dataset(128, seed=1000))
    workers = []
    for w in range(n_streams):
        torch.manual_seed(0)
        model = TransModel(cfg)
        model.train()
        tr = Trainer(model)
        batches = [DeviceBatch(store.batch(range((4 * w + i) * B % 96, (4 * w + i) * B % 96 + B)), cfg) for i in range(4)]
        workers.append((model, tr, batches, torch.cuda.Stream()))
    torch.cuda.synchronize()
    bar = threading.Barrier(n_streams + 1)

    def loop(model, tr, batches, stream):
        with torch.cuda.stream(stream):
            for i in range(warmup):
                tr.step(batches[i % 4])
            stream.synchronize()
            bar.wait()
            for i in range(steps):
                tr.step(batches[i % 4])
            stream.synchronize()
        bar.wait()

    ths = [threading.Thread(target=loop, args=w) for w in workers]
    for t in ths:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for t in ths:
        t.join()
    print("streams %d x batch %d: %.3f ms per round, %.0f commits/s" % (n_streams, B, dt / steps * 1e3,
                                                                      n_streams * B * steps / dt), flush=True)


if __name__ == "__main__":
    run(1, 32)
    run(2, 16)
    run(2, 32)
    run(4, 8)
