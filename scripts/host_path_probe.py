"""Host data path probe: per-batch cost of collate + DeviceBatch (pinned arena + one H2D copy), no training."""
import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import data, synth
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import DeviceBatch

cfg = FiraConfig()
store = data.process_raw(cfg, synth.generate_dataset(256, seed=3))
for B in (32, 64):
    idx = [list(range(i * B % 256, i * B % 256 + B)) for i in range(40)]
    for i in idx[:3]:
        DeviceBatch(store.batch(i), cfg, "cuda:0")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in idx:
        hb = store.batch(i)
    t1 = time.perf_counter()
    for i in idx:
        db = DeviceBatch(hb, cfg, "cuda:0")
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B=%d collate %.3f ms  DeviceBatch %.3f ms" % (B, (t1 - t0) / 40 * 1e3, (t2 - t1) / 40 * 1e3))
pr = cProfile.Profile(); pr.enable()
for i in idx[:20]:
    db = DeviceBatch(store.batch(i), cfg, "cuda:0")
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
