"""Greedy search with 1..4 batches of 64 in flight (Searcher.greedy_many), freshly initialised weights (29 steps each)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import data, synth
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import TransModel, DeviceBatch
from fira_icse_amd.decode import Searcher

cfg = FiraConfig()
store = data.process_raw(cfg, synth.generate_dataset(256, seed=1000))
torch.manual_seed(0)
model = TransModel(cfg)
n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if n_train:                                            # the bench's situation: a trainer (its streams, workspace) came first
    from fira_icse_amd.train import Trainer
    model.train()
    tr = Trainer(model)
    tb = [DeviceBatch(store.batch(range(32 * i, 32 * i + 32)), cfg) for i in range(4)]
    for i in range(n_train):
        tr.step(tb[i % 4])
    torch.cuda.synchronize()
    print("trained %d steps, loss %.3f" % (n_train, tr.last_loss()), flush=True)
model.eval()
NB = int(os.environ.get("N_BATCH", "4"))                # batches per group (256 synthetic commits: reused modulo 4)
dbs = [DeviceBatch(store.batch(range(64 * (i % 4), 64 * (i % 4) + 64)), cfg) for i in range(NB)]
for n in [int(x) for x in os.environ.get("N_LANES", "1,2,3,4").split(",")]:
    s = Searcher(model)
    for _ in range(2):
        s.greedy_many(dbs, in_flight=n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        s.greedy_many(dbs, in_flight=n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("in flight %d: %.2f ms for %d batches of 64 x 29 steps = %.4f ms per batch-step, %.0f step-tokens/s" %
          (n, dt * 1e3, NB, dt * 1e3 / (NB * 29), NB * 64 * 29 / dt), flush=True)
