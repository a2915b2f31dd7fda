"""The data-parallel step's code path on ONE device (world size 1 would skip the collectives: two ranks' worth of schedule is
emulated by calling the two-call step directly): fira_train_step_begin(_rows) / _end(_rows) with the mid event, as Trainer.step
does for a rank, without the all-reduce.  Times the step at batch 32 / 64."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fira_icse_amd import data, synth, ops
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import TransModel, DeviceBatch
from fira_icse_amd.train import Trainer

cfg = FiraConfig()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
store = data.process_raw(cfg, synth.generate_dataset(4 * B, seed=1000))
torch.manual_seed(0)
model = TransModel(cfg)
model.train()
tr = Trainer(model)                      # single-device trainer object: its buffers; the step below is the DP one
dbs = [DeviceBatch(store.batch(range(B * i, B * i + B)), cfg) for i in range(4)]
mid = torch.cuda.Event(); mid.record()
ev = torch.cuda.Event(); ev.record()
stats = torch.zeros(2, device="cuda")
split, live = model.layout.split, model.layout.live

def step(i):
    tr.t += 1
    rows = (tr.m, tr.v, tr.lr, tr.t, 0.9, 0.999, 1e-8, tr.row_step) if tr.row_step is not None else None
    loss_sum, n_tok = model.train_step_begin(dbs[i % 4], mid, rows=rows)
    ops.pack_stats(loss_sum, n_tok, stats)
    ev.record()
    model.train_step_end(tr.m, tr.v, tr.lr, tr.t, early_event=ev, count=stats[1:2], row_step=tr.row_step)
    tr._rows_hyper = (tr.lr, 0.9, 0.999, 1e-8)
    tr._adam_slice(split, live, stats[1:2], table=1)

for i in range(8):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 40
for i in range(N):
    step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print("two-call step, batch %d: %.3f ms, %.0f commits/s" % (B, dt * 1e3, B / dt))
