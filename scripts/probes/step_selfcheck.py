import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import util
from fira_icse_amd import data
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import TransModel, DeviceBatch, reference_init_state_dict
from fira_icse_amd.train import Trainer
cfg = FiraConfig()
store = data.process_raw(cfg, util.load_golden_raw())
idx = data.split_index(*util.GOLDEN_SPLIT, seed=0)
hb = store.batch(idx["train"][:util.GOLDEN_B])
sd = util.perturb_state_dict(reference_init_state_dict(cfg), seed=1)
model = TransModel(cfg, init=False)
db = DeviceBatch(hb, cfg)
def run(fused, dtype, steps=3):
    model.compute_dtype = dtype
    model.load_state_dict(sd); model.train(); model.set_dropout_stream(7, 0); model.dropout_step = 0
    tr = Trainer(model); tr.fused_step = fused
    outs = []
    for _ in range(steps):
        tr.step(db); torch.cuda.synchronize()
        outs.append(tr.m.clone())
    return outs
live = model.layout.live
for dtype in ("f32", "bf16"):
    a = run(False, dtype); b = run(False, dtype); c = run(True, dtype); d = run(True, dtype)
    for i in range(3):
        f = lambda x, y: float((x[i][:live] - y[i][:live]).norm() / y[i][:live].norm())
        print(dtype, "step", i + 1, "two-call vs two-call %.2e" % f(a, b), " one-call vs one-call %.2e" % f(c, d), " one vs two %.2e" % f(c, a), flush=True)
