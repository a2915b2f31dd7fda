"""Gradients of train_fwd_bwd with and without a mid event, per tensor (the decoder's weight gradients go out per 3 layers or in
one launch behind the loop depending on it: FIRA_DEC_WGRAD_DP)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fira_icse_amd import data, synth
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import TransModel, DeviceBatch

cfg = FiraConfig()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
store = data.process_raw(cfg, synth.generate_dataset(8, seed=1000))
torch.manual_seed(0)
model = TransModel(cfg)
model.eval()
db = DeviceBatch(store.batch(range(B)), cfg)
mid = torch.cuda.Event(); mid.record()
model.train_fwd_bwd(db, zero_grad=True)
torch.cuda.synchronize()
g0 = model.gbuf.clone()
model.train_fwd_bwd(db, zero_grad=True, mid_event=mid)
torch.cuda.synchronize()
g1 = model.gbuf.clone()
base = model.flat.data.data_ptr()
bad = []
for n, t in model.named_views().items():
    o = (t.data_ptr() - base) // 4
    a, b = g0[o:o + t.numel()], g1[o:o + t.numel()]
    den = float(a.norm())
    if den == 0:
        continue
    r = float((a - b).norm()) / den
    if r > 1e-5:
        bad.append((n, r))
print("batch %d: %d tensors differ by more than 1e-5" % (B, len(bad)), bad[:10])
