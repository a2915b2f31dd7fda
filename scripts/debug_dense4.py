import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import util
from fira_icse_amd import data
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import TransModel, DeviceBatch, reference_init_state_dict
cfg = FiraConfig()
store = data.process_raw(cfg, util.load_golden_raw())
idx = data.split_index(*util.GOLDEN_SPLIT, seed=0)
hb = store.batch(idx["train"][:util.GOLDEN_B])
torch.manual_seed(0)
sd = util.perturb_state_dict(reference_init_state_dict(cfg), seed=1)
model = TransModel(cfg, init=False)
model.load_state_dict(sd)
model.eval()
for name, skip in (("compact", True), ("dense", False)):
    b = DeviceBatch(hb, cfg, skip_padding=skip)
    c1, s1 = model.encoder(None, None, None, None, None, b, None)
    mem = torch.cat([c1, s1], 1).view(-1, 256)
    model.train_fwd_bwd(b)
    torch.cuda.synchronize()
    ws = model.workspace(b.B, 1).view(torch.float32)
    found = 0
    rows = [0, 1, 5, 100, 210, 215, 370, 371, 800]
    for r in rows:
        v = mem[r]
        if float(v.abs().max()) == 0:
            continue
        cand = torch.nonzero(ws == v[0]).flatten()
        ok = False
        for c in cand.tolist():
            if c + 256 <= ws.numel() and bool(torch.equal(ws[c:c + 256], v)):
                ok = True
                break
        found += ok
        print(name, "eval memory row", r, "found bitwise in the training workspace:", ok, "(candidates %d)" % cand.numel())
