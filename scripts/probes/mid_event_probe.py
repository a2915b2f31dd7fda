"""Is every gradient of [0, split) final when the mid event fires?  train_fwd_bwd with a mid event; a second stream waits for
the event and snapshots gbuf[:split]; after the call has finished the snapshot must equal the buffer."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fira_icse_amd import data, synth
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import TransModel, DeviceBatch

cfg = FiraConfig()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
store = data.process_raw(cfg, synth.generate_dataset(2 * B, seed=1000))
torch.manual_seed(0)
model = TransModel(cfg)
model.train()
db = DeviceBatch(store.batch(range(B)), cfg)
split = model.layout.split
mid = torch.cuda.Event(); mid.record()
side = torch.cuda.Stream()
snap = torch.empty(split, device="cuda")
for it in range(3):
    model.train_fwd_bwd(db, zero_grad=True, mid_event=mid)
    with torch.cuda.stream(side):
        side.wait_event(mid)
        snap.copy_(model.gbuf[:split])
    torch.cuda.synchronize()
    d = (snap - model.gbuf[:split])
    nz = torch.nonzero(d).flatten()
    print("iter %d: %d of %d elements of [0, split) changed after the mid event" % (it, nz.numel(), split), end="")
    if nz.numel():
        v = model.named_views()
        base = model.flat.data.data_ptr()
        lo, hi = int(nz.min()), int(nz.max())
        names = [n for n, t in v.items() if (t.data_ptr() - base) // 4 <= hi and (t.data_ptr() - base) // 4 + t.numel() > lo]
        hit = [n for n, t in v.items() if d[(t.data_ptr() - base) // 4:(t.data_ptr() - base) // 4 + t.numel()].abs().sum() > 0
               and (t.data_ptr() - base) // 4 < split]
        print("; tensors:", hit[:12])
    else:
        print()
