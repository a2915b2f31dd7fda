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
out = {}
for name, b in (("compact", DeviceBatch(hb, cfg)), ("dense", DeviceBatch(hb, cfg, skip_padding=False))):
    model.train_fwd_bwd(b)
    torch.cuda.synchronize()
    out[name] = {k: v.clone().cpu() for k, v in model.grad_views().items()}
    print(name, "n_code", b.n_code, "n_nodes", b.n_nodes, "n_mem", b.n_mem)
tag = sys.argv[1]
torch.save(out, "/tmp/grads_%s.pt" % tag)
if len(sys.argv) > 2:
    other = torch.load("/tmp/grads_%s.pt" % sys.argv[2])
    for mode in ("compact", "dense"):
        tot = sum(float(v.double().norm()) ** 2 for v in out[mode].values()) ** 0.5
        rows = sorted(((float((out[mode][k] - other[mode][k]).double().norm()) / tot, float((out[mode][k] - other[mode][k]).double().norm()) / max(float(out[mode][k].double().norm()), 1e-30), k) for k in out[mode]), reverse=True)
        print(mode, "%s vs %s: total %.3e" % (tag, sys.argv[2], sum(r[0] ** 2 for r in rows) ** 0.5))
        for r in rows[:6]:
            print("   %.3e of total %.3e rel %s" % r)
