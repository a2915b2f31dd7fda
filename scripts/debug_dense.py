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
db = DeviceBatch(hb, cfg)
dense = DeviceBatch(hb, cfg, skip_padding=False)
l1, n1 = model.train_fwd_bwd(db)
g1 = {k: v.clone() for k, v in model.grad_views().items()}
gb1 = model.gbuf.clone()
l2, n2 = model.train_fwd_bwd(dense)
g2 = model.grad_views()
print("loss", float(l1), float(l2), "total err", float((gb1 - model.gbuf).double().norm() / model.gbuf.double().norm()))
tot = float(model.gbuf.double().norm())
rows = []
for k in g1:
    d = float((g1[k] - g2[k]).double().norm())
    n = float(g2[k].double().norm())
    rows.append((d / tot, d / max(n, 1e-30), k))
rows.sort(reverse=True)
for r in rows[:12]:
    print("%.3e of total  %.3e rel  %s" % r)

def run(b):
    l, n = model.train_fwd_bwd(b)
    torch.cuda.synchronize()
    return float(l), model.gbuf.clone()
res = [run(db), run(db), run(dense), run(dense), run(db)]
tot = float(res[0][1].double().norm())
for i in range(len(res)):
    for j in range(i + 1, len(res)):
        print("run %d vs %d: loss %.10g %.10g  grad err %.3e" % (i, j, res[i][0], res[j][0], float((res[i][1] - res[j][1]).double().norm()) / tot))

c1, s1 = model.encoder(None, None, None, None, None, db, None)
c2, s2 = model.encoder(None, None, None, None, None, dense, None)
v = torch.zeros((db.B * cfg.mem_len,), dtype=torch.bool, device="cuda"); v[db.mem_dst.long()] = True
m1 = torch.cat([c1, s1], 1).view(-1, 256)[v]; m2 = torch.cat([c2, s2], 1).view(-1, 256)[v]
print("memory compact vs dense: max abs", float((m1 - m2).abs().max()), "rel", float((m1 - m2).norm() / m1.norm()), "bitwise", bool(torch.equal(m1, m2)))
d = (m1 - m2).abs().max(1).values
print("rows differing:", int((d > 0).sum()), "of", d.numel())
