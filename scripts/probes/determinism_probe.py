"""Bit-reproducibility of the fused encoder kernels (no atomics in them): the same launch N times on the same inputs, every output
compared bitwise with the first.  A race inside a kernel (e.g. a barrier that waits for too little) shows up as a mismatch."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from fira_icse_amd import data, synth, ops
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import DeviceBatch
N_REP = int(sys.argv[1]) if len(sys.argv) > 1 else 200
DEV = "cuda"
torch.manual_seed(0)
rn = lambda *s, scale=1.0: torch.randn(*s, device=DEV) * scale
cfg = FiraConfig()
store = data.process_raw(cfg, synth.generate_dataset(64, seed=1000))
db = DeviceBatch(store.batch(list(range(64))), cfg, DEV)
rp, c, v, n = db.rowptr, db.col, db.val, db.n_nodes


def rep(name, fn):
    first, bad, which, worst = None, 0, {}, {}
    for i in range(N_REP):
        out = [t.clone() for t in fn()]
        if first is None:
            first = out
        else:
            eq = [torch.equal(a, b) for a, b in zip(out, first)]
            if not all(eq):
                bad += 1
                for k, e in enumerate(eq):
                    if not e:
                        which[k] = which.get(k, 0) + 1
                        worst[k] = max(worst.get(k, 0.0), float((out[k] - first[k]).abs().max() / first[k].abs().max()))
    torch.cuda.synchronize()
    print("%-34s %d launches, %d differ from the first %s" % (name, N_REP, bad, {k: (which[k], "%.1e" % worst[k]) for k in which}), flush=True)


X = rn(n, 256)
W21, b2, c21 = rn(256, 256, scale=0.06), rn(256, scale=0.1), rn(256, scale=0.1)
gamma, beta = 1 + rn(256, scale=0.1), rn(256, scale=0.1)
W21t = W21.t().contiguous()
dY, dX0 = rn(n, 256), rn(n, 256)
for dt in (2, 3):
    rep("gcn_layer_fwd dtype %d (%d rows)" % (dt, n), lambda: ops.gcn_layer_fwd(rp, c, v, X, W21t, b2, c21, gamma, beta, dropout=0.2, seed=7, site=3, dtype=dt)[:3])

    def bwd():
        dX = dX0.clone()
        V = ops.gcn_layer_bwd(rp, c, v, dY, W21, dX, dtype=dt)
        return V, dX
    rep("gcn_layer_bwd dtype %d" % dt, bwd)
nc = 10000
Xc = rn(nc, 256)
Wqk, bqk = rn(512, 256, scale=0.08), rn(512, scale=0.1)
Wo, bo = rn(256, 256, scale=0.08), rn(256, scale=0.1)
vtab = rn(4, 256)
mark = torch.randint(0, 4, (nc,), device=DEV, dtype=torch.int32)
rows = torch.randperm(2 * nc, device=DEV)[:nc].to(torch.int32)
dG0 = rn(2 * nc, 256)
for dt in (2, 3):
    fw = lambda: ops.combination_block_fwd(Xc, Wqk, bqk, Wo, bo, vtab, mark, gamma, beta, dropout=0.1, seed=5, site_gate=17, site_out=18, dtype=dt)
    rep("combination_block_fwd dtype %d" % dt, fw)
    qk, cc, summ, y, stats = fw()
    rep("combination_block_bwd dtype %d" % dt, lambda: ops.combination_block_bwd(dG0.clone(), rows, summ, stats, gamma, Wo, Wqk, qk, vtab, mark, dropout=0.1,
                                                                                 seed=5, site_gate=17, site_out=18, dtype=dt))
