"""The single-process 3-step run of tests/test_dp_gpu.py repeated in fresh processes: per step the gradient buffer, the loss and
the first moment are saved; runs are compared pairwise -- which step, which tensors differ when a run takes the other outcome."""
import os, sys, tempfile, pathlib
import torch
import torch.multiprocessing as mp
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))


def run(rank, out):
    import util
    from fira_icse_amd import data
    from fira_icse_amd.config import FiraConfig
    from fira_icse_amd.model import TransModel, DeviceBatch, reference_init_state_dict
    from fira_icse_amd.train import Trainer
    cfg = FiraConfig()
    store = data.process_raw(cfg, util.load_golden_raw())
    idx = data.split_index(*util.GOLDEN_SPLIT, seed=0)["train"]
    torch.manual_seed(0)
    model = TransModel(cfg, init=False)
    model.load_state_dict(util.perturb_state_dict(reference_init_state_dict(cfg), seed=1))
    model.eval()
    trainer = Trainer(model, distributed=False)
    rec = []
    for step in range(3):
        gidx = idx[4 * step:4 * step + 4] if step < 2 else idx[8:9]
        trainer.step(DeviceBatch(store.batch(gidx), cfg))
        torch.cuda.synchronize()
        rec.append({"g": model.gbuf.detach().cpu().clone(), "loss": trainer.last_loss()})
    opt = trainer.state_dict()
    torch.save({"rec": rec, "m": opt["m"].cpu(), "flat": model.flat.data.cpu()}, out)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    d = pathlib.Path(tempfile.mkdtemp())
    outs = []
    for i in range(n):
        o = str(d / ("r%d.pt" % i))
        mp.spawn(run, args=(o,), nprocs=1, join=True)
        outs.append(torch.load(o, weights_only=False))
    from fira_icse_amd.config import FiraConfig
    from fira_icse_amd.model import TransModel
    model = TransModel(FiraConfig(), device="cuda")
    base = model.flat.data.data_ptr()
    views = {n_: ((t.data_ptr() - base) // 4, t.numel()) for n_, t in model.named_views().items()}
    ref = outs[0]
    for i in range(1, n):
        dm = float((outs[i]["m"] - ref["m"]).norm() / ref["m"].norm())
        line = "run %d vs run 0: m %.3e |" % (i, dm)
        for s in range(3):
            g0, g1 = ref["rec"][s]["g"], outs[i]["rec"][s]["g"]
            line += " step %d: g %.3e loss %.3e |" % (s, float((g1 - g0).norm() / g0.norm()), abs(outs[i]["rec"][s]["loss"] - ref["rec"][s]["loss"]))
        print(line)
        if dm > 1e-5:
            for s in range(3):
                g0, g1 = ref["rec"][s]["g"], outs[i]["rec"][s]["g"]
                rows = []
                for n_, (o, k) in views.items():
                    x, y = g0[o:o + k], g1[o:o + k]
                    nx = float(x.norm())
                    rows.append((float((x - y).norm()), nx, n_))
                rows.sort(reverse=True)
                print("   step %d largest |dg| (tensor norm):" % s, ["%s %.2e (%.2e)" % (n_, e, nx) for e, nx, n_ in rows[:10]])
                if s == 1:                                  # WHERE inside the worst tensors
                    for e, nx, n_ in rows[:4]:
                        o, k = views[n_]
                        shp = tuple(model.named_views()[n_].shape)
                        if len(shp) != 2:
                            continue
                        x, y = g0[o:o + k].view(shp), g1[o:o + k].view(shp)
                        d = (x - y).abs()
                        thr = 1e-4 * float(x.abs().max())
                        rr = (d.max(1).values > thr).nonzero().flatten().tolist()
                        cc = (d.max(0).values > thr).nonzero().flatten().tolist()
                        print("      %s %s: rows off %d %s ... cols off %d %s" % (n_, shp, len(rr), rr[:40], len(cc), cc[:40]))
