"""tests/test_dp_gpu.py's three two-rank comparisons in file order in ONE parent process, then the per-tensor difference of the
ZeRO-1 run's first moment (the test fails only in that order)."""
import os, sys, tempfile, pathlib
import torch
import torch.multiprocessing as mp
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_dp_gpu as T
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import TransModel

if __name__ == "__main__":
    order = sys.argv[1] if len(sys.argv) > 1 else "abz"
    for ch in order:
        d = pathlib.Path(tempfile.mkdtemp())
        one, two = str(d / "one.pt"), str(d / "two.pt")
        port = {"a": 29811, "b": 29361, "z": 29511}[ch]
        mp.spawn(T._run, args=(1, port, one), nprocs=1, join=True)
        args = {"a": (2, port + 1, two), "b": (2, port + 1, two, False, "gloo", "bf16"), "z": (2, port + 1, two, True)}[ch]
        mp.spawn(T._run, args=args, nprocs=2, join=True)
        a, b = torch.load(one, weights_only=False), torch.load(two, weights_only=False)
        if ch == order[0]:
            gold = a                                    # the first single-process run: which side of a later pair moved?
        else:
            rd = lambda x, y: float((x["m"] - y["m"]).norm() / y["m"].norm())
            print("   ", ch, "single-process run vs the first one: %.3e;  two-rank run vs the first single-process run: %.3e" % (rd(a, gold), rd(b, gold)))
        print(ch, "rel diff m %.3e  v %.3e  flat %.3e" % tuple(float((a[k] - b[k]).norm() / a[k].norm()) for k in ("m", "v", "flat")))
    model = TransModel(FiraConfig(), device="cuda")
    base = model.flat.data.data_ptr()
    rows = []
    for n, t in model.named_views().items():
        o = (t.data_ptr() - base) // 4
        x, y = a["m"][o:o + t.numel()], b["m"][o:o + t.numel()]
        rows.append((float((x - y).norm()), float(x.norm()), n))
    rows.sort(reverse=True)
    for dn, xn, n in rows[:6]:
        print("%-55s diff %.3e  norm %.3e" % (n, dn, xn))
    # WHERE the worst tensors differ: count and index pattern of the elements that are off
    views = model.named_views()
    for dn, xn, n in rows[:3]:
        t = views[n]
        o = (t.data_ptr() - base) // 4
        x, y = a["m"][o:o + t.numel()].view(t.shape), b["m"][o:o + t.numel()].view(t.shape)
        d = (x - y).abs()
        bad = d > 1e-3 * x.abs().max()
        print(n, tuple(t.shape), "elements off by > 1e-3 of max:", int(bad.sum()), "of", t.numel(), " max |diff| %.3e  max |m| %.3e" % (float(d.max()), float(x.abs().max())))
        if bad.any():
            r, c = bad.nonzero(as_tuple=True)
            print("   rows", sorted(set(r.tolist()))[:24], "... cols", sorted(set(c.tolist()))[:24])
        rel = d.sum(1) / x.abs().sum(1).clamp_min(1e-30)
        print("   per-row relative L1 diff: min %.2e  median %.2e  max %.2e" % (float(rel.min()), float(rel.median()), float(rel.max())))
