"""Per-tensor difference of the Adam first moment between one process and two ZeRO-1 ranks (tests/test_dp_gpu.py's runs)."""
import os, sys, tempfile
import torch
import torch.multiprocessing as mp
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_dp_gpu as T
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import TransModel

if __name__ == "__main__":
    d = tempfile.mkdtemp()
    one, two = os.path.join(d, "one.pt"), os.path.join(d, "two.pt")
    mp.spawn(T._run, args=(1, 29711, one), nprocs=1, join=True)
    mp.spawn(T._run, args=(2, 29712, two, True), nprocs=2, join=True)
    a, b = torch.load(one, weights_only=False), torch.load(two, weights_only=False)
    model = TransModel(FiraConfig(), device="cuda")
    base = model.flat.data.data_ptr()
    rows = []
    for n, t in model.named_views().items():
        o = (t.data_ptr() - base) // 4
        x, y = a["m"][o:o + t.numel()], b["m"][o:o + t.numel()]
        rows.append((float((x - y).norm()), float(x.norm()), n))
    rows.sort(reverse=True)
    tot = float(a["m"].norm())
    print("total |m| %.4e, |diff| %.4e" % (tot, float((a["m"] - b["m"]).norm())))
    for dn, xn, n in rows[:8]:
        print("%-55s diff %.3e  norm %.3e" % (n, dn, xn))
