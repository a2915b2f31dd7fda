"""Probe: one training step (fixed batch -> static shapes) captured into a hipGraph and replayed, against the same
step launched eagerly.  Tells how much of the step is host launch cost / inter-kernel dispatch latency that a graph
removes on this ROCm build."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import data, synth
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import TransModel, DeviceBatch
from fira_icse_amd.train import Trainer


def main():
    cfg = FiraConfig()
    store = data.process_raw(cfg, synth.generate_dataset(64, seed=1000))
    torch.manual_seed(0)
    model = TransModel(cfg)
    model.train()
    tr = Trainer(model)
    db = DeviceBatch(store.batch(range(32)), cfg)
    for _ in range(5):
        tr.step(db)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        tr.step(db)
    torch.cuda.synchronize()
    print("eager: %.3f ms/step" % ((time.perf_counter() - t0) / 30 * 1e3), flush=True)
    for env in ("0", "1"):
        os.environ["FIRA_NO_WGRAD_OVERLAP"] = env
    os.environ["FIRA_NO_WGRAD_OVERLAP"] = "0"
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        tr.step(db)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            tr.step(db)
        for _ in range(5):
            g.replay()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            g.replay()
        s.synchronize()
        print("graph replay: %.3f ms/step (loss %.4f)" % ((time.perf_counter() - t0) / 30 * 1e3, tr.last_loss()), flush=True)


if __name__ == "__main__":
    main()
