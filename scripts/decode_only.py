"""Search loops alone (no training): greedy batch 64 and beam-3 batch 20 on freshly initialised weights (every hypothesis
runs all 29 steps), a few repetitions each; prints ms per batch / per step.  Used for A/B switches (FIRA_DECODE_ATTN, ...)
and as the command of the decode rocprofv3 trace."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import data, synth
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import TransModel, DeviceBatch
from fira_icse_amd.decode import Searcher


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    cfg = FiraConfig()
    store = data.process_raw(cfg, synth.generate_dataset(64, seed=1000))
    torch.manual_seed(0)
    model = TransModel(cfg)
    model.eval()
    search = Searcher(model)
    db = DeviceBatch(store.batch(range(64)), cfg)
    for _ in range(2):
        out, length, p = search.greedy(db)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out, length, p = search.greedy(db)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    steps = int(length.max().item()) - 1
    print("greedy batch 64: %.3f ms per batch, %d steps, %.4f ms per step" % (dt * 1e3, steps, dt * 1e3 / max(steps, 1)), flush=True)
    db20 = DeviceBatch(store.batch(range(20)), cfg)
    for _ in range(2):
        gen, blen, bp = search.beam(db20, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        gen, blen, bp = search.beam(db20, 3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    steps = int(blen.max().item()) - 1
    print("beam-3 batch 20: %.3f ms per batch, %d steps, %.4f ms per step" % (dt * 1e3, steps, dt * 1e3 / max(steps, 1)), flush=True)


if __name__ == "__main__":
    main()
