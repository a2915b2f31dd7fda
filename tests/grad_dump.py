"""Helper of tests/test_model_gpu.py::test_encoder_wgrad_schedules_give_the_same_gradient: the gradient of the golden batch
under whatever FIRA_* switches the parent put in the environment (they are read once per process), saved to argv[1]."""
import sys

import numpy as np
import torch

import util
from fira_icse_amd import data
from fira_icse_amd.config import FiraConfig


def main(path, dtype):
    from fira_icse_amd.model import TransModel, DeviceBatch, reference_init_state_dict
    cfg = FiraConfig()
    store = data.process_raw(cfg, util.load_golden_raw())
    idx = data.split_index(*util.GOLDEN_SPLIT, seed=0)
    hb = store.batch(idx["train"][:util.GOLDEN_B])
    torch.manual_seed(0)
    sd = util.perturb_state_dict(reference_init_state_dict(cfg), seed=1)
    model = TransModel(cfg, init=False)
    model.load_state_dict(sd)
    model.compute_dtype = dtype
    model.eval()
    db = DeviceBatch(hb, cfg)
    loss, ntok = model.train_fwd_bwd(db)
    torch.cuda.synchronize()
    out = dict(loss=float(loss), g=model.gbuf[:model.layout.live].cpu().numpy())
    # dropout on: the masks are a function of (seed, rank, step) -- the same in every process
    model.train()
    model.set_dropout_stream(5, 0)
    model.dropout_step = 0
    loss, ntok = model.train_fwd_bwd(db)
    torch.cuda.synchronize()
    out.update(loss_train=float(loss), g_train=model.gbuf[:model.layout.live].cpu().numpy())
    np.savez(path, **out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "f32")
