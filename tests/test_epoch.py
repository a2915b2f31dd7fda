"""BASELINE configs[0] / SURVEY.md §8(d) row 1: one epoch of the reference's own train() (run_model.py:83-117, loader at
:387) on the 128-commit synthetic DataSet, batch 4, dropout 0 -- batch ORDER (the RandomSampler permutation drawn from the
global torch RNG after the weight initialisation) and per-batch loss, pinned by tests/golden/epoch_ref.json
(tests/golden/make_golden_epoch.py ran the reference)."""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

import util
from fira_icse_amd import data, synth
from fira_icse_amd.config import FiraConfig


def golden():
    with open(os.path.join(util.GOLDEN, "epoch_ref.json")) as f:
        return json.load(f)


def seed_everything(seed=0):            # run_model.py:65-72 of the reference (the CLI's own copy is in /run_model.py)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def test_split_and_batch_order_equal_the_reference_loader():
    """Host logic only: the split shuffle, the weight initialisation's RNG consumption and the epoch permutation."""
    from fira_icse_amd.model import reference_init_state_dict
    g = golden()
    cfg = FiraConfig()
    seed_everything(0)
    idx = data.split_index(*util.EPOCH_SPLIT, seed=0)
    assert idx["train"] == g["index"]["train"]
    reference_init_state_dict(cfg)                              # TransModel(args): run_model.py:389
    order = list(data.iterate_batches(util.EPOCH_SPLIT[0], util.EPOCH_B, shuffle=True))
    assert order == g["order"]


def test_oracle_loss_of_the_first_batch():
    """The oracle on the epoch's first batch with the freshly initialised weights == the reference's first loss."""
    from fira_icse_amd.model import reference_init_state_dict
    from oracle import fira_oracle as O
    g = golden()
    cfg = FiraConfig()
    torch.set_num_threads(8)
    store = data.process_raw(cfg, synth.generate_dataset(util.EPOCH_N, seed=util.EPOCH_SEED))
    seed_everything(0)
    sd = reference_init_state_dict(cfg)
    train = g["index"]["train"]
    hb = store.batch([train[i] for i in g["order"][0]])
    tb = util.to_torch_batch(hb, cfg)
    with torch.no_grad():
        ls, nt = O.forward(sd, cfg, tb["sou"], tb["tar"], tb["mark"], tb["ast_change"], tb["edge"], tb["tar_label"],
                           tb["sub_token"], "train")
    assert int(nt) == g["n_tok"][0]
    assert abs(float(ls) / int(nt) - g["loss"][0]) < 1e-5 * g["loss"][0]


@pytest.mark.gpu
def test_cli_epoch_equals_the_reference_epoch(tmp_path):
    """`python run_model.py train` for one epoch: same batches in the same order, loss per batch within 2e-4 (24 Adam
    steps: the fp32 re-association of every step feeds the next)."""
    g = golden()
    root = str(tmp_path)
    synth.write_dataset(root, synth.generate_dataset(util.EPOCH_N, seed=util.EPOCH_SEED))
    log = os.path.join(root, "loss.jsonl")
    env = dict(os.environ, PYTHONPATH=util.REPO)
    r = subprocess.run([sys.executable, os.path.join(util.REPO, "run_model.py"), "train", "--splits",
                        ",".join(str(x) for x in util.EPOCH_SPLIT), "--batch-size", str(util.EPOCH_B), "--epochs", "1",
                        "--no-dropout", "--loss-log", log], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    with open(os.path.join(root, "all_index")) as f:
        assert json.load(f)["train"] == g["index"]["train"]
    rows = [json.loads(l) for l in open(log)]
    assert [row["index"] for row in rows] == g["order"]
    got = np.array([row["loss"] for row in rows])
    want = np.array(g["loss"])
    assert got.shape == want.shape
    assert abs(got[0] - want[0]) < 1e-5 * want[0], (got[0], want[0])
    assert np.max(np.abs(got - want) / want) < 2e-4, np.max(np.abs(got - want) / want)
    assert want[-1] < want[0]                                    # (and it learns)
