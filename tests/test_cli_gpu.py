"""The drop-in command line (python run_model.py train|test) end to end on a tiny synthetic DataSet."""
import os
import subprocess
import sys

import pytest
import torch

import util
from fira_icse_amd import synth

pytestmark = pytest.mark.gpu


def run(args, cwd):
    env = dict(os.environ, PYTHONPATH=util.REPO)
    r = subprocess.run([sys.executable, os.path.join(util.REPO, "run_model.py")] + args, cwd=cwd, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_train_then_test_roundtrip(tmp_path):
    root = str(tmp_path)
    synth.write_dataset(root, util.load_golden_raw())
    out = run(["train", "--splits", "16,4,4", "--batch-size", "4", "--epochs", "3", "--dev-from-epoch", "1",
               "--dev-every", "2", "--save-optimizer"], root)
    assert "epoch: 2 batch: 0/4" in out
    losses = [float(l.split("loss: ")[1].split()[0]) for l in out.splitlines() if "loss: " in l]
    assert losses[-1] < losses[0]                                   # it learns
    sd = torch.load(os.path.join(root, "best_model.pt"), map_location="cpu")
    assert len(sd) == 338 and sd["out_fc.weight"].shape == (24650, 256)          # the reference's checkpoint layout
    proc = open(os.path.join(root, "OUTPUT", "train_process")).read().strip().split("\n")
    assert proc[0].startswith("epoch: 1 batch: 0 dev bleu: ") and proc[0].endswith("is better: True")
    assert len(open(os.path.join(root, "OUTPUT", "dev_output")).read().strip().split("\n")) == 4
    assert os.path.exists(os.path.join(root, "fira_train_state.pt"))
    for beam in ("3", "1"):
        run(["test", "--splits", "16,4,4", "--test-batch-size", "3", "--beam", beam], root)
        lines = open(os.path.join(root, "OUTPUT", "output_fira")).read().split("\n")
        assert len(lines) == 5 and lines[-1] == ""                  # one line per test commit, in all_index order
