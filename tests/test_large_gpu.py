"""Whole-model parity at BASELINE batch sizes against fixtures produced by the
reference itself (tests/golden/make_golden_large.py -> large_ref.json):

  * 20 Adam steps at batch 32, dropout off: the mean token loss of every step (the reference's fp32 CPU run);
  * greedy search at test batch 64: the reference's own output lines (`run_model.test`, beam_size 1);
  * beam-3 search over 32 commits: the reference's output lines.

Tolerance of the loss curve: Adam divides by sqrt(v) and so turns gradient entries that are pure rounding noise on both
sides (|g| ~ 1e-9, e.g. the attention key biases) into +-lr steps of either sign; over 20 steps the two fp32
implementations drift apart by a few 1e-5 relative in the loss.  The gate is 2e-4 per step (the 3-step golden test
uses the same bound).
"""
import json
import os

import numpy as np
import pytest
import torch

import util
from fira_icse_amd import data, synth, text
from fira_icse_amd.config import FiraConfig

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def large():
    from fira_icse_amd.model import reference_init_state_dict
    cfg = FiraConfig()
    raw = synth.generate_dataset(util.LARGE_N, seed=util.LARGE_SEED)
    store = data.process_raw(cfg, raw)
    idx = data.split_index(*util.LARGE_SPLIT, seed=0)
    gold = json.load(open(os.path.join(util.GOLDEN, "large_ref.json")))
    assert idx["train"] == gold["index"]["train"] and idx["test"] == gold["index"]["test"]     # the reference's split
    torch.manual_seed(0)
    sd = reference_init_state_dict(cfg)
    return cfg, raw, store, idx, gold, sd


def test_loss_curve_20_steps_batch32_matches_reference(large):
    from fira_icse_amd.model import TransModel, DeviceBatch
    from fira_icse_amd.train import Trainer
    cfg, raw, store, idx, gold, sd = large
    model = TransModel(cfg, init=False)
    model.load_state_dict(util.perturb_state_dict({k: v.clone() for k, v in sd.items()}, seed=1))
    model.eval()                                             # dropout off, as in the fixture
    trainer = Trainer(model)
    batches = [DeviceBatch(store.batch(idx["train"][i * 32:(i + 1) * 32]), cfg) for i in range(4)]
    curve, ntok = [], []
    for it in range(20):
        trainer.step(batches[it % 4])
        curve.append(trainer.last_loss())
        ntok.append(int(model.n_tok.item()))
    assert ntok == gold["n_tok"]
    ref = np.array(gold["loss_curve"])
    dev = np.abs(np.array(curve) - ref) / ref
    assert dev[0] < 1e-5, dev[0]                             # step 0: pure forward parity
    assert dev.max() < 2e-4, (float(dev.max()), int(dev.argmax()))


@pytest.fixture(scope="module")
def searcher(large):
    from fira_icse_amd.model import TransModel
    from fira_icse_amd.decode import Searcher
    cfg, raw, store, idx, gold, sd = large
    model = TransModel(cfg, init=False)
    model.load_state_dict(util.peaked_state_dict({k: v.clone() for k, v in sd.items()}, seed=2))
    model.eval()
    return model, Searcher(model)


def _lines(search, out, raw, ids):
    r_vocab = {v: k for k, v in raw["word_vocab"].items()}
    return [text.detokenize(h, r_vocab, raw["variable"][i]) for h, i in zip(search.best(*out), ids)]


def test_greedy_batch64_token_ids_match_reference(large, searcher):
    from fira_icse_amd.model import DeviceBatch
    cfg, raw, store, idx, gold, sd = large
    model, search = searcher
    ids = idx["test"][:64]
    db = DeviceBatch(store.batch(ids), cfg)
    got = _lines(search, search.greedy(db), raw, ids)
    bad = [i for i, (a, b) in enumerate(zip(got, gold["greedy64"])) if a != b]
    assert not bad, (bad, got[bad[0]], gold["greedy64"][bad[0]])
    assert _lines(search, search.greedy(db, use_graphs=False), raw, ids) == gold["greedy64"]


def test_beam3_batch32_lines_match_reference(large, searcher):
    from fira_icse_amd.model import DeviceBatch
    cfg, raw, store, idx, gold, sd = large
    model, search = searcher
    ids = idx["test"][:32]
    db = DeviceBatch(store.batch(ids), cfg)
    got = _lines(search, search.beam(db, 3), raw, ids)
    bad = [i for i, (a, b) in enumerate(zip(got, gold["beam3_32"])) if a != b]
    assert not bad, (bad, got[bad[0]], gold["beam3_32"][bad[0]])


def test_search_under_exact_ties_equals_the_reference_up_to_the_tied_twins(large):
    """Twin vocabulary rows (util.tie_state_dict): hypotheses that differ in a twin token have bit-identical probabilities,
    so which of them survive -- and which one is printed -- is decided by tie-breaking alone.  The fixture
    (tests/golden/make_golden_tie.py -> tie_ref.json) holds the reference's own output lines.

    The reference breaks ties with ``torch.sort(descending=True)`` (run_model.py:305), an UNSTABLE sort: on this very data
    it emits the higher twin in one place and the lower in another (and a CUDA run of the reference orders them
    differently again), so there is no rule to reproduce.  Ours is fixed -- probability descending, then flattened
    (beam, token) index ascending (beam.hip) -- and the parity statement is: after mapping every twin to its even
    representative the lines are IDENTICAL, i.e. the searches differ only inside classes of exactly tied candidates
    (odd twins still appear in our lines where they are COPIED from the input: copy slots are separate candidates)."""
    from fira_icse_amd.model import TransModel, DeviceBatch
    from fira_icse_amd.decode import Searcher
    cfg, raw, store, idx, gold, sd = large
    tie = json.load(open(os.path.join(util.GOLDEN, "tie_ref.json")))
    lo, hi = util.TIE_TWINS
    assert (lo, hi) == tuple(tie["twins"])
    vocab = raw["word_vocab"]
    r_vocab = {v: k for k, v in vocab.items()}

    def canon(line):
        out = []
        for w in line.split():
            i = vocab.get(w)
            out.append(r_vocab[i - ((i - lo) & 1)] if i is not None and lo <= i < hi else w)
        return " ".join(out)

    model = TransModel(cfg, init=False)
    model.load_state_dict(util.tie_state_dict({k: v.clone() for k, v in sd.items()}, seed=2))
    model.eval()
    search = Searcher(model)
    ids = idx["test"][:16]
    db = DeviceBatch(store.batch(ids), cfg)
    for got, want in ((_lines(search, search.greedy(db), raw, ids), tie["greedy_tie16"]),
                      (_lines(search, search.beam(db, 3), raw, ids), tie["beam3_tie16"])):
        assert [canon(l) for l in got] == [canon(l) for l in want], \
            [(a, b) for a, b in zip(got, want) if canon(a) != canon(b)][:2]
        assert any(a != b for a, b in zip(got, want))         # the fixture does exercise ties (the reference chose odd twins)
