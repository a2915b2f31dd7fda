"""Decode parity: KV-cached greedy / beam search of the HIP engine against (a) the reference's own output lines
(tests/golden/decode_ref.json, produced by run_model.test on the peaked weights) and (b) the CPU oracle's search."""
import json
import os

import numpy as np
import pytest
import torch

import search_ref
import util
from fira_icse_amd import data, text
from fira_icse_amd.config import FiraConfig

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from fira_icse_amd.model import TransModel, DeviceBatch, reference_init_state_dict
    from fira_icse_amd.decode import Searcher
    cfg = FiraConfig()
    raw = util.load_golden_raw()
    store = data.process_raw(cfg, raw)
    idx = data.split_index(*util.GOLDEN_SPLIT, seed=0)
    ids = idx["test"][:util.GOLDEN_B]
    hb = store.batch(ids)
    torch.manual_seed(0)
    sd = util.peaked_state_dict(reference_init_state_dict(cfg), seed=2)
    model = TransModel(cfg, init=False)
    model.load_state_dict(sd)
    model.eval()
    return cfg, raw, ids, hb, sd, model, DeviceBatch(hb, cfg), Searcher(model)


def lines_of(hyps, raw, ids):
    r_vocab = {v: k for k, v in raw["word_vocab"].items()}
    return [text.detokenize(h, r_vocab, raw["variable"][i]) for h, i in zip(hyps, ids)]


def test_greedy_matches_reference_output_lines(setup):
    cfg, raw, ids, hb, sd, model, db, search = setup
    gold = json.load(open(os.path.join(util.GOLDEN, "decode_ref.json")))["beam1"]
    out, length, prob = search.greedy(db)                            # captures the step loop into hipGraphs
    assert lines_of(search.best(out, length, prob), raw, ids) == gold
    out_r, length_r, prob_r = search.greedy(db)                      # graph replay
    out_e, length_e, prob_e = search.greedy(db, use_graphs=False)    # eager launches
    assert torch.equal(out, out_r) and torch.equal(out, out_e) and torch.equal(length, length_e)
    assert torch.allclose(prob, prob_e, rtol=1e-5, atol=0) and torch.equal(prob, prob_r)
    out2, length2, prob2 = search.beam(db, 1)                       # the general path at beam 1 is the same search
    assert lines_of(search.best(out2, length2, prob2), raw, ids) == gold


def test_greedy_with_several_batches_in_flight_equals_batch_by_batch(setup):
    """Searcher.greedy_many: batches of the test set searched two / three at a time on separate streams (own workspace,
    hypothesis state and captured graphs per lane) -- the ids, lengths and probabilities must be exactly those of searching
    the batches one after the other (run_model.py:225 walks the test set sequentially; nothing couples two batches), for
    equal and unequal batch sizes, more batches than lanes, and a repeated call (graph replay on every lane)."""
    from fira_icse_amd.model import DeviceBatch
    from fira_icse_amd.decode import Searcher
    cfg, raw, ids, hb, sd, model, db, search = setup
    store = data.process_raw(cfg, raw)
    groups = [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10], [11, 12, 13, 14], [0, 1, 2, 3]]      # (any commits of the 24-commit store)
    dbs = [DeviceBatch(store.batch(g), cfg) for g in groups]
    one = Searcher(model)
    ref = [tuple(t.clone() for t in one.greedy(d)) for d in dbs]
    for in_flight in (2, 3):
        many = Searcher(model)
        for _ in range(2):
            got = many.greedy_many(dbs, in_flight=in_flight)
            torch.cuda.synchronize()
            for (o, l, p), (o2, l2, p2) in zip(ref, got):
                assert torch.equal(o, o2) and torch.equal(l, l2) and torch.equal(p, p2)
    gold = json.load(open(os.path.join(util.GOLDEN, "decode_ref.json")))["beam1"]
    got = search.greedy_many([db, db])
    assert lines_of(search.best(*got[1]), raw, ids) == gold


def test_search_with_a_bf16_cross_kv_copy_agrees_with_the_fp32_search(setup):
    """FIRA_DECODE_KV_BF16 (Searcher(kv_bf16=True)): the step loop streams a bf16 copy of the cross-attention K|V (half of
    the bytes a step moves).  Not bit-identical by construction -- the attention scores see bf16-rounded keys / values --
    but the search must stay the same search: on the peaked fixture weights every greedy and beam-3 hypothesis is the
    reference's line; in general the token agreement is reported by bench.py."""
    from fira_icse_amd.decode import Searcher
    cfg, raw, ids, hb, sd, model, db, search = setup
    gold = json.load(open(os.path.join(util.GOLDEN, "decode_ref.json")))
    s16 = Searcher(model, kv_bf16=True)
    out, length, prob = s16.greedy(db)
    out0, length0, prob0 = search.greedy(db)
    agree = float((out == out0).float().mean())
    assert agree > 0.95, agree
    assert torch.allclose(prob, prob0, rtol=5e-2)
    assert lines_of(s16.best(out, length, prob), raw, ids) == gold["beam1"]
    gen, length3, prob3 = s16.beam(db, 3)
    assert lines_of(s16.best(gen, length3, prob3), raw, ids) == gold["beam3"]


def test_beam3_matches_reference_output_lines(setup):
    cfg, raw, ids, hb, sd, model, db, search = setup
    gold = json.load(open(os.path.join(util.GOLDEN, "decode_ref.json")))["beam3"]
    gen, length, prob = search.beam(db, 3)
    assert lines_of(search.best(gen, length, prob), raw, ids) == gold


def test_beam3_all_hypotheses_and_probabilities_vs_oracle(setup):
    from oracle import fira_oracle as O
    cfg, raw, ids, hb, sd, model, db, search = setup
    tb = util.to_torch_batch(hb, cfg)
    hyp, prob = O.beam_decode(sd, cfg, tb["sou"], tb["mark"], tb["ast_change"], tb["edge"], tb["sub_token"], 3)
    gen, length, p = search.beam(db, 3)
    gen, length, p = gen.cpu(), length.cpu(), p.cpu()
    for i in range(len(hyp)):
        for j in range(3):
            assert gen[i, j, :length[i, j]].tolist() == hyp[i][j], (i, j)
            assert abs(float(p[i, j]) - prob[i][j]) <= 2e-4 * abs(prob[i][j]) + 1e-30, (i, j)


def test_step_distribution_equals_full_recompute(setup):
    """KV-cached step == the reference's full 30-position recompute at that position (run_model.py:256-267)."""
    import ctypes as C
    from fira_icse_amd import _lib
    from oracle import fira_oracle as O
    cfg, raw, ids, hb, sd, model, db, search = setup
    tb = util.to_torch_batch(hb, cfg)
    B, T, W = db.B, cfg.tar_len, cfg.out_len
    prefix = torch.tensor(np.ascontiguousarray(hb.tar[:, :6]))          # teacher tokens as a fixed prefix
    with torch.no_grad():
        memory, mem_mask = O.encode_memory(sd, cfg, tb["sou"], tb["mark"], tb["ast_change"], tb["edge"], tb["sub_token"])
        full = torch.zeros(B, T, dtype=torch.int64)
        full[:, :6] = prefix
        dec = O.decoder(sd, cfg, full, memory, mem_mask, full != 0)
        ref = O.output_distribution(sd, memory, mem_mask, dec)
    ws = search._begin(db, 1)
    dist = torch.empty((B, W), dtype=torch.float32, device="cuda")
    for step in range(6):
        search._step(ws, B, 1, step, prefix[:, step].to(torch.int32).cuda().contiguous(), None, dist, None, None)
        got = dist.cpu()
        err = float((got - ref[:, step]).abs().max() / ref[:, step].abs().max())
        assert err < 2e-4, (step, err)
        assert torch.equal(got.argmax(-1), ref[:, step].argmax(-1))
    mem = _lib.lib().fira_decode_memory(C.byref(model.dims), _lib.ptr(ws), B, 1)
    assert mem


@pytest.mark.parametrize("beam", [2, 3, 5])
def test_device_beam_bookkeeping_equals_torch_statement(setup, beam):
    """csrc/beam.hip (selection without a sort, state in HBM, hipGraph replay) against the same search written in
    torch ops: identical hypotheses, lengths and probabilities, eager == captured == replayed."""
    from fira_icse_amd.model import DeviceBatch
    cfg, raw, ids, hb, sd, model, db, search = setup
    store = data.process_raw(cfg, raw)
    for sel in (None, list(range(7))):                       # the fixture batch, and 7 other commits (odd batch)
        d = db if sel is None else DeviceBatch(store.batch(sel), cfg)
        gen_t, len_t, p_t = search_ref.beam_torch(search, d, beam)
        gen_e, len_e, p_e = search.beam(d, beam, use_graphs=False)
        gen_g, len_g, p_g = search.beam(d, beam)
        gen_r, len_r, p_r = search.beam(d, beam)
        for gen, length, p in ((gen_e, len_e, p_e), (gen_g, len_g, p_g), (gen_r, len_r, p_r)):
            assert torch.equal(length, len_t)
            assert torch.equal(p, p_t)
            T = gen.shape[-1]
            live = torch.arange(T, device=gen.device)[None, None, :] < length[:, :, None]
            assert torch.equal(gen * live, gen_t * live)
