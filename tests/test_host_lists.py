"""fira_host_node_lists (csrc/hostlists.cpp, host-only C++) against its specification, the numpy functions of model.py
(computed_nodes + compact_embedding_lists): every list bit-identical on synthetic batches of several sizes, with and
without padding skipping, on the hand-made edge commits, and through DeviceBatch (CPU arena)."""
import os

import numpy as np
import pytest

import util
from fira_icse_amd import data, synth
from fira_icse_amd.config import FiraConfig
from fira_icse_amd import model as M

NAMES = ("node_rows", "rowptr", "col", "val", "code_rows", "code_mark", "mem_rows", "mem_dst", "item_tok", "item_ptr",
         "emb_rows", "ast_rows", "ast_ids")


def both(hb, cfg, skip):
    os.environ["FIRA_HOST_LISTS"] = "numpy"
    try:
        ref = M.batch_lists(hb, cfg, skip)
    finally:
        os.environ.pop("FIRA_HOST_LISTS")
    return ref, M.batch_lists(hb, cfg, skip)


def check(hb, cfg, skip):
    ref, got = both(hb, cfg, skip)
    assert len(ref) == len(got) == len(NAMES)
    for name, a, b in zip(NAMES, ref, got):
        a, b = np.asarray(a), np.asarray(b)
        assert a.dtype == b.dtype and a.shape == b.shape, (name, a.dtype, b.dtype, a.shape, b.shape)
        assert np.array_equal(a, b), name


@pytest.fixture(scope="module")
def store():
    cfg = FiraConfig()
    return cfg, data.process_raw(cfg, synth.generate_dataset(96, seed=77, overlong_every=5))


@pytest.mark.parametrize("skip", [True, False])
@pytest.mark.parametrize("ids", [[0], [5, 5, 5], list(range(32)), list(range(95, 31, -1)), [3, 90, 17, 44, 2, 81, 60]])
def test_native_lists_equal_the_numpy_specification(store, ids, skip):
    cfg, st = store
    check(st.batch(ids), cfg, skip)


def test_golden_and_edge_commits():
    cfg = FiraConfig()
    st = data.process_raw(cfg, util.load_golden_raw())
    for ids in (list(range(len(st))), [0, 1], [len(st) - 1]):
        check(st.batch(ids), cfg, True)
        check(st.batch(ids), cfg, False)


def test_a_batch_without_any_token_or_edge():
    """All ids zero, self-loops only: no computed node, no item -- both paths return the same empty lists."""
    cfg, N = FiraConfig(), FiraConfig().graph_len
    B = 2
    z = lambda n: np.zeros((B, n), dtype=np.int64)
    rowptr = np.arange(B * N + 1, dtype=np.int32)
    hb = data.HostBatch(z(cfg.sou_len), z(cfg.tar_len), z(cfg.sou_len), z(N - cfg.sou_len - cfg.sub_token_len), z(cfg.tar_len),
                        z(cfg.sub_token_len), rowptr, np.arange(B * N, dtype=np.int32), np.ones(B * N, dtype=np.float32))
    check(hb, cfg, True)
    check(hb, cfg, False)


def test_device_batch_is_the_same_through_both_paths(store):
    cfg, st = store
    hb = st.batch(list(range(16)))
    os.environ["FIRA_HOST_LISTS"] = "numpy"
    try:
        a = M.DeviceBatch(hb, cfg, device="cpu")
    finally:
        os.environ.pop("FIRA_HOST_LISTS")
    b = M.DeviceBatch(hb, cfg, device="cpu")
    assert a.arena.numel() == b.arena.numel()               # (the 256-byte alignment gaps of the arena are uninitialised)
    for k in ("sou", "tar", "mark", "ast_change", "tar_label", "sub_token", "node_rows", "rowptr", "col", "val", "code_rows",
              "code_mark", "mem_rows", "mem_dst", "head_rows", "ast_rows", "ast_ids", "emb_item_tok", "emb_item_ptr", "emb_rows"):
        x, y = getattr(a, k), getattr(b, k)
        assert x.dtype == y.dtype and x.shape == y.shape and bool((x == y).all()), k
    for k in ("n_nodes", "n_code", "n_mem", "nnz", "n_head_rows", "n_ast_items", "n_emb_items"):
        assert getattr(a, k) == getattr(b, k), k


@pytest.mark.parametrize("ids", [[0], [7, 7], list(range(40)), [95, 3, 50, 50, 1]])
def test_native_collate_equals_the_numpy_statement(store, ids):
    """GraphStore.batch through fira_host_collate_csr == its numpy statement: every array of the HostBatch, the lazily
    gathered attr included."""
    cfg, st = store
    os.environ["FIRA_HOST_LISTS"] = "numpy"
    try:
        ref = st.batch(ids)
    finally:
        os.environ.pop("FIRA_HOST_LISTS")
    got = st.batch(ids)
    for k in ("sou", "tar", "mark", "ast_change", "tar_label", "sub_token", "rowptr", "col", "val", "attr"):
        a, b = getattr(ref, k), getattr(got, k)
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), k
    assert np.array_equal(ref.dense_edge(cfg.graph_len), got.dense_edge(cfg.graph_len))


def test_native_collate_rejects_an_index_outside_the_store(store):
    from fira_icse_amd._lib import FiraError
    cfg, st = store
    with pytest.raises((FiraError, IndexError)):
        st.batch([0, len(st)])


def test_computed_target_rows_are_the_prefix_that_covers_keys_and_loss_rows(store):
    """fira_batch.dec_off (DeviceBatch): per commit the shortest prefix of the 30 target positions that contains every
    non-zero id (attention keys) and every position whose shifted label is non-zero (loss rows); at least one row."""
    cfg, st = store
    hb = st.batch(range(40))
    hb.tar[3, :] = 0                                  # a commit without any target token keeps one row
    hb.tar_label[3, :] = 0
    hb.tar[5, 20] = 7                                 # a hole: position 20 used although 12..19 may be padding
    db = M.DeviceBatch(hb, cfg, device="cpu")
    off = db.dec_off.numpy()
    T = cfg.tar_len
    assert off[0] == 0 and off[-1] == db.n_dec_rows == len(db.dec_rows_host)
    tar, lab = np.asarray(hb.tar), np.asarray(hb.tar_label)
    for b in range(len(hb)):
        n = int(off[b + 1] - off[b])
        used = [t for t in range(T) if tar[b, t] != 0 or (t + 1 < T and lab[b, t + 1] != 0)]
        assert n == (max(used) + 1 if used else 1), (b, n, used)
        assert list(db.dec_rows_host[off[b]:off[b + 1]]) == [b * T + t for t in range(n)]
    assert off[4] - off[3] == 1 and off[6] - off[5] >= 21
    assert db.n_dec_rows < len(hb) * T                # the synthetic messages are shorter than 30 tokens
