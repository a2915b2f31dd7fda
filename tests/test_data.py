"""Host data layer against the fixtures the reference's Dataset.py produced (tests/golden/dataset_ref.npz)."""
import os

import numpy as np
import pytest

import util
from fira_icse_amd import data, synth
from fira_icse_amd.config import FiraConfig


@pytest.fixture(scope="module")
def store():
    return data.process_raw(FiraConfig(), util.load_golden_raw())


def test_split_matches_reference_all_index():
    g = util.golden_npz("dataset_ref.npz")
    idx = data.split_index(*util.GOLDEN_SPLIT, seed=0)
    for name in ("train", "valid", "test"):
        assert idx[name] == g["%s_index" % name].tolist()


@pytest.mark.parametrize("split", ["train", "valid", "test"])
def test_arrays_bit_exact_vs_reference(store, split):
    g = util.golden_npz("dataset_ref.npz")
    idx = g["%s_index" % split]
    for key in ("sou", "tar", "attr", "mark", "ast_change", "tar_label", "sub_token"):
        assert np.array_equal(getattr(store, key)[idx], g["%s_%s" % (split, key)]), key


@pytest.mark.parametrize("split", ["train", "valid", "test"])
def test_adjacency_bit_exact_vs_reference(store, split):
    g = util.golden_npz("dataset_ref.npz")
    N = FiraConfig().graph_len
    off = np.concatenate([[0], np.cumsum(g["%s_edge_nnz" % split])])
    for k, i in enumerate(g["%s_index" % split]):
        ref = np.zeros((N, N))
        sl = slice(off[k], off[k + 1])
        ref[g["%s_edge_row" % split][sl], g["%s_edge_col" % split][sl]] = g["%s_edge_val" % split][sl]
        assert np.array_equal(store.dense_edge(int(i)), ref)        # float64, exact
        assert store.nnz[i] == off[k + 1] - off[k]


def test_batch_csr_is_block_diagonal_and_symmetric(store):
    cfg = FiraConfig()
    hb = store.batch([3, 0, 7])
    N = cfg.graph_len
    assert hb.rowptr.shape == (3 * N + 1,) and hb.rowptr[-1] == hb.col.shape[0]
    rows = np.repeat(np.arange(3 * N), np.diff(hb.rowptr))
    assert np.all(rows // N == hb.col // N)                          # no edge leaves its graph
    dense = hb.dense_edge(N)
    assert np.array_equal(dense, np.transpose(dense, (0, 2, 1)))     # A_hat symmetric: backward == forward SpMM
    assert np.array_equal(dense[1], store.dense_edge(0).astype(np.float32).astype(np.float64))


def test_overlong_commit_keeps_reference_truncation(store):
    raw = util.load_golden_raw()
    long_ids = [i for i, d in enumerate(raw["difftoken"]) if len(d) > 208]
    assert long_ids, "golden set must contain an over-long commit"
    i = long_ids[0]
    assert store.sou[i][-1] != 0 and 1 not in store.sou[i].tolist()[1:]   # truncated: no <eos> left, no padding
    # sequential edges run past the code range unguarded: node 210 (first sub-token) links to 209
    assert store.dense_edge(i)[209, 210] > 0


def test_reference_asserts_are_kept():
    cfg = FiraConfig()
    raw = synth.generate_dataset(2, seed=1)
    raw["edge_ast"][0].append([3, 3])                                  # self pair (Dataset.py:275)
    with pytest.raises(AssertionError):
        data.process_raw(cfg, raw)
    raw = synth.generate_dataset(2, seed=1)
    raw["diffatt"][0][0] = ["Upper"]                                   # Dataset.py:150
    with pytest.raises(AssertionError):
        data.process_raw(cfg, raw)
    raw = synth.generate_dataset(2, seed=1)
    raw["ast"][0][0] = "not_in_vocab"                                  # no <unkm> in ast vocab: KeyError as reference
    with pytest.raises(KeyError):
        data.process_raw(cfg, raw)


def test_dataset_class_roundtrip(tmp_path):
    cfg = FiraConfig()
    synth.write_dataset(str(tmp_path), util.load_golden_raw())
    ds = data.TransDataset(cfg, "train", root=str(tmp_path), splits=util.GOLDEN_SPLIT)
    g = util.golden_npz("dataset_ref.npz")
    assert len(ds) == util.GOLDEN_SPLIT[0]
    item = ds[2]
    assert np.array_equal(item[0], g["train_sou"][2]) and item[5].dtype == np.float64 and item[5].shape == (650, 650)
    ds2 = data.TransDataset(cfg, "test", root=str(tmp_path))           # served from the cache
    assert np.array_equal(ds2.store.tar_label, g["test_tar_label"])


def test_computed_node_lists(store):
    """Host side of the padded-node skipping: lists are consistent, the compact CSR is the dense one restricted to
    computed rows, and the over-long commit's unguarded edges keep their (id 0) endpoints computed."""
    from fira_icse_amd.model import computed_nodes
    cfg = FiraConfig()
    N, L, S = cfg.graph_len, cfg.sou_len, cfg.sub_token_len
    raw = util.load_golden_raw()
    long_i = [i for i, d in enumerate(raw["difftoken"]) if len(d) > 208][0]
    hb = store.batch([0, long_i, 5])
    node_rows, rowptr, col, val, code_rows, code_mark, mem_rows, mem_dst = computed_nodes(hb, cfg)
    dn, dr, dc, dv, dcr, dcm, dmr, dmd = computed_nodes(hb, cfg, skip_padding=False)
    assert dn.shape[0] == 3 * N and np.array_equal(dr, hb.rowptr) and np.array_equal(dc, hb.col)
    assert node_rows.shape[0] < 3 * N and np.all(np.diff(node_rows) > 0)
    dense = hb.dense_edge(N).reshape(3 * N, N)
    for k in (0, 7, len(node_rows) // 2, len(node_rows) - 1):
        g = node_rows[k]
        cols = node_rows[col[rowptr[k]:rowptr[k + 1]]]
        assert np.array_equal(np.sort(cols % N), np.nonzero(dense[g])[0]) and np.all(cols // N == g // N)
        assert np.allclose(val[rowptr[k]:rowptr[k + 1]], dense[g][cols % N].astype(np.float32))
    skipped = np.setdiff1d(np.arange(3 * N), node_rows)
    ids = np.concatenate([hb.sou, hb.sub_token, hb.ast_change], axis=1).reshape(-1)
    assert np.all(ids[skipped] == 0) and np.all(np.diff(hb.rowptr)[skipped] == 1)     # only self-loop padding is skipped
    assert np.array_equal(node_rows[code_rows] % N < L, np.ones(len(code_rows), bool))
    assert np.array_equal(code_mark, hb.mark.reshape(-1)[(node_rows[code_rows] // N) * L + node_rows[code_rows] % N])
    assert np.array_equal(mem_dst, (node_rows[mem_rows] // N) * (L + S) + node_rows[mem_rows] % N)
    # every unmasked memory slot (id != 0) is computed
    mem_ids = np.concatenate([hb.sou, hb.sub_token], axis=1).reshape(-1)
    assert np.all(np.isin(np.nonzero(mem_ids != 0)[0], mem_dst))
    # the over-long commit: node 210 (sub-token id 0 or not) has the unguarded sequential edge -> computed
    assert (1 * N + 210) in node_rows


def test_embedding_items_cover_every_nonpad_position_once():
    """fira_batch.emb_*: every code / sub-token position with a non-zero id appears in exactly one item of its id,
    items hold at most 32 positions, and a scatter through the items equals the plain per-position scatter."""
    from fira_icse_amd.model import embedding_items
    cfg = FiraConfig()
    store = data.process_raw(cfg, util.edge_case_raw())
    hb = store.batch([0, 1, 2, 3])
    tok, ptr, rows = embedding_items(hb, cfg)
    assert ptr[0] == 0 and ptr[-1] == rows.shape[0] and np.all(np.diff(ptr) >= 1) and np.all(np.diff(ptr) <= 32)
    assert np.all(tok != 0)
    ids = np.concatenate([hb.sou, hb.sub_token], axis=1)
    B, N, W = ids.shape[0], cfg.graph_len, ids.shape[1]
    flat = {int(b * N + i): int(ids[b, i]) for b in range(B) for i in range(W) if ids[b, i] != 0}
    assert sorted(rows.tolist()) == sorted(flat)
    for k in range(tok.shape[0]):
        assert all(flat[int(r)] == int(tok[k]) for r in rows[ptr[k]:ptr[k + 1]])
    # an item boundary never separates two ids in the wrong order: ids ascend over the items
    assert np.all(np.diff(tok.astype(np.int64)) >= 0)


@pytest.mark.parametrize("skip_padding", [True, False])
def test_compact_embedding_lists_point_at_the_right_nodes(skip_padding):
    """Word items and AST (row, id) pairs index the compact node list: node_rows[row] is a node that carries that id,
    every id-carrying node is listed exactly once, padding (id 0) never."""
    from fira_icse_amd.model import compact_embedding_lists, computed_nodes
    cfg = FiraConfig()
    hb = data.process_raw(cfg, util.edge_case_raw()).batch([0, 1, 2, 3])
    node_rows = computed_nodes(hb, cfg, skip_padding)[0]
    tok, ptr, rows, ast_rows, ast_ids = compact_embedding_lists(hb, cfg, node_rows)
    N, L, S = cfg.graph_len, cfg.sou_len, cfg.sub_token_len
    words = np.concatenate([hb.sou, hb.sub_token], axis=1)
    for k in range(tok.shape[0]):
        g = node_rows[rows[ptr[k]:ptr[k + 1]]]
        assert np.all(g % N < L + S) and np.all(words[g // N, g % N] == tok[k])
    assert rows.shape[0] == int((words != 0).sum()) == np.unique(rows).shape[0]
    g = node_rows[ast_rows]
    assert np.all(g % N >= L + S) and np.array_equal(hb.ast_change[g // N, g % N - L - S], ast_ids)
    assert ast_rows.shape[0] == int((hb.ast_change != 0).sum()) == np.unique(ast_rows).shape[0] and np.all(ast_ids != 0)


def test_cache_is_rebuilt_when_its_inputs_change_and_followers_wait_for_it(tmp_path):
    """the split cache is keyed by what it depends on (splits, seed, lengths) and ranks other than 0 wait
    for it through the file system instead of a collective with a timeout."""
    import json as _json
    import threading
    import time
    cfg = FiraConfig()
    root = str(tmp_path)
    synth.write_dataset(root, util.load_golden_raw())
    got = {}

    def follower():                                   # a rank != 0: never builds, polls for a fresh cache
        got["ds"] = data.TransDataset(cfg, "train", root=root, splits=util.GOLDEN_SPLIT, seed=0, build=False, wait_s=60)

    th = threading.Thread(target=follower)
    th.start()
    time.sleep(0.3)
    assert th.is_alive()                              # nothing to read yet
    a = data.TransDataset(cfg, "train", root=root, splits=util.GOLDEN_SPLIT, seed=0)
    th.join(30)
    assert not th.is_alive() and np.array_equal(got["ds"].store.sou, a.store.sou)
    idx0 = _json.load(open(os.path.join(root, "all_index")))
    b = data.TransDataset(cfg, "train", root=root, splits=util.GOLDEN_SPLIT, seed=5)     # other seed: other split
    idx5 = _json.load(open(os.path.join(root, "all_index")))
    assert idx0 != idx5 and not np.array_equal(a.store.sou, b.store.sou)
    c = data.TransDataset(cfg, "train", root=root, splits=(12, 6, 6), seed=5)            # other sizes: rebuilt again
    assert len(c) == 12
