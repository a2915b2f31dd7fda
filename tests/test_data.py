"""Host data layer against the fixtures the reference's Dataset.py produced (tests/golden/dataset_ref.npz)."""
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
