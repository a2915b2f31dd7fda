"""The C-ABI library loads and exports every symbol include/fira_hip.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import util  # noqa: F401
from fira_icse_amd import _lib
from fira_icse_amd.config import FiraConfig


def header_symbols():
    txt = open(os.path.join(util.REPO, "include", "fira_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fira_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "libfira_hip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "no ctypes signature for %s" % s
    assert sorted(_lib.SIGNATURES) == syms


def test_abi_version_and_error_string(lib):
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fira_hip.h")).read()
    assert lib.fira_abi_version() == int(re.search(r"#define FIRA_ABI_VERSION (\d+)", header).group(1)) == 10
    bad = _lib.make_dims(FiraConfig(embedding_dim=128))
    assert lib.fira_param_count(C.byref(bad)) == -1
    assert b"d_model=256" in lib.fira_last_error()


def test_param_layout_matches_reference_checkpoint_keys(lib):
    from fira_icse_amd.model import ParamLayout
    cfg = FiraConfig()
    lay = ParamLayout(cfg)
    assert len(lay.entries) == 338
    total = sum(int(__import__("numpy").prod(s)) for _, s in lay.entries.values())
    assert total == 30963534                            # SURVEY.md §6 "Model size"
    # no two tensors overlap, everything 256-byte aligned
    spans = sorted((off, off + int(__import__("numpy").prod(s))) for off, s in lay.entries.values())
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0
    assert all(off % 64 == 0 for off, _ in lay.entries.values())
    # the two gradient-readiness buckets are multiples of 64 elements: they split exactly over 2 / 4 / 8 ranks (ZeRO-1
    # reduce-scatter / all-gather run in place, parallel.ShardedOptimizerComm)
    assert lay.split % 64 == 0 and lay.live % 64 == 0 and 0 < lay.split < lay.live <= lay.total
    assert lay.entries["out_fc.weight"][1] == (24650, 256)
    assert lay.entries["copy_net.LinearProb.bias"][1] == (2,)
    # fused groups are contiguous: q|k of a Combination, q|k|v of self-attention, k|v of all cross-attentions
    e = lay.entries
    assert e["encoder.combination_list2.3.linear_layers.1.weight"][0] == e["encoder.combination_list2.3.linear_layers.0.weight"][0] + 65536
    assert e["decoder.attention_list.2.fc_v.weight"][0] == e["decoder.attention_list.2.fc_q.weight"][0] + 2 * 65536
    assert e["decoder.cross_attention_list.1.fc_k.weight"][0] == e["decoder.cross_attention_list.0.fc_k.weight"][0] + 2 * 65536
    assert lib.fira_workspace_bytes(C.byref(lay.dims), 4, 1) > lib.fira_workspace_bytes(C.byref(lay.dims), 4, 0) > 0
    assert lib.fira_decode_workspace_bytes(C.byref(lay.dims), 4, 3) > 0
