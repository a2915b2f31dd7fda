"""Edge cases of the hot path against the CPU oracle: single commit, a commit without sub-tokens / AST / edits, a
message made only of copied tokens (no row needs the vocabulary head), a maximum-length message, an over-long diff."""
import numpy as np
import pytest
import torch

import util
from fira_icse_amd import data, synth
from fira_icse_amd.config import FiraConfig

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model_sd():
    from fira_icse_amd.model import TransModel, reference_init_state_dict
    cfg = FiraConfig()
    torch.manual_seed(0)
    sd = util.perturb_state_dict(reference_init_state_dict(cfg), seed=1)
    model = TransModel(cfg, init=False)
    model.load_state_dict(sd)
    model.eval()
    return cfg, model, sd


@pytest.mark.parametrize("sel", [[0], [1], [2], [3], [0, 1, 2, 3]])
def test_loss_grad_and_ids_vs_oracle(model_sd, sel):
    from oracle import fira_oracle as O
    from fira_icse_amd.model import DeviceBatch
    cfg, model, sd = model_sd
    store = data.process_raw(cfg, util.edge_case_raw())
    hb = store.batch(sel)
    db = DeviceBatch(hb, cfg)
    if sel == [1]:
        assert db.n_head_rows <= 1                      # only the <eos> row can still need the generator
    tb = util.to_torch_batch(hb, cfg)
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ls, nt = O.forward(P, cfg, tb["sou"], tb["tar"], tb["mark"], tb["ast_change"], tb["edge"], tb["tar_label"],
                       tb["sub_token"], "train")
    ls.backward()
    loss, ntok = model.train_fwd_bwd(db)
    assert int(ntok) == int(nt)
    assert abs(float(loss) - float(ls.detach())) <= 1e-5 * float(ls.detach())
    gv = model.grad_views()
    num = sum(float((gv[k].cpu().double() - p.grad.double()).norm() ** 2) for k, p in P.items() if p.grad is not None)
    den = sum(float(p.grad.double().norm() ** 2) for p in P.values() if p.grad is not None)
    assert (num / den) ** 0.5 < 1e-4
    with torch.no_grad():
        ids = O.forward(sd, cfg, tb["sou"], tb["tar"], tb["mark"], tb["ast_change"], tb["edge"], tb["tar_label"],
                        tb["sub_token"], "dev")
    real = tb["tar"] != 0                                # positions past the message are never read by dev()
    assert torch.equal(model.forward_dev(db).cpu().long()[real], ids[real])


def test_greedy_early_stop_and_single_commit(model_sd):
    """A batch whose every hypothesis ends immediately stops the chunked graph replay early; batch of one works."""
    from fira_icse_amd.model import DeviceBatch, TransModel
    from fira_icse_amd.decode import Searcher
    cfg, model, sd = model_sd
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["out_fc.bias"][1] += 1e4                         # <eos> wins every step
    sd2["copy_net.LinearProb.bias"][0] += 50.0           # generator branch only
    m2 = TransModel(cfg, init=False)
    m2.load_state_dict(sd2)
    m2.eval()
    store = data.process_raw(cfg, util.edge_case_raw())
    for sel in ([2], [0, 1, 2, 3]):
        out, length, prob = Searcher(m2).greedy(DeviceBatch(store.batch(sel), cfg))
        assert length.tolist() == [2] * len(sel) and out[:, 1].tolist() == [1] * len(sel)
        assert float(prob.min()) > 0.99


def test_dense_graphs_take_the_unfused_gcn_path_and_match_the_oracle():
    """Graphs far denser than FIRA's (BASELINE config 5's regime: here ~70 entries per computed row from thousands of random
    AST-AST and AST-code edges) make the engine run the GCN layers as aggregation + product + row kernel instead of the fused
    launch (engine.hip: note_graph_density).  Loss and every gradient tensor against the oracle on the dense float64
    adjacency, as for the sparse fixtures."""
    import numpy as np
    from oracle import fira_oracle as O
    from fira_icse_amd import data, synth
    from fira_icse_amd.config import FiraConfig
    from fira_icse_amd.model import TransModel, DeviceBatch, reference_init_state_dict
    cfg = FiraConfig()
    raw = synth.generate_dataset(3, seed=31)
    rng = np.random.default_rng(5)
    for i in range(3):
        n_a, n_d = len(raw["ast"][i]), len(raw["difftoken"][i])
        pairs = rng.integers(0, n_a, size=(9000, 2))
        raw["edge_ast"][i] = raw["edge_ast"][i] + [[int(a), int(b)] for a, b in pairs if a != b]
        ac = np.stack([rng.integers(0, n_a, 3000), rng.integers(0, n_d, 3000)], 1)
        raw["edge_ast_code"][i] = raw["edge_ast_code"][i] + [[int(a), int(b)] for a, b in ac]
    store = data.process_raw(cfg, raw)
    hb = store.batch([0, 1, 2])
    db = DeviceBatch(hb, cfg)
    assert db.nnz > 48 * db.n_nodes, (db.nnz, db.n_nodes)          # beyond the fused kernels' density limit
    torch.manual_seed(0)
    sd = util.perturb_state_dict(reference_init_state_dict(cfg), seed=1)
    model = TransModel(cfg, init=False)
    model.load_state_dict(sd)
    model.eval()
    tb = util.to_torch_batch(hb, cfg)
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ls, nt = O.forward(P, cfg, tb["sou"], tb["tar"], tb["mark"], tb["ast_change"], tb["edge"], tb["tar_label"],
                       tb["sub_token"], "train")
    ls.backward()
    loss, ntok = model.train_fwd_bwd(db)
    assert int(ntok) == int(nt)
    assert abs(float(loss) - float(ls.detach())) / float(ls.detach()) < 1e-5
    gv = model.grad_views()
    floor = 1e-6 * max(float(p.grad.double().norm()) for p in P.values() if p.grad is not None)
    for k, p in P.items():
        if p.grad is None:
            continue
        ref = p.grad.double()
        err = float((gv[k].cpu().double() - ref).norm())
        tol = 3e-4 if ("feed_forward_list" in k and ".fc1." in k) else 1e-4
        assert err <= tol * float(ref.norm()) + floor, (k, err, float(ref.norm()))
    ids = model.forward_dev(db).cpu()
    with torch.no_grad():
        want = O.forward(sd, cfg, tb["sou"], tb["tar"], tb["mark"], tb["ast_change"], tb["edge"], tb["tar_label"],
                         tb["sub_token"], "dev")
    assert torch.equal(ids.long(), want.long())
