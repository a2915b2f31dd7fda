"""Row-sparse Adam of the two vocabulary-sized embedding tables (fira_train_step_rows, include/fira_hip.h v10) against the
dense update it replaces (fira_adam_step_mb = torch.optim.Adam of run_model.py:396 on every row every step).

The claim is bit-exactness: a row whose gradient row is zero is updated later, in registers, by the same instruction sequence
the dense kernel runs on it -- so after a sync the tables and both moments are EQUAL (torch.equal), not close.  The pieces are
compared in isolation (the training step around them has split-K atomic sums and is not bit-reproducible run to run), then the
trainer built on them is compared with the dense trainer at the tolerance the one-call test uses."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup():
    from fira_icse_amd import _lib
    from fira_icse_amd.config import FiraConfig
    from fira_icse_amd.model import TransModel
    cfg = FiraConfig()
    model = TransModel(cfg, device="cuda")
    return _lib, cfg, model


def _tables(model, cfg):
    v = model.named_views()
    base = model.flat.data.data_ptr()
    out = []
    for name in ("decoder.embedding.weight", "encoder.embedding.weight"):
        off = (v[name].data_ptr() - base) // 4
        out.append((off, off + cfg.vocab_size * 256))
    return out


@pytest.mark.parametrize("touch_frac", [0.02, 0.3])
def test_rows_update_equals_dense_update_bit_for_bit(touch_frac):
    _lib, cfg, model = _setup()
    lib = _lib.lib()
    dims = model.dims
    V = cfg.vocab_size
    dev = model.flat.data.device
    gen = torch.Generator(device="cuda").manual_seed(5)
    tabs = _tables(model, cfg)
    lo = min(t[0] for t in tabs)
    total = model.flat.data.numel()
    p0 = torch.randn(total, device=dev, generator=gen) * 0.05
    pd, md, vd = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)          # dense side
    pr, mr, vr = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)          # row-sparse side
    last = torch.zeros(2 * V, dtype=torch.int32, device=dev)
    g = torch.zeros(total, device=dev)
    n_tok = torch.tensor([37], dtype=torch.int32, device=dev)
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    s = _lib.cur_stream()

    def opts(step, m, v):
        return _lib.AdamOpts(lr, b1, b2, eps, step, _lib.ptr(m), _lib.ptr(v))

    def equal_on_tables():
        for a, b in tabs:
            assert torch.equal(pd[a:b], pr[a:b]) and torch.equal(md[a:b], mr[a:b]) and torch.equal(vd[a:b], vr[a:b])

    T = 75                                                       # crosses the every-row steps 32 and 64
    for step in range(1, T + 1):
        g.zero_()
        touched = []
        for t, (a, b) in enumerate(tabs):
            rows = torch.nonzero(torch.rand(V, device=dev, generator=gen) < touch_frac).flatten()
            vals = torch.randn(rows.numel(), 256, device=dev, generator=gen)
            if rows.numel() > 3:
                vals[0].zero_()                                  # a touched row whose gradient happens to be zero
                vals[1, 5:].zero_()                              # ... and one with a few non-zero elements only
            g[a:b].view(V, 256)[rows] = vals
            touched.append(rows.to(torch.int32))
        # dense: every row of both tables
        for a, b in tabs:
            _lib.check(lib.fira_adam_step_mb(s, b - a, _lib.ptr(pd[a:]), _lib.ptr(g[a:]), None, _lib.ptr(md[a:]), _lib.ptr(vd[a:]),
                                             lr, b1, b2, eps, step, _lib.ptr(n_tok), None))
        # row-sparse: the rows the "forward pass" would gather (duplicates, a few untouched ones, an out-of-range id) are
        # brought up to step - 1 first -- and must then hold what the dense side held BEFORE its step
        for t in (0, 1):
            extra = torch.randint(0, V, (64,), device=dev, generator=gen, dtype=torch.int32)
            ids = torch.cat([touched[t], touched[t][:17], extra, torch.tensor([V + 3, -1], dtype=torch.int32, device=dev)])
            if step > 1:
                ad = opts(step - 1, mr, vr)
                _lib.check(lib.fira_adam_rows_catchup(s, C.byref(dims), _lib.ptr(pr), C.byref(ad), _lib.ptr(last), t, _lib.ptr(ids),
                                                      ids.numel()))
        ad = opts(step, mr, vr)
        _lib.check(lib.fira_adam_rows_step(s, C.byref(dims), _lib.ptr(pr), _lib.ptr(g), C.byref(ad), _lib.ptr(last), _lib.ptr(n_tok), None, 3))
        for t, (a, b) in enumerate(tabs):                        # the touched rows are current after the step
            rows = touched[t].long()
            nz = (g[a:b].view(V, 256)[rows] != 0).any(1)
            assert torch.equal(pd[a:b].view(V, 256)[rows[nz]], pr[a:b].view(V, 256)[rows[nz]]), (step, t)
        if step in (20, 32, 47, T):                              # a sync mid-window, on an every-row step, at the end
            ad = opts(step, mr, vr)
            _lib.check(lib.fira_adam_rows_sync(s, C.byref(dims), _lib.ptr(pr), C.byref(ad), _lib.ptr(last)))
            equal_on_tables()
            assert int(last.min()) == step == int(last.max())
    # nothing outside the two tables was written
    mask = torch.ones(total, dtype=torch.bool, device=dev)
    for a, b in tabs:
        mask[a:b] = False
    assert torch.equal(pr[mask], p0[mask]) and not mr[mask].any() and not vr[mask].any()
    assert lo >= 0


def test_rows_update_leaves_untouched_rows_alone_between_syncs():
    """The point of the exercise: between the every-row steps, a step reads the gradient rows and moves (p, m, v) of the
    touched rows only."""
    _lib, cfg, model = _setup()
    lib = _lib.lib()
    V = cfg.vocab_size
    dev = model.flat.data.device
    tabs = _tables(model, cfg)
    total = model.flat.data.numel()
    p = torch.randn(total, device=dev) * 0.05
    m = torch.rand(total, device=dev) * 1e-3
    v = torch.rand(total, device=dev) * 1e-6
    last = torch.full((2 * V,), 4, dtype=torch.int32, device=dev)
    g = torch.zeros(total, device=dev)
    a, b = tabs[1]
    g[a:b].view(V, 256)[[7, 9000]] = 1.0
    n_tok = torch.tensor([5], dtype=torch.int32, device=dev)
    p0, m0 = p.clone(), m.clone()
    ad = _lib.AdamOpts(1e-3, 0.9, 0.999, 1e-8, 5, _lib.ptr(m), _lib.ptr(v))
    _lib.check(lib.fira_adam_rows_step(_lib.cur_stream(), C.byref(model.dims), _lib.ptr(p), _lib.ptr(g), C.byref(ad), _lib.ptr(last),
                                       _lib.ptr(n_tok), None, 3))
    changed = (p != p0).view(-1)
    rows = torch.nonzero(changed[a:b].view(V, 256).any(1)).flatten().tolist()
    assert rows == [7, 9000]
    assert not changed[:a].any() and not changed[b:].any() and torch.equal(m[:a], m0[:a])
    assert last[V + 7].item() == 5 and last[V + 9000].item() == 5 and last[V + 8].item() == 4


def test_rows_trainer_equals_dense_trainer():
    """Trainer on fira_train_step_rows against Trainer on fira_train_step (FIRA_ADAM_ROWS=0): 40 steps over four batches with
    dropout on (same masks), crossing an every-row step, then a sync.  The step is not bit-reproducible run to run (split-K
    atomic sums), and 40 steps amplify that: the yardstick is the distance between TWO dense runs."""
    from fira_icse_amd import data, synth
    from fira_icse_amd.config import FiraConfig
    from fira_icse_amd.model import DeviceBatch, TransModel
    from fira_icse_amd.train import Trainer
    cfg = FiraConfig()
    store = data.process_raw(cfg, synth.generate_dataset(32, seed=11))
    out = {}
    for run, rows in (("rows", "1"), ("dense", "0"), ("dense2", "0")):
        os.environ["FIRA_ADAM_ROWS"] = rows
        try:
            torch.manual_seed(0)
            model = TransModel(cfg, device="cuda")
            model.train()
            model.set_dropout_stream(3, 0)
            tr = Trainer(model)
        finally:
            os.environ.pop("FIRA_ADAM_ROWS", None)
        assert (tr.row_step is not None) == (rows == "1")
        batches = [DeviceBatch(store.batch(list(range(8 * i, 8 * i + 8))), cfg, model.device_) for i in range(4)]
        losses = []
        for i in range(40):
            tr.step(batches[i % 4])
            if i % 10 == 9:
                losses.append(tr.last_loss())
        if rows == "1":                                          # rows no batch touched lag behind until the sync
            assert int(tr.row_step.min()) < 40
        sd = model.state_dict()                                  # (syncs)
        if rows == "1":
            assert int(tr.row_step.min()) == 40
        out[run] = (model.flat.data.clone(), tr.m.clone(), tr.v.clone(), sd["encoder.embedding.weight"].clone(),
                    sd["decoder.embedding.weight"].clone(), losses)
    live = model.layout.live

    def dist(x, y):
        return float((x - y).norm() / y.norm())
    for k, name in enumerate(("params", "m", "v", "encoder.embedding", "decoder.embedding")):
        a, b, b2 = (out[r][k] for r in ("rows", "dense", "dense2"))
        if k < 3:
            a, b, b2 = a[:live], b[:live], b2[:live]
        noise = dist(b2, b)
        # (floors: two runs of the SAME trainer can also differ by a ReLU tie taken the other way -- a pre-activation of +-1e-9
        #  whose mask flips with the atomics' rounding noise: the loss does not move, one row of a weight gradient does; seen as
        #  |dv| / |v| = 7e-5 against a quiet pair's 8e-6, |dm| / |m| up to 1e-3: scripts/probes/nondet_probe.py.  A row the
        #  sparse update left behind is caught by the table check below and, bit for bit, by the torch.equal tests above.)
        floor = {"params": 1e-4, "m": 1e-2, "v": 1e-3}.get(name, 1e-4)
        assert dist(a, b) < 6 * noise + floor, (name, dist(a, b), noise)
    # a row left behind would differ by whole updates, step after step: every row of the tables moved as the dense update moves
    # it.  (Floor: a ReLU tie taken the other way changes gradients everywhere, embedding rows included, and Adam turns that into
    # differences of a fraction of lr per element -- 6.3e-5 seen once in the full suite against a quiet pair's 3e-7; a row the
    # sparse update forgot lags by ~lr per step it was owed.  The bit-for-bit guarantees are the torch.equal tests above, which
    # feed both kernels the same gradients.)
    for k in (3, 4):
        a, b, b2 = (out[r][k] for r in ("rows", "dense", "dense2"))
        worst = ((a - b).abs().amax(1) / (b2 - b).abs().amax(1).clamp_min(1e-7)).max()
        assert float((a - b).abs().max()) < 6 * float((b2 - b).abs().max()) + 2 * cfg.lr, (k, float(worst))
    assert np.allclose(out["rows"][5], out["dense"][5], rtol=5e-3), (out["rows"][5], out["dense"][5])
