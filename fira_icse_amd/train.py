"""Native training step (reference run_model.py:101-112) on the HIP engine: no autograd, no per-tensor optimizer.

    fwd+bwd (one C call, whole step enqueued on the stream)  ->  [RCCL all-reduce]  ->  fused Adam over the flat buffer

The 1/n_tok normaliser stays on the device (no ``loss.item()`` sync per step; the reference syncs every step at
run_model.py:112); ``last_loss()`` reads it back only when asked.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib, ops
from .model import DeviceBatch, TransModel
from .parallel import GradReducer, ShardedOptimizerComm


class Trainer:
    def __init__(self, model: TransModel, lr: Optional[float] = None, betas=(0.9, 0.999), eps: float = 1e-8,
                 distributed: bool = False, zero1: bool = False, grad_wire: str = "f32"):
        """``zero1`` (with ``distributed``): reduce-scatter + Adam on the owned 1/world shard + all-gather instead of
        all-reduce + replicated Adam; Adam moments exist only for the owned shard (parallel.ShardedOptimizerComm).
        ``grad_wire`` (with ``distributed``, all-reduce path): "f32", or "bf16" = the two gradient buckets travel as bf16
        (half the bytes on xGMI; parallel.GradReducer)."""
        self.model = model
        self.lr = model.cfg.lr if lr is None else lr
        self.betas, self.eps = betas, eps
        self.zero = ShardedOptimizerComm(model.layout.split, model.layout.live, model.layout.total) \
            if (distributed and zero1) else None
        if self.zero is None:
            self.m = torch.zeros_like(model.gbuf)
            self.v = torch.zeros_like(model.gbuf)
        else:
            dev = model.gbuf.device
            chunks = [q["chunk"] for q in self.zero.buckets]
            self.m_sh = [torch.zeros(c, dtype=torch.float32, device=dev) for c in chunks]
            self.v_sh = [torch.zeros(c, dtype=torch.float32, device=dev) for c in chunks]
            self.g_sh = [torch.zeros(c, dtype=torch.float32, device=dev) for c in chunks]
            self.zstream = torch.cuda.Stream()
            self.end_event = torch.cuda.Event()
        # one library call per step (fira_train_step: the head + decoder slice of the update beside the last weight gradients);
        # FIRA_FUSED_STEP=0 = fira_train_fwd_bwd + one Adam launch over [0, live) (A/B switch; identical results)
        fused_on = os.environ.get("FIRA_FUSED_STEP", "1") != "0"
        self.fused_step = (not distributed) and fused_on and _lib.has_symbol("fira_train_step")
        # data parallel (round 6): the same schedule as two calls with the collectives in between (fira_train_step_begin / _end:
        # Adam of [0, split) inside the library, beside its last weight gradients, behind the early bucket's event)
        self.fused_dp = distributed and not zero1 and fused_on and _lib.has_symbol("fira_train_step_begin")
        self.t = 0
        # row-sparse Adam of the two vocabulary-sized embedding tables (fira_train_step_rows, round 6): the rows a batch did
        # not touch are updated lazily, bit for bit (include/fira_hip.h); FIRA_ADAM_ROWS=0 = every row every step (A/B switch).
        # The model calls self.sync before anything but this trainer's step reads the parameters.
        self.row_step = None
        self._rows_dirty = False
        self._rows_hyper = None
        if (self.fused_step or self.fused_dp) and os.environ.get("FIRA_ADAM_ROWS", "1") != "0" \
                and _lib.has_symbol("fira_train_step_end_rows") and model.cfg.embedding_dim == 256:
            model.sync_params()                                  # (an earlier trainer of this model may still owe rows)
            self.row_step = torch.zeros(2 * model.cfg.vocab_size, dtype=torch.int32, device=model.gbuf.device)
            model._rows_sync = self.sync
        self.inv = torch.zeros(1, dtype=torch.float32, device=model.gbuf.device)
        self.stats = torch.zeros(2, dtype=torch.float32, device=model.gbuf.device)
        self.reducer = GradReducer(model.layout.split, model.layout.live, wire=grad_wire) if distributed else None
        self.mid_event = None
        if distributed:
            # fires inside the backward pass when the gradients of [0, split) (head + decoder) are final: their all-reduce
            # then runs beside the encoder's backward pass
            self.mid_event = torch.cuda.Event()
            self.mid_event.record()            # torch creates the hipEvent lazily: force it so its handle can be passed

    def step(self, db: Optional[DeviceBatch]):
        """One optimisation step on this rank's shard of the global batch.

        ``db is None`` = this rank's shard of the global batch is empty (``shard_range`` chunks like
        ``DataParallel.scatter``: the tail batch of an epoch can leave trailing ranks without commits).  Such a rank
        still joins every collective of the step with a zero gradient and zero (loss_sum, n_tok) and applies the same
        Adam update as the others, so the replicas stay identical and nobody waits for a peer that never arrives."""
        m = self.model
        dp = self.reducer is not None and self.reducer.world > 1
        fused_dp = False
        if db is None:
            if not dp and not (self.zero is not None and self.zero.world > 1):
                return                                           # nothing to learn from, nobody to keep in step
            m.gbuf[:m.layout.live].zero_()
            m.loss_sum.zero_()
            m.n_tok.zero_()
            self.mid_event.record()
            loss_sum, n_tok = m.loss_sum, m.n_tok
        elif self.fused_step and self.reducer is None and self.zero is None:
            self._rows_check_hyper()
            self.t += 1
            m.train_step(db, self.m, self.v, self.lr, self.t, self.betas[0], self.betas[1], self.eps, row_step=self.row_step)
            self._rows_dirty = self.row_step is not None
            return
        elif dp and self.fused_dp and self.zero is None:
            fused_dp = True
            self._rows_check_hyper()
            rows = None if self.row_step is None else \
                (self.m, self.v, self.lr, self.t + 1, self.betas[0], self.betas[1], self.eps, self.row_step)
            loss_sum, n_tok = m.train_step_begin(db, self.mid_event, rows=rows)
        else:
            loss_sum, n_tok = m.train_fwd_bwd(db, zero_grad=True, mid_event=self.mid_event)
        b1, b2 = self.betas
        if self.zero is not None and self.zero.world > 1:
            self._step_zero1(loss_sum, n_tok)
            return
        if dp:
            red, split, live = self.reducer, m.layout.split, m.layout.live
            # stats pair + head/decoder bucket on the communication stream behind the mid-backward event (beside the encoder's
            # backward pass).  One launch packs {loss_sum, float(n_tok)} (exact below 2^24 tokens); the update kernels form
            # 1 / max(count, 1) from the all-reduced pair themselves: no torch arithmetic between the collective and Adam
            ev = red.reduce_early(m.gbuf, self.stats, self.mid_event, pack=lambda: ops.pack_stats(loss_sum, n_tok, self.stats))
            self.t += 1
            count = self.stats[1:2]
            if fused_dp:
                # encoder backward; Adam of [0, split) inside the library as soon as the caller's stream has passed the
                # encoder's chain and `ev`, beside the last weight gradients; the join
                m.train_step_end(self.m, self.v, self.lr, self.t, early_event=ev, count=count, beta1=b1, beta2=b2, eps=self.eps,
                                 row_step=self.row_step)
            else:
                red.wait_early()
                self._adam_slice(0, split, count, table=0)
            red.start_late(m.gbuf)
            red.wait_late(m.gbuf)
            self._adam_slice(split, live, count, table=1)
            self._rows_dirty = self.row_step is not None
            return
        self.t += 1
        # [live, total) holds the tensors no kernel touches (encoder.lstm, combination_list1, gate_fc): their gradient is
        # None in the reference, so torch.optim.Adam skips them too.  1 / n_tok (run_model.py:105) is formed inside the
        # Adam kernel from the device counter (no separate launch, no host sync).
        # One launch over [0, live).  (Running the head+decoder slice on its own stream beside the encoder's backward pass
        # was measured on one box: 8 497 vs 8 553 commits/s -- the HBM-bound update only slows the backward kernels it
        # overlaps; profiles/r2_probes.md.)
        n = m.layout.live
        ops.adam_step_mb(m.flat.data[:n], m.gbuf[:n], None, self.m[:n], self.v[:n], self.lr, self.t, n_tok, None, b1, b2,
                         self.eps)

    def _step_zero1(self, loss_sum, n_tok):
        """reduce-scatter -> Adam on the owned shard -> all-gather, per readiness bucket, on a side stream: the head+decoder
        bucket's reduce-scatter starts at the mid-backward event, beside the encoder's backward pass."""
        m, z, zs = self.model, self.zero, self.zstream
        b1, b2 = self.betas
        main = torch.cuda.current_stream()
        zs.wait_event(self.mid_event)
        with torch.cuda.stream(zs):
            z.reduce_scatter(0, m.gbuf, self.g_sh[0])
        ops.pack_stats(loss_sum, n_tok, self.stats)           # {loss_sum, float(n_tok)}: exact below 2^24 tokens
        torch.distributed.all_reduce(self.stats, op=torch.distributed.ReduceOp.SUM, group=z.group)
        self.end_event.record(main)                            # backward pass done, global token count known
        self.t += 1
        zs.wait_event(self.end_event)
        with torch.cuda.stream(zs):
            for b in (0, 1):
                if b == 1:
                    z.reduce_scatter(1, m.gbuf, self.g_sh[1])
                lo, hi = z.owned(b)
                if hi > lo:
                    n = hi - lo
                    ops.adam_step_count(m.flat.data[lo:hi], self.g_sh[b][:n], self.m_sh[b][:n], self.v_sh[b][:n], self.lr,
                                        self.t, self.stats[1:2], b1, b2, self.eps)
                z.all_gather(b, m.flat.data)
        main.wait_stream(zs)                                   # the next forward pass reads every parameter

    def _rows_check_hyper(self):
        hyper = (self.lr, self.betas[0], self.betas[1], self.eps)
        if self._rows_hyper != hyper:                            # lazily applied updates use the step's own lr / beta / eps
            self.sync()
            self._rows_hyper = hyper

    def _adam_slice(self, lo: int, hi: int, count, table: int):
        """Adam on ``[lo, hi)`` of the flat buffers, scaled by ``1 / count`` (data-parallel step).  On the row-sparse path the
        vocabulary-sized table at the head of the slice (``table`` 0: decoder.embedding at 0, 1: encoder.embedding at
        ``split``) is updated on the rows of the all-reduced gradient that are not zero (fira_adam_rows_step)."""
        m = self.model
        b1, b2 = self.betas
        if self.row_step is not None:
            import ctypes as C
            self._rows_check_hyper()
            adam = _lib.AdamOpts(self.lr, b1, b2, self.eps, int(self.t), _lib.ptr(self.m), _lib.ptr(self.v))
            _lib.check(_lib.lib().fira_adam_rows_step(_lib.cur_stream(), C.byref(m.dims), _lib.ptr(m.flat.data), _lib.ptr(m.gbuf),
                                                      C.byref(adam), _lib.ptr(self.row_step), None, _lib.ptr(count), 1 << table),
                       "fira_adam_rows_step")
            lo += m.cfg.vocab_size * 256
        ops.adam_step_count(m.flat.data[lo:hi], m.gbuf[lo:hi], self.m[lo:hi], self.v[lo:hi], self.lr, self.t, count, b1, b2,
                            self.eps)

    def sync(self):
        """Apply the embedding-row updates the row-sparse path still owes (fira_adam_rows_sync): afterwards ``model.flat``,
        ``self.m`` and ``self.v`` are what the dense update leaves after ``self.t`` steps, bit for bit.  Cheap when nothing
        is owed.  Synchronises the stream (a phase change -- checkpoint, dev pass, search -- not a per-step call)."""
        if self.row_step is None or not self._rows_dirty:
            return
        self._rows_dirty = False
        lr, b1, b2, eps = self._rows_hyper
        import ctypes as C
        adam = _lib.AdamOpts(lr, b1, b2, eps, int(self.t), _lib.ptr(self.m), _lib.ptr(self.v))
        _lib.check(_lib.lib().fira_adam_rows_sync(_lib.cur_stream(), C.byref(self.model.dims), _lib.ptr(self.model.flat.data),
                                                  C.byref(adam), _lib.ptr(self.row_step)), "fira_adam_rows_sync")
        torch.cuda.current_stream().synchronize()

    def last_loss(self) -> float:
        """Mean token loss of the last (global) batch; synchronises."""
        if self.reducer is not None and self.reducer.world > 1:
            s = self.stats.tolist()
        else:
            s = [float(self.model.loss_sum.item()), float(self.model.n_tok.item())]
        return s[0] / max(s[1], 1.0)

    def state_dict(self):
        """Everything a restart needs besides the weights: Adam moments, Adam step and the dropout step counter (so
        that a resumed run continues the mask sequence instead of replaying it from step 1)."""
        if self.zero is not None:                              # collective: every rank calls it, any rank may save it
            total = self.model.layout.total
            return {"m": self.zero.gather_full(self.m_sh, total), "v": self.zero.gather_full(self.v_sh, total),
                    "t": self.t, "dropout_step": self.model.dropout_step}
        self.sync()
        return {"m": self.m, "v": self.v, "t": self.t, "dropout_step": self.model.dropout_step}

    def load_state_dict(self, sd):
        if self.zero is not None:                              # the checkpoint holds full moments: keep the owned shards
            for b in (0, 1):
                lo, hi = self.zero.owned(b)
                self.m_sh[b][:hi - lo].copy_(sd["m"][lo:hi]); self.v_sh[b][:hi - lo].copy_(sd["v"][lo:hi])
        else:
            self.sync()
            self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
        self.t = int(sd["t"])
        if self.row_step is not None:
            self.row_step.fill_(self.t)                          # a checkpoint holds synced tables
        self.model.dropout_step = int(sd.get("dropout_step", self.t))
