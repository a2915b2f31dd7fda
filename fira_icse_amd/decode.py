"""Test-time search of the reference (run_model.py:202-340) on the HIP engine.

The reference re-runs the whole 6-layer decoder, the 24 650-way generator and the copy head over all 30 positions for
every beam at every step, and moves hypotheses through python lists with a host<->device round trip per step and beam.
Here the encoder, the cross-attention K/V of all layers and ``LinearSource(memory)`` are computed once per batch
(``fira_decode_begin``); every step is one KV-cached pass over the (commit, beam) rows (``fira_decode_step``), and the
hypothesis bookkeeping (probability products, -1 for finished rows, carried finished beams, descending sort, copy-id
resolution) stays on the device with the reference's exact semantics.  ``beam = 1`` is the reference's "greedy".
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import torch

from . import _lib
from .config import EOS, PAD, START
from .model import DeviceBatch, TransModel


class Searcher:
    def __init__(self, model: TransModel):
        self.model = model
        self.cfg = model.cfg
        self._ws = {}

    def _workspace(self, B, beam):
        key = (B, beam)
        if key not in self._ws:
            n = _lib.lib().fira_decode_workspace_bytes(C.byref(self.model.dims), B, beam)
            if n == 0:
                _lib.check(1, "fira_decode_workspace_bytes")
            self._ws[key] = torch.empty(n, dtype=torch.uint8, device=self.model.device_)
        return self._ws[key]

    def _begin(self, db, beam):
        ws = self._workspace(db.B, beam)
        _lib.check(_lib.lib().fira_decode_begin(_lib.cur_stream(), C.byref(self.model.dims), C.byref(db.struct),
                                                _lib.ptr(self.model.flat.data), _lib.ptr(ws), ws.numel(), beam),
                   "fira_decode_begin")
        return ws

    def _step(self, ws, B, beam, step, tokens, parent, dist, best_id, best_p):
        _lib.check(_lib.lib().fira_decode_step(_lib.cur_stream(), C.byref(self.model.dims),
                                               _lib.ptr(self.model.flat.data), _lib.ptr(ws), ws.numel(), B, beam, step,
                                               _lib.ptr(tokens), _lib.ptr(parent), _lib.ptr(dist), _lib.ptr(best_id),
                                               _lib.ptr(best_p)), "fira_decode_step")

    def _resolve(self, idx, sou, sub):
        """output index -> vocabulary id (run_model.py:334-338); idx [B,k] int64, sou [B,L], sub [B,S]."""
        V, L = self.cfg.vocab_size, self.cfg.sou_len
        from_sou = torch.gather(sou, 1, (idx - V).clamp(0, sou.shape[1] - 1))
        from_sub = torch.gather(sub, 1, (idx - V - L).clamp(0, sub.shape[1] - 1))
        return torch.where(idx >= V + L, from_sub, torch.where(idx >= V, from_sou, idx))

    # ------------------------------------------------------------------ greedy (beam 1): no sort, no dist tensor
    def _greedy_state(self, B):
        """Static device buffers + captured hipGraphs of the step loop for batch size B (built on first use)."""
        key = ("greedy", B)
        if key in self._ws:
            return self._ws[key]
        cfg, dev = self.cfg, self.model.device_
        T = cfg.tar_len
        st = dict(
            sou=torch.zeros((B, cfg.sou_len), dtype=torch.int64, device=dev),
            sub=torch.zeros((B, cfg.sub_token_len), dtype=torch.int64, device=dev),
            out=torch.zeros((B, T), dtype=torch.int64, device=dev),
            length=torch.ones(B, dtype=torch.int64, device=dev),
            prob=torch.ones(B, dtype=torch.float32, device=dev),
            alive=torch.ones(B, dtype=torch.bool, device=dev),
            tok=torch.full((B,), START, dtype=torch.int32, device=dev),
            best_id=torch.empty(B, dtype=torch.int32, device=dev),
            best_p=torch.empty(B, dtype=torch.float32, device=dev),
            graphs=None)
        self._ws[key] = st
        return st

    def _greedy_steps(self, st, ws, B, lo, hi):
        """Steps lo..hi-1 of run_model.py:225-340 at beam 1, entirely on the device (no host round trip)."""
        for step in range(lo, hi):
            self._step(ws, B, 1, step, st["tok"], None, None, st["best_id"], st["best_p"])
            nxt = self._resolve(st["best_id"].long()[:, None], st["sou"], st["sub"])[:, 0]
            alive = st["alive"]
            st["out"][:, step + 1] = torch.where(alive, nxt, st["out"][:, step + 1])
            st["prob"].copy_(torch.where(alive, st["prob"] * st["best_p"], st["prob"]))
            st["length"].add_(alive.long())
            st["alive"].copy_(alive & (nxt != EOS))
            st["tok"].copy_(torch.where(st["alive"], nxt, torch.zeros_like(nxt)).to(torch.int32))

    @torch.no_grad()
    def greedy(self, db: DeviceBatch, chunk: int = 5, use_graphs: bool = True):
        """Returns (tokens [B,T] int64 starting with <start>, lengths [B], probability [B]).

        The step loop is launch-bound (~85 tiny kernels per generated token), so it is captured once per batch size
        into hipGraphs of ``chunk`` steps each and replayed; between chunks one flag is read back to stop as soon as
        every hypothesis has emitted <eos> (run_model.py:276-279)."""
        cfg = self.cfg
        B, T = db.B, cfg.tar_len
        ws = self._begin(db, 1)
        st = self._greedy_state(B)
        st["sou"].copy_(db.sou)
        st["sub"].copy_(db.sub_token)
        st["out"].zero_()
        st["out"][:, 0] = START
        st["length"].fill_(1)
        st["prob"].fill_(1.0)
        st["alive"].fill_(True)
        st["tok"].fill_(START)
        bounds = [(lo, min(lo + chunk, T - 1)) for lo in range(0, T - 1, chunk)]
        if use_graphs and st["graphs"] is None:
            # warm-up outside capture (lazy initialisation inside the library / torch), then capture every chunk
            snap = {k: v.clone() for k, v in st.items() if isinstance(v, torch.Tensor)}
            self._greedy_steps(st, ws, B, 0, 1)
            torch.cuda.synchronize()
            for k, v in snap.items():
                st[k].copy_(v)
            graphs = []
            for lo, hi in bounds:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._greedy_steps(st, ws, B, lo, hi)
                graphs.append(g)
            st["graphs"] = graphs
            for k, v in snap.items():           # capture does not execute: restore nothing but be explicit
                st[k].copy_(v)
        for i, (lo, hi) in enumerate(bounds):
            if use_graphs:
                st["graphs"][i].replay()
            else:
                self._greedy_steps(st, ws, B, lo, hi)
            if hi < T - 1 and not bool(st["alive"].any()):
                break
        return st["out"].clone(), st["length"].clone(), st["prob"].clone()

    # ------------------------------------------------------------------ beam search with the reference's semantics
    def _beam_state(self, B, beam):
        key = ("beam", B, beam)
        if key in self._ws:
            return self._ws[key]
        cfg, dev = self.cfg, self.model.device_
        BR, T = B * beam, cfg.tar_len
        i32 = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=dev)
        st = dict(gen=[i32(BR, T), i32(BR, T)], length=[i32(BR), i32(BR)],
                  prob=[torch.zeros(BR, dtype=torch.float32, device=dev) for _ in range(2)],
                  tok=i32(BR), parent=i32(BR), fin=i32(BR), active=i32(9), done=i32(1),
                  dist=torch.empty((BR, cfg.out_len), dtype=torch.float32, device=dev),
                  sou=i32(B, cfg.sou_len), sub=i32(B, cfg.sub_token_len), graphs=None)
        self._ws[key] = st
        return st

    def _beam_reset(self, st, B, beam):
        for k in ("gen", "length", "prob"):
            st[k][0].zero_()
            st[k][1].zero_()
        st["gen"][0][:, 0] = START
        st["length"][0].fill_(1)
        st["prob"][0].view(B, beam)[:, 0] = 1.0
        st["done"].zero_()

    def _beam_steps(self, st, ws, B, beam, lo, hi):
        """Steps lo..hi-1 of run_model.py:225-340: prepare -> KV-cached decoder step -> select, all on the device."""
        lib, s, T = _lib.lib(), _lib.cur_stream(), self.cfg.tar_len
        for step in range(lo, hi):
            cur, nxt = step & 1, (step + 1) & 1
            _lib.check(lib.fira_beam_prepare(s, B, beam, T, step, _lib.ptr(st["gen"][cur]), _lib.ptr(st["length"][cur]),
                                             _lib.ptr(st["tok"]), _lib.ptr(st["fin"]), _lib.ptr(st["active"]),
                                             _lib.ptr(st["done"])), "fira_beam_prepare")
            self._step(ws, B, beam, step, st["tok"], st["parent"] if step > 0 else None, st["dist"], None, None)
            _lib.check(lib.fira_beam_select(s, C.byref(self.model.dims), B, beam, _lib.ptr(st["dist"]),
                                            _lib.ptr(st["fin"]), _lib.ptr(st["active"]), _lib.ptr(st["done"]),
                                            _lib.ptr(st["sou"]), _lib.ptr(st["sub"]), _lib.ptr(st["gen"][cur]),
                                            _lib.ptr(st["length"][cur]), _lib.ptr(st["prob"][cur]),
                                            _lib.ptr(st["gen"][nxt]), _lib.ptr(st["length"][nxt]),
                                            _lib.ptr(st["prob"][nxt]), _lib.ptr(st["parent"])), "fira_beam_select")

    @torch.no_grad()
    def beam(self, db: DeviceBatch, beam: int, chunk: int = 4, use_graphs: bool = True):
        """Returns (hypotheses [B,beam,T] int64, lengths [B,beam], probabilities [B,beam]).

        Per step: fira_beam_prepare, fira_decode_step, fira_beam_select (csrc/beam.hip) -- three library calls, no torch
        op and no host round trip; the loop is captured into hipGraphs of ``chunk`` steps per (batch, beam) shape, and
        the ``done`` latch is read back between chunks (run_model.py:276-279)."""
        cfg = self.cfg
        B, T = db.B, cfg.tar_len
        ws = self._begin(db, beam)
        st = self._beam_state(B, beam)
        st["sou"].copy_(db.sou)
        st["sub"].copy_(db.sub_token)
        self._beam_reset(st, B, beam)
        bounds = [(lo, min(lo + chunk, T - 1)) for lo in range(0, T - 1, chunk)]
        if use_graphs and st["graphs"] is None:
            self._beam_steps(st, ws, B, beam, 0, 1)            # warm-up outside capture
            torch.cuda.synchronize()
            graphs = []
            for lo, hi in bounds:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._beam_steps(st, ws, B, beam, lo, hi)
                graphs.append(g)
            st["graphs"] = graphs
            self._beam_reset(st, B, beam)
        last = 0
        for i, (lo, hi) in enumerate(bounds):
            if use_graphs:
                st["graphs"][i].replay()
            else:
                self._beam_steps(st, ws, B, beam, lo, hi)
            last = hi
            if hi < T - 1 and bool(st["done"].item()):
                break
        cur = last & 1
        return (st["gen"][cur].view(B, beam, T).long(), st["length"][cur].view(B, beam).long(),
                st["prob"][cur].view(B, beam).clone())

    @torch.no_grad()
    def beam_torch(self, db: DeviceBatch, beam: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """The same search with the bookkeeping written in torch ops (descending sort of all candidates, one host sync
        per step): kept as an independent statement of run_model.py:268-340 that the tests hold ``beam`` against."""
        cfg, dev = self.cfg, self.model.device_
        B, T, W = db.B, cfg.tar_len, cfg.out_len
        BR = B * beam
        ws = self._begin(db, beam)
        sou, sub = db.sou.long(), db.sub_token.long()
        gen = torch.zeros((B, beam, T), dtype=torch.int64, device=dev)
        gen[:, :, 0] = START
        length = torch.ones((B, beam), dtype=torch.int64, device=dev)
        prob = torch.zeros((B, beam), dtype=torch.float32, device=dev)
        prob[:, 0] = 1.0
        dist = torch.empty((BR, W), dtype=torch.float32, device=dev)
        parent = None
        slot = torch.arange(beam, device=dev)
        rowbase = (torch.arange(B, device=dev) * beam)[:, None]
        bidx = torch.arange(B, device=dev)[:, None]
        for step in range(T - 1):
            last = torch.gather(gen, 2, (length - 1)[:, :, None])[:, :, 0]
            finished = last == EOS                                             # [B,beam]
            active = (~finished).any(0)                                        # slot j runs iff some item is unfinished
            active_slots = active.nonzero().view(-1)
            n_act = int(active_slots.numel())                                  # the step's only host sync
            if n_act == 0:
                break
            tok = torch.where(length > step, gen[:, :, step], torch.zeros_like(last)).to(torch.int32).reshape(-1)
            self._step(ws, B, beam, step, tok.contiguous(), parent, dist, None, None)
            cand = dist.view(B, beam, W) * prob[:, :, None]
            cand = torch.where(finished[:, :, None], torch.full_like(cand, -1.0), cand)
            blocks = cand[:, active_slots, :].reshape(B, n_act * W)
            # finished hypotheses of the item in slot order, padded with -1 (run_model.py:283-296)
            order = torch.argsort(torch.where(finished, slot[None, :], slot[None, :] + beam), dim=1)
            n_fin = finished.sum(1, keepdim=True)
            carried = torch.where(slot[None, :] < n_fin, torch.gather(prob, 1, order), torch.full_like(prob, -1.0))
            allv = torch.cat([blocks, carried], 1)
            top_p, top_i = torch.sort(allv, descending=True, dim=-1)
            top_p, top_i = top_p[:, :beam], top_i[:, :beam]
            which, tokidx = top_i // W, top_i % W
            carry = which == n_act
            src_slot = torch.where(carry, torch.gather(order, 1, tokidx.clamp(max=beam - 1)),
                                   active_slots[which.clamp(max=n_act - 1)])
            new_tok = self._resolve(tokidx.clamp(max=W - 1), sou, sub)
            src_len = torch.gather(length, 1, src_slot)
            gen = torch.gather(gen, 1, src_slot[:, :, None].expand(B, beam, T)).clone()
            pos = src_len.clamp(max=T - 1)
            appended = gen.scatter(2, pos[:, :, None], new_tok[:, :, None])
            gen = torch.where(carry[:, :, None], gen, appended)
            length = torch.where(carry, src_len, src_len + 1)
            prob = top_p.contiguous()
            parent = (rowbase + src_slot).to(torch.int32).reshape(-1).contiguous()
        return gen, length, prob

    def best(self, gen, length, prob) -> List[List[int]]:
        """argmax-probability hypothesis per item, first on ties (run_model.py:351-352)."""
        if gen.dim() == 2:
            return [row[:n] for row, n in zip(gen.tolist(), length.tolist())]
        j = torch.argmax(prob, dim=1)        # first maximal index, like np.argmax
        g = gen[torch.arange(gen.shape[0], device=gen.device), j].tolist()
        n = length[torch.arange(gen.shape[0], device=gen.device), j].tolist()
        return [row[:k] for row, k in zip(g, n)]
