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
    @torch.no_grad()
    def greedy(self, db: DeviceBatch, sync_every: int = 4) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Returns (tokens [B,T] int64 starting with <start>, lengths [B], probability [B])."""
        cfg, dev = self.cfg, self.model.device_
        B, T = db.B, cfg.tar_len
        ws = self._begin(db, 1)
        sou, sub = db.sou.long(), db.sub_token.long()
        out = torch.zeros((B, T), dtype=torch.int64, device=dev)
        out[:, 0] = START
        length = torch.ones(B, dtype=torch.int64, device=dev)
        prob = torch.ones(B, dtype=torch.float32, device=dev)
        alive = torch.ones(B, dtype=torch.bool, device=dev)
        tok = torch.full((B,), START, dtype=torch.int32, device=dev)
        best_id = torch.empty(B, dtype=torch.int32, device=dev)
        best_p = torch.empty(B, dtype=torch.float32, device=dev)
        for step in range(T - 1):
            self._step(ws, B, 1, step, tok, None, None, best_id, best_p)
            nxt = self._resolve(best_id.long()[:, None], sou, sub)[:, 0]
            out[:, step + 1] = torch.where(alive, nxt, out[:, step + 1])
            prob = torch.where(alive, prob * best_p, prob)
            length = length + alive.long()
            alive = alive & (nxt != EOS)
            tok = torch.where(alive, nxt, torch.zeros_like(nxt)).to(torch.int32)
            if (step + 1) % sync_every == 0 and not bool(alive.any()):      # run_model.py:276-279
                break
        return out, length, prob

    # ------------------------------------------------------------------ beam search with the reference's semantics
    @torch.no_grad()
    def beam(self, db: DeviceBatch, beam: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Returns (hypotheses [B,beam,T] int64, lengths [B,beam], probabilities [B,beam])."""
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
