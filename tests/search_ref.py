"""Test-only statement of the reference's beam search (run_model.py:268-340) in torch ops on top of the engine's
per-step distribution: descending sort over all candidates, one host sync per step.  ``tests/test_decode_gpu.py`` holds
the device-side bookkeeping kernels (``fira_beam_prepare`` / ``fira_beam_select``) against it.  Not part of the product
package."""
from typing import Tuple

import torch

from fira_icse_amd.config import EOS, START


def _resolve(cfg, idx, sou, sub):
    """output index -> vocabulary id (run_model.py:334-338); idx [B,k] int64, sou [B,L], sub [B,S]."""
    V, L = cfg.vocab_size, cfg.sou_len
    from_sou = torch.gather(sou, 1, (idx - V).clamp(0, sou.shape[1] - 1))
    from_sub = torch.gather(sub, 1, (idx - V - L).clamp(0, sub.shape[1] - 1))
    return torch.where(idx >= V + L, from_sub, torch.where(idx >= V, from_sou, idx))


@torch.no_grad()
def beam_torch(search, db, beam: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """The same search with the bookkeeping written in torch ops (descending sort of all candidates, one host sync
    per step): kept as an independent statement of run_model.py:268-340 that the tests hold ``beam`` against."""
    cfg, dev = search.cfg, search.model.device_
    B, T, W = db.B, cfg.tar_len, cfg.out_len
    BR = B * beam
    ws = search._begin(db, beam)
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
        search._step(ws, B, beam, step, tok.contiguous(), parent, dist, None, None)
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
        new_tok = _resolve(cfg, tokidx.clamp(max=W - 1), sou, sub)
        src_len = torch.gather(length, 1, src_slot)
        gen = torch.gather(gen, 1, src_slot[:, :, None].expand(B, beam, T)).clone()
        pos = src_len.clamp(max=T - 1)
        appended = gen.scatter(2, pos[:, :, None], new_tok[:, :, None])
        gen = torch.where(carry[:, :, None], gen, appended)
        length = torch.where(carry, src_len, src_len + 1)
        prob = top_p.contiguous()
        parent = (rowbase + src_slot).to(torch.int32).reshape(-1).contiguous()
    return gen, length, prob

