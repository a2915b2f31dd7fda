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
from typing import List

import torch

from . import _lib
from .config import EOS, PAD, START
from .model import DeviceBatch, TransModel


def concurrent_streams(n: int, device, pool: int = 12):
    """``n`` HIP streams that really run side by side.  HIP multiplexes streams onto a few hardware queues (4 by default,
    assigned round-robin in creation order over everything the process has created so far): two streams that land on the same
    queue serialise, and which ones do depends on the process's history.  So: create ``pool`` candidate streams, time a short
    spin kernel on pairs of them, and keep a set whose members overlap with each other (measured: 4 lanes on 4 distinct queues
    run 4 batches of 64 in 17 ms, on 2 queues in 32 ms -- DESIGN.md section 6).  Costs a few milliseconds, once per Searcher."""
    import time
    cands = [torch.cuda.Stream(device=device) for _ in range(max(pool, n))]
    if n <= 1:
        return cands[:1]
    spin = 400_000                                         # cycles of torch.cuda._sleep: ~0.2 ms

    def run(streams):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for st in streams:
            with torch.cuda.stream(st):
                torch.cuda._sleep(spin)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    run(cands[:2])                                         # warm-up
    one = min(run(cands[:1]) for _ in range(3))
    chosen = [cands[0]]
    for c in cands[1:]:
        if len(chosen) == n:
            break
        if all(min(run([c, o]) for _ in range(2)) < 1.5 * one for o in chosen):
            chosen.append(c)
    for c in cands:                                        # fewer distinct queues than lanes: fill up with the rest
        if len(chosen) == n:
            break
        if c not in chosen:
            chosen.append(c)
    return chosen


class Searcher:
    def __init__(self, model: TransModel, kv_bf16: bool = False):
        """``kv_bf16``: stream a bf16 copy of the cross-attention K|V in the step loop (FIRA_DECODE_KV_BF16: half of the
        bytes a step moves; ids no longer bit-identical to the fp32 search -- off by default)."""
        self.model = model
        self.cfg = model.cfg
        self.flags = 1 if kv_bf16 else 0
        self._ws = {}

    def _workspace(self, B, beam):
        key = (B, beam)
        if key not in self._ws:
            n = _lib.lib().fira_decode_workspace_bytes_ex(C.byref(self.model.dims), B, beam, self.flags)
            if n == 0:
                _lib.check(1, "fira_decode_workspace_bytes")
            self._ws[key] = torch.empty(n, dtype=torch.uint8, device=self.model.device_)
        return self._ws[key]

    def _begin(self, db, beam):
        ws = self._workspace(db.B, beam)
        self.model.sync_params()
        db.wait_ready()
        _lib.check(_lib.lib().fira_decode_begin_ex(_lib.cur_stream(), C.byref(self.model.dims), C.byref(db.struct),
                                                   _lib.ptr(self.model.flat.data), _lib.ptr(ws), ws.numel(), beam,
                                                   self.flags), "fira_decode_begin")
        return ws

    def _step(self, ws, B, beam, step, tokens, parent, dist, best_id, best_p):
        _lib.check(_lib.lib().fira_decode_step_ex(_lib.cur_stream(), C.byref(self.model.dims),
                                                  _lib.ptr(self.model.flat.data), _lib.ptr(ws), ws.numel(), B, beam, step,
                                                  _lib.ptr(tokens), _lib.ptr(parent), _lib.ptr(dist), _lib.ptr(best_id),
                                                  _lib.ptr(best_p), self.flags), "fira_decode_step")

    # ------------------------------------------------------------------ greedy (beam 1): no sort, no dist tensor
    def _greedy_state(self, B):
        """Static device buffers + captured hipGraphs of the step loop for batch size B (built on first use)."""
        key = ("greedy", B)
        if key in self._ws:
            return self._ws[key]
        cfg, dev = self.cfg, self.model.device_
        T = cfg.tar_len
        i32 = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=dev)
        st = dict(sou=i32(B, cfg.sou_len), sub=i32(B, cfg.sub_token_len), out=i32(B, T), length=i32(B),
                  prob=torch.ones(B, dtype=torch.float32, device=dev), alive=i32(B), tok=i32(B), n_alive=i32(T),
                  best_id=i32(B), best_p=torch.empty(B, dtype=torch.float32, device=dev), graphs=None)
        self._ws[key] = st
        return st

    def _greedy_reset(self, st):
        st["out"].zero_()
        st["out"][:, 0] = START
        st["length"].fill_(1)
        st["prob"].fill_(1.0)
        st["alive"].fill_(1)
        st["tok"].fill_(START)
        st["n_alive"].zero_()

    def _greedy_steps(self, st, ws, B, lo, hi):
        """Steps lo..hi-1 of run_model.py:225-340 at beam 1: two library calls per step (KV-cached decoder step with the
        arg-max of the output distribution, then the hypothesis bookkeeping), no torch op, no host round trip."""
        lib, s = _lib.lib(), _lib.cur_stream()
        for step in range(lo, hi):
            self._step(ws, B, 1, step, st["tok"], None, None, st["best_id"], st["best_p"])
            _lib.check(lib.fira_greedy_advance(s, C.byref(self.model.dims), B, step, _lib.ptr(st["best_id"]),
                                               _lib.ptr(st["best_p"]), _lib.ptr(st["sou"]), _lib.ptr(st["sub"]),
                                               _lib.ptr(st["out"]), _lib.ptr(st["length"]), _lib.ptr(st["prob"]),
                                               _lib.ptr(st["alive"]), _lib.ptr(st["tok"]), _lib.ptr(st["n_alive"])),
                       "fira_greedy_advance")

    def _greedy_start(self, db: DeviceBatch, chunk: int, use_graphs: bool):
        """Encoder pass + reset of the hypothesis state for one batch on the CURRENT stream; returns the loop's context."""
        cfg = self.cfg
        B, T = db.B, cfg.tar_len
        ws = self._begin(db, 1)
        st = self._greedy_state(B)
        st["sou"].copy_(db.sou)
        st["sub"].copy_(db.sub_token)
        self._greedy_reset(st)
        bounds = [(lo, min(lo + chunk, T - 1)) for lo in range(0, T - 1, chunk)]
        if use_graphs and st["graphs"] is None:
            # warm-up outside capture (lazy initialisation inside the library / torch), then capture every chunk
            self._greedy_steps(st, ws, B, 0, 1)
            torch.cuda.synchronize()
            graphs = []
            for lo, hi in bounds:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._greedy_steps(st, ws, B, lo, hi)
                graphs.append(g)
            st["graphs"] = graphs
            self._greedy_reset(st)
        return dict(st=st, ws=ws, B=B, bounds=bounds, i=0, use_graphs=use_graphs)

    def _greedy_launch(self, ctx):
        """Enqueue the next chunk of steps (one hipGraph replay) on the current stream."""
        lo, hi = ctx["bounds"][ctx["i"]]
        if ctx["use_graphs"]:
            ctx["st"]["graphs"][ctx["i"]].replay()
        else:
            self._greedy_steps(ctx["st"], ctx["ws"], ctx["B"], lo, hi)

    def _greedy_done(self, ctx) -> bool:
        """After the chunk launched last: read the alive counter back (synchronises with the chunk) and advance."""
        lo, hi = ctx["bounds"][ctx["i"]]
        ctx["i"] += 1
        if ctx["i"] >= len(ctx["bounds"]):
            return True
        return int(ctx["st"]["n_alive"][hi - 1].item()) == 0      # every hypothesis has emitted <eos> (run_model.py:276-279)

    @torch.no_grad()
    def greedy(self, db: DeviceBatch, chunk: int = 5, use_graphs: bool = True):
        """Returns (tokens [B,T] int64 starting with <start>, lengths [B], probability [B]).

        The step loop is launch-bound (~58 small kernels per generated token), so it is captured once per batch size
        into hipGraphs of ``chunk`` steps each and replayed; between chunks one counter is read back to stop as soon
        as every hypothesis has emitted <eos> (run_model.py:276-279)."""
        ctx = self._greedy_start(db, chunk, use_graphs)
        while True:
            self._greedy_launch(ctx)
            if self._greedy_done(ctx):
                break
        st = ctx["st"]
        return st["out"].long(), st["length"].long(), st["prob"].clone()

    @torch.no_grad()
    def greedy_many(self, dbs, in_flight: int = 4, chunk: int = 5):
        """Greedy search over a sequence of batches with ``in_flight`` of them on the GPU at once, each on its own stream
        (its own workspace, hypothesis state and captured graphs); results are returned in the order of ``dbs``.

        One decode step is ~58 dependent launches of 16-48 workgroups each: a single batch of 64 keeps a fraction of the 256
        CUs busy and the loop is bound by the launch chain, not by the chip.  The reference walks the test set batch after
        batch (run_model.py:225); nothing couples two batches, so independent chains share the chip.  HIP multiplexes the
        lanes' streams onto GPU_MAX_HW_QUEUES hardware queues: on the default 4, three lanes gave x1.7 step-tokens/s and four fell
        BELOW one lane (x0.8) once the process had created more streams (a trainer's) -- two lanes on one queue run one after
        the other.  On 8 queues (run_model.py / bench.py / this package set GPU_MAX_HW_QUEUES=8 before HIP initialises) three
        lanes give x2.06 and four x2.35 in a process that trained first; six collapse again (profiles/r6_probes.md).
        Same arithmetic, same ids as ``greedy`` batch by batch."""
        dbs = list(dbs)
        n_lane = max(1, min(in_flight, len(dbs)))
        if not hasattr(self, "_lanes") or len(self._lanes) < n_lane:
            streams = concurrent_streams(n_lane, self.model.device_)
            old = getattr(self, "_lanes", [])
            self._lanes = [(old[k][0] if k < len(old) else (Searcher(self.model, kv_bf16=bool(self.flags)) if k else self),
                            streams[k]) for k in range(n_lane)]
        main = torch.cuda.current_stream()
        results = [None] * len(dbs)
        active = [None] * n_lane                                # per lane: (batch index, loop context)
        nxt = 0
        for lane, stream in self._lanes[:n_lane]:
            stream.wait_stream(main)
        while True:
            busy = False
            for k in range(n_lane):
                lane, stream = self._lanes[k]
                with torch.cuda.stream(stream):
                    if active[k] is not None:
                        j, ctx = active[k]
                        if lane._greedy_done(ctx):              # (synchronises with this lane's last chunk only)
                            st = ctx["st"]
                            results[j] = (st["out"].long(), st["length"].long(), st["prob"].clone())
                            for r_ in results[j]:
                                r_.record_stream(main)          # allocated on the lane's stream, consumed on the caller's
                            active[k] = None
                        else:
                            lane._greedy_launch(ctx)
                    if active[k] is None and nxt < len(dbs):
                        ctx = lane._greedy_start(dbs[nxt], chunk, True)
                        lane._greedy_launch(ctx)
                        active[k] = (nxt, ctx)
                        nxt += 1
                busy = busy or active[k] is not None
            if not busy:
                break
        for lane, stream in self._lanes[:n_lane]:
            main.wait_stream(stream)
        return results

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

    def best(self, gen, length, prob) -> List[List[int]]:
        """argmax-probability hypothesis per item, first on ties (run_model.py:351-352)."""
        if gen.dim() == 2:
            return [row[:n] for row, n in zip(gen.tolist(), length.tolist())]
        j = torch.argmax(prob, dim=1)        # first maximal index, like np.argmax
        g = gen[torch.arange(gen.shape[0], device=gen.device), j].tolist()
        n = length[torch.arange(gen.shape[0], device=gen.device), j].tolist()
        return [row[:k] for row, k in zip(g, n)]
