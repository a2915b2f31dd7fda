"""Data parallelism by commit over RCCL/xGMI: one process per GPU, ``torch.distributed``.

The reference's only multi-GPU mode is single-process ``nn.DataParallel`` (run_model.py:392-394): per step it scatters
the batch, broadcasts 124 MB of weights, and reduces 111 MB of gradients onto GPU 0, where Adam runs alone.  Here every
rank keeps its own replica and optimizer state; per step there is ONE all-reduce (sum) of the flat gradient buffer, in
two buckets so that the head+decoder slice overlaps the encoder backward, plus one 2-element all-reduce of
(loss_sum, n_tok); gradients are scaled by 1/n_tok_global inside the fused Adam kernel, which is exactly the
normalisation of run_model.py:105 for the global batch.  No parameter broadcast after step 0.

Backend-agnostic (works with ``gloo`` on CPU tensors, which is how the N>1 logic is tested without GPUs).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's environment; initialises the default process group when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"     # "nccl" IS RCCL on ROCm
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous chunk of a global batch for this rank: ``DataParallel.scatter``'s chunking (ceil(n/world) per
    replica, the last ones may be short or empty), so an N-rank step sees the same commits per replica as the
    reference's N-GPU DataParallel step."""
    per = -(-n // world)
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def shard_indices(idx: Sequence[int], rank: int, world: int) -> List[int]:
    lo, hi = shard_range(len(idx), rank, world)
    return list(idx[lo:hi])


class GradReducer:
    """All-reduce of the flat gradient buffer in two readiness buckets + the (loss_sum, n_tok) pair.

    Order of the collectives of a step (identical on every rank, whatever path computed the gradients):
      1. ``reduce_early``: on the communication stream, behind ``mid_event`` (the decoder's backward pass is through): the
         2-float stats, then the head + decoder bucket ``[0, split)`` -- beside the encoder's backward pass; an event is recorded
         behind them (``early_event``: what ``fira_train_step_end`` / ``wait_early`` wait for before Adam of that slice);
      2. ``start_late`` / ``wait_late``: the encoder bucket ``[split, live)`` once the backward pass is complete.
    ``wire="bf16"`` (BASELINE configs[2]; SURVEY.md 2.2: 55.6 MB instead of 111.2 MB per step on xGMI): each bucket is rounded
    to bf16 into a staging buffer, all-reduced in bf16 and widened back into the fp32 gradient buffer -- Adam, the master
    weights and the token normaliser stay fp32.  Error: one bf16 rounding (2^-9 relative) per rank's contribution and per
    partial sum of the collective (tests/test_parallel.py: 2 ranks, gradient rel-L2 <= 4e-3 against the fp32 reduction).
    """

    def __init__(self, split: int, live: int, group=None, wire: str = "f32"):
        assert wire in ("f32", "bf16"), wire
        self.split, self.live, self.group, self.wire = split, live, group, wire
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._late = None
        self._stage = None
        self._side = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.early_event = None
        if torch.cuda.is_available():
            self.early_event = torch.cuda.Event()
            self.early_event.record()          # torch creates the hipEvent lazily: force it so its handle can be passed on
        # optional per-collective timing (bench.py's process_group object): HIP events around each bucket on the stream
        # that carries it, and around the points where the caller's stream waits for it (= the EXPOSED communication)
        self.timing = False
        self._ev = {}
        self._timings = []

    def bytes_per_step(self) -> int:
        """Bytes every rank contributes to the collectives of one step (both buckets + the stats pair)."""
        return self.live * (2 if self.wire == "bf16" else 4) + 8

    def _mark(self, name, stream=None):
        if self.timing and torch.cuda.is_available():
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream if stream is not None else torch.cuda.current_stream())
            self._ev[name] = e

    def timing_summary(self):
        """Mean milliseconds per step of every timed interval since ``timing`` was switched on; synchronises."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        out = {}
        for ev in self._timings:
            for k, (a, b) in (("allreduce_ms_early", ("early0", "early1")), ("allreduce_ms_late", ("late0", "late1")),
                              ("exposed_comm_ms_early", ("wait_e0", "wait_e1")), ("exposed_comm_ms_late", ("wait_l0", "wait_l1"))):
                if a in ev and b in ev:
                    out.setdefault(k, []).append(ev[a].elapsed_time(ev[b]))
        res = {k: sum(v) / len(v) for k, v in out.items()}
        if "exposed_comm_ms_early" in res and "exposed_comm_ms_late" in res:
            res["exposed_comm_ms"] = res["exposed_comm_ms_early"] + res["exposed_comm_ms_late"]
        res["timed_steps"] = len(self._timings)
        self._timings = []
        return res

    # ---- wire format ------------------------------------------------------------------------------------------------
    def _down(self, gbuf, lo, hi):
        """The tensor that goes on the wire for gbuf[lo:hi] (the slice itself, or its bf16 rounding in the staging buffer)."""
        if self.wire != "bf16":
            return gbuf[lo:hi]
        if self._stage is None or self._stage.device != gbuf.device:
            self._stage = torch.empty(self.live, dtype=torch.bfloat16, device=gbuf.device)
        w = self._stage[lo:hi]
        if gbuf.is_cuda:
            from . import ops
            ops.f32_to_bf16(gbuf[lo:hi], w)
        else:                                   # CPU tensors: the gloo tests of the N > 1 logic
            w.copy_(gbuf[lo:hi])
        return w

    def _up(self, gbuf, lo, hi):
        if self.wire != "bf16":
            return
        if gbuf.is_cuda:
            from . import ops
            ops.bf16_to_f32(self._stage[lo:hi], gbuf[lo:hi])
        else:
            gbuf[lo:hi].copy_(self._stage[lo:hi])

    # ---- the step's collectives -------------------------------------------------------------------------------------
    def reduce_early(self, gbuf: torch.Tensor, stats: torch.Tensor, mid_event=None, pack=None):
        """Stats pair + bucket [0, split) on the communication stream, behind ``mid_event`` (None: behind everything the
        current stream holds).  ``pack``: optional callable that fills ``stats`` from the step's device scalars; it runs on
        the communication stream too (loss and token count are final before ``mid_event``).  Returns ``early_event``
        (None without a GPU: the CPU collectives have completed when the call returns)."""
        self._ev = {}
        cuda = self._side is not None and gbuf.is_cuda
        if not cuda:
            if pack is not None:
                pack()
            if self.world > 1:
                dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)
                w = self._down(gbuf, 0, self.split)
                dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group)
                self._up(gbuf, 0, self.split)
            return None
        side = self._side
        if mid_event is not None:
            side.wait_event(mid_event)
        else:
            side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if pack is not None:
                pack()
            self._mark("early0")
            if self.world > 1:
                dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)
                w = self._down(gbuf, 0, self.split)
                dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group)
                self._up(gbuf, 0, self.split)
            self._mark("early1")
            self.early_event.record(side)
        return self.early_event

    def wait_early(self):
        """The current stream continues behind the early bucket (callers that hand ``early_event`` to the library skip this)."""
        if self.early_event is not None:
            self._mark("wait_e0")
            torch.cuda.current_stream().wait_event(self.early_event)
            self._mark("wait_e1")

    def start_late(self, gbuf: torch.Tensor):
        """Bucket [split, live), asynchronously, behind everything the current stream holds (the backward pass is complete)."""
        if self.world == 1:
            return
        self._mark("late0")
        w = self._down(gbuf, self.split, self.live)
        self._late = dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait_late(self, gbuf: torch.Tensor):
        if self._late is not None:
            self._mark("wait_l0")
            self._late.wait()
            self._up(gbuf, self.split, self.live)
            self._mark("wait_l1")
            self._ev["late1"] = self._ev.get("wait_l1")
            self._late = None
        if self.timing and self._ev:
            self._timings.append({k: v for k, v in self._ev.items() if v is not None})
            self._ev = {}

    def finish(self, gbuf: torch.Tensor, stats: torch.Tensor):
        """Everything of a step, in order, completed on return (CPU tests; a caller without overlap)."""
        self.reduce_early(gbuf, stats, None)
        self.wait_early()
        self.start_late(gbuf)
        self.wait_late(gbuf)


class ShardedOptimizerComm:
    """ZeRO-1 style data parallelism over the flat buffers: reduce-scatter of the gradient, Adam on the owned shard,
    all-gather of the updated parameters -- the same bytes on the wire as the all-reduce (which is a reduce-scatter
    followed by an all-gather), but every rank updates, and keeps Adam moments for, only 1/world of the parameters.

    Each readiness bucket [lo, hi) (head+decoder, encoder) is cut into ``world`` chunks of ceil((hi-lo)/world) elements;
    rank r owns chunk r clipped to the bucket.  The bucket bounds are multiples of 64 elements (256-byte aligned tensor
    offsets, csrc/layout.cpp), so for world in {2, 4, 8, 16, 32, 64} the chunks tile the bucket exactly and the collectives
    run in place on gbuf[lo:hi] / flat[lo:hi].  A world size that does not divide the bucket (3, in the tests) goes through
    a zero-padded scratch window of world*chunk elements: nothing outside [lo, hi) is ever read or written.
    """

    def __init__(self, split: int, live: int, total: int, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.native = dist.is_initialized() and dist.get_backend(group) == "nccl"     # RCCL: reduce_scatter / all_gather
        self.buckets = []
        for lo, hi in ((0, split), (split, live)):
            chunk = -(-(hi - lo) // self.world)
            a = min(hi, lo + self.rank * chunk)
            self.buckets.append({"lo": lo, "hi": hi, "chunk": chunk, "a": a, "b": min(hi, a + chunk)})

    def owned(self, b: int) -> Tuple[int, int]:
        """[a, b) of bucket ``b`` this rank updates (may be empty on trailing ranks of a tiny bucket)."""
        return self.buckets[b]["a"], self.buckets[b]["b"]

    def reduce_scatter(self, b: int, gbuf: torch.Tensor, out: torch.Tensor):
        """out[:chunk] = sum over ranks of gbuf[window of this rank]; enqueued on the current stream."""
        q = self.buckets[b]
        lo, hi, chunk = q["lo"], q["hi"], q["chunk"]
        if self.world * chunk == hi - lo:
            window = gbuf[lo:hi]
        else:                                      # ragged bucket: zero-padded scratch, the bucket itself is only read
            window = torch.zeros(self.world * chunk, dtype=gbuf.dtype, device=gbuf.device)
            window[:hi - lo].copy_(gbuf[lo:hi])
        if self.world == 1:
            out[:chunk].copy_(window)
        elif self.native:
            dist.reduce_scatter_tensor(out[:chunk], window, op=dist.ReduceOp.SUM, group=self.group)
        else:       # gloo has no reduce-scatter: all-reduce the window, keep the own chunk (the tested N>1 logic is the same)
            tmp = window.clone()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
            out[:chunk].copy_(tmp[self.rank * chunk:(self.rank + 1) * chunk])

    def all_gather(self, b: int, flat: torch.Tensor):
        """Every rank's chunk of the (updated) parameters -> all ranks, in place in the flat parameter buffer."""
        q = self.buckets[b]
        lo, hi, chunk = q["lo"], q["hi"], q["chunk"]
        if self.world == 1:
            return
        exact = self.world * chunk == hi - lo
        if exact:
            window = flat[lo:hi]
        else:                                      # ragged bucket: gather into scratch, copy the bucket's part back
            window = torch.zeros(self.world * chunk, dtype=flat.dtype, device=flat.device)
            window[:hi - lo].copy_(flat[lo:hi])
        mine = window[self.rank * chunk:(self.rank + 1) * chunk]
        if self.native:
            # (a copy of the own chunk as the input: RCCL allows the in-place form, torch's checks on overlapping
            # input / output views have changed between releases and this path cannot be run on the one-GPU boxes)
            dist.all_gather_into_tensor(window, mine.clone(), group=self.group)
        else:
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(parts, mine.clone(), group=self.group)
            for r, part in enumerate(parts):
                window[r * chunk:(r + 1) * chunk].copy_(part)
        if not exact:
            flat[lo:hi].copy_(window[:hi - lo])

    def gather_full(self, shards: List[torch.Tensor], total: int) -> torch.Tensor:
        """Full-length [total] tensor from the per-bucket owned shards of every rank (checkpointing; collective)."""
        full = torch.zeros(total, dtype=shards[0].dtype, device=shards[0].device)
        for b, q in enumerate(self.buckets):
            chunk = q["chunk"]
            if self.world == 1:
                parts = [shards[b]]
            else:
                parts = [torch.empty_like(shards[b]) for _ in range(self.world)]
                dist.all_gather(parts, shards[b].contiguous(), group=self.group)
            for r, part in enumerate(parts):
                a = min(q["hi"], q["lo"] + r * chunk)
                e = min(q["hi"], a + chunk)
                full[a:e].copy_(part[:e - a])
        return full


def gather_lines(lines: List[str], group=None) -> List[str]:
    """Ordered gather of per-rank output lines onto every rank (decode shards are contiguous ranges of the test set, so
    concatenating in rank order keeps ``all_index['test']`` order)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return lines
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, lines, group=group)
    return [l for part in out for l in part]
