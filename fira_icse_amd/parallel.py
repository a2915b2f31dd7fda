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
    """All-reduce of the flat gradient buffer in two readiness buckets + the (loss_sum, n_tok) pair."""

    def __init__(self, split: int, live: int, group=None):
        self.split, self.live, self.group = split, live, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._pending = None
        self._late = None
        self._side = torch.cuda.Stream() if torch.cuda.is_available() else None
        # optional per-collective timing (bench.py's process_group object): HIP events around each bucket on the stream
        # that carries it, and around the points where the caller's stream waits for it (= the EXPOSED communication)
        self.timing = False
        self._ev = {}
        self._timings = []

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

    def start_early_bucket(self, gbuf: torch.Tensor, mid_event=None):
        """Launch the all-reduce of [0, split) as soon as ``mid_event`` fires (decoder backward done)."""
        if self.world == 1:
            return
        self._ev = {}
        if self._side is not None and mid_event is not None:
            self._side.wait_event(mid_event)
            with torch.cuda.stream(self._side):
                self._mark("early0")
                self._pending = dist.all_reduce(gbuf[:self.split], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if self.timing:               # stream-level wait on the side stream (the host does not block with RCCL)
                    self._pending.wait()
                    self._mark("early1")
        else:
            self._pending = dist.all_reduce(gbuf[:self.split], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def reduce_stats_and_start_late_bucket(self, gbuf: torch.Tensor, stats: torch.Tensor):
        """After the backward pass: all-reduce the 2-element stats (needed first: the token normaliser of every Adam
        call), then launch the encoder bucket asynchronously.  ``wait_early`` / ``wait_late`` then let the caller run
        Adam on the head+decoder slice while the encoder slice is still on the wire."""
        if self.world == 1:
            return
        if self._pending is None:
            self.start_early_bucket(gbuf, None)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)
        self._mark("late0")
        self._late = dist.all_reduce(gbuf[self.split:self.live], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait_early(self):
        if self._pending is not None:
            self._mark("wait_e0")
            self._pending.wait()
            if self.timing and self._side is not None:
                torch.cuda.current_stream().wait_stream(self._side)     # (the side stream already waited for the collective)
            self._mark("wait_e1")
            self._pending = None

    def wait_late(self):
        if getattr(self, "_late", None) is not None:
            self._mark("wait_l0")
            self._late.wait()
            self._mark("wait_l1")
            self._ev["late1"] = self._ev.get("wait_l1")
            self._late = None
            if self.timing and self._ev:
                self._timings.append({k: v for k, v in self._ev.items() if v is not None})
                self._ev = {}

    def finish(self, gbuf: torch.Tensor, stats: torch.Tensor):
        """Reduce the encoder bucket and the 2-element stats (fp32: loss_sum, n_tok); wait for both buckets."""
        if self.world == 1:
            return
        self.reduce_stats_and_start_late_bucket(gbuf, stats)
        self.wait_early()
        self.wait_late()


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
