"""Background preparation of device batches (device-resident, prefetched data path).

Collating a batch (CSR concatenation, computed-node lists, one pinned host->device copy of ~0.5 MB, see
``model.DeviceBatch``) costs the host about a millisecond at batch 32; the training step itself is enqueued
asynchronously, so one worker thread preparing batch k+1 (and k+2) while the GPU runs batch k keeps the host off the
critical path.  The reference does this work -- plus a 108 MB dense float64 adjacency per 32 commits -- synchronously
inside the step loop (run_model.py:87-103).
"""
from __future__ import annotations

import os
import sys
import threading
from typing import Callable, Iterable, Iterator, TypeVar

T = TypeVar("T")
U = TypeVar("U")


class prefetch(Iterator[U]):
    """Iterator over ``prepare(item)`` for every item, IN ORDER, computed up to ``depth`` items ahead by ``workers``
    threads (numpy releases the GIL inside its sorts / scans, so two workers prepare batch k+1 and k+2 side by side; the
    default comes from FIRA_PREFETCH_WORKERS, 1 if unset).  Exceptions raised by ``prepare`` (or by the iterable) surface
    at the consumer, in order.  ``close()`` (also called when the iterator is exhausted or garbage-collected) stops the
    workers and drops what they had prepared, so an early ``break`` out of the consuming loop does not leave a thread
    blocked holding device batches."""

    _DONE = object()

    def __init__(self, items: Iterable[T], prepare: Callable[[T], U], depth: int = 2, workers: int = 0, device=None):
        # The workers' preparation is a string of short numpy calls that hold the GIL; the consumer (the training loop)
        # needs the GIL for a few microseconds between its long GIL-free library calls and would otherwise wait for the
        # interpreter's default 5 ms forced-switch interval each time (measured: +2.5 ms per 4 ms step).
        iv = float(os.environ.get("FIRA_SWITCH_INTERVAL", "1e-4"))
        self._saved_interval = None
        if iv > 0 and sys.getswitchinterval() > iv:
            self._saved_interval = sys.getswitchinterval()       # restored by close()
            sys.setswitchinterval(iv)
        # The CUDA/HIP current device is thread-local: a worker that builds device batches must select the rank's GPU
        # itself, or its pinned allocations, copies and events land on cuda:0 (``device``: torch.device / index / None).
        self._device = device
        if workers <= 0:
            workers = max(1, int(os.environ.get("FIRA_PREFETCH_WORKERS", "1")))
        self._depth = max(1, depth, workers)
        self._cv = threading.Condition()
        self._ready = {}                          # sequence number -> (value, error)
        self._next_in = 0                         # next sequence number a worker will take
        self._next_out = 0                        # next sequence number the consumer will return
        self._end = None                          # sequence number of the end marker, once the iterable is exhausted
        self._stop = False
        self._finished = False
        self._it = iter(items)
        self._src_lock = threading.Lock()

        def work():
            if self._device is not None:
                import torch
                d = torch.device(self._device)
                if d.type == "cuda":
                    torch.cuda.set_device(d)
            while True:
                with self._cv:                    # at most `depth` items prepared or in preparation ahead of the consumer
                    while not self._stop and self._next_in - self._next_out >= self._depth:
                        self._cv.wait(0.05)
                    if self._stop or self._end is not None:
                        return
                    with self._src_lock:
                        seq = self._next_in
                        try:
                            it = next(self._it)
                        except StopIteration:
                            self._end = seq
                            self._cv.notify_all()
                            return
                        except BaseException as e:    # noqa: BLE001 - forwarded to the consumer
                            self._ready[seq] = (None, e)
                            self._end = seq + 1
                            self._next_in = seq + 1
                            self._cv.notify_all()
                            return
                        self._next_in = seq + 1
                try:
                    out = (prepare(it), None)
                except BaseException as e:            # noqa: BLE001 - forwarded to the consumer
                    out = (None, e)
                with self._cv:
                    if self._stop:
                        return
                    self._ready[seq] = out
                    self._cv.notify_all()

        self._threads = [threading.Thread(target=work, daemon=True) for _ in range(workers)]
        for th in self._threads:
            th.start()

    def __iter__(self):
        return self

    def __next__(self) -> U:
        if self._finished:
            raise StopIteration
        with self._cv:
            while self._next_out not in self._ready:
                if self._end is not None and self._next_out >= self._end:
                    break
                self._cv.wait(0.05)
            if self._next_out in self._ready:
                val, err = self._ready.pop(self._next_out)
                self._next_out += 1
                self._cv.notify_all()
            else:
                val, err = self._DONE, None
        if err is not None:
            self.close()
            raise err
        if val is self._DONE:
            self.close()
            raise StopIteration
        return val

    def close(self):
        self._finished = True
        with self._cv:
            self._stop = True
            self._ready.clear()                   # release the prepared batches
            self._cv.notify_all()
        for th in self._threads:
            if th.is_alive() and threading.current_thread() is not th:
                th.join(timeout=5.0)
        if self._saved_interval is not None:
            sys.setswitchinterval(self._saved_interval)
            self._saved_interval = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
