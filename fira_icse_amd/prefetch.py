"""Background preparation of device batches (device-resident, prefetched data path).

Collating a batch (CSR concatenation, computed-node lists, one pinned host->device copy of ~0.5 MB, see
``model.DeviceBatch``) costs the host about a millisecond at batch 32; the training step itself is enqueued
asynchronously, so one worker thread preparing batch k+1 (and k+2) while the GPU runs batch k keeps the host off the
critical path.  The reference does this work -- plus a 108 MB dense float64 adjacency per 32 commits -- synchronously
inside the step loop (run_model.py:87-103).
"""
from __future__ import annotations

import os
import queue
import sys
import threading
from typing import Callable, Iterable, Iterator, TypeVar

T = TypeVar("T")
U = TypeVar("U")


class prefetch(Iterator[U]):
    """Iterator over ``prepare(item)`` for every item, computed up to ``depth`` items ahead on a worker thread.
    Exceptions raised by ``prepare`` (or by the iterable) surface at the consumer, in order.  ``close()`` (also called
    when the iterator is exhausted or garbage-collected) stops the worker and drops what it had prepared, so an early
    ``break`` out of the consuming loop does not leave a thread blocked on a full queue holding device batches."""

    _DONE = object()

    def __init__(self, items: Iterable[T], prepare: Callable[[T], U], depth: int = 2):
        # The worker's preparation is a string of short numpy calls that hold the GIL; the consumer (the training loop)
        # needs the GIL for a few microseconds between its long GIL-free library calls and would otherwise wait for the
        # interpreter's default 5 ms forced-switch interval each time (measured: +2.5 ms per 4 ms step).
        iv = float(os.environ.get("FIRA_SWITCH_INTERVAL", "1e-4"))
        if iv > 0 and sys.getswitchinterval() > iv:
            sys.setswitchinterval(iv)
        self._q: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
        self._stop = threading.Event()
        self._finished = False

        def put(x) -> bool:                       # False once the consumer has gone away
            while not self._stop.is_set():
                try:
                    self._q.put(x, timeout=0.05)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            try:
                for it in items:
                    if self._stop.is_set() or not put((prepare(it), None)):
                        return
            except BaseException as e:            # noqa: BLE001 - forwarded to the consumer
                put((None, e))
                return
            put((self._DONE, None))

        self._th = threading.Thread(target=work, daemon=True)
        self._th.start()

    def __iter__(self):
        return self

    def __next__(self) -> U:
        if self._finished:
            raise StopIteration
        val, err = self._q.get()
        if err is not None:
            self.close()
            raise err
        if val is self._DONE:
            self.close()
            raise StopIteration
        return val

    def close(self):
        self._finished = True
        self._stop.set()
        try:
            while True:
                self._q.get_nowait()              # release the prepared batches
        except queue.Empty:
            pass
        if self._th.is_alive() and threading.current_thread() is not self._th:
            self._th.join(timeout=5.0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
