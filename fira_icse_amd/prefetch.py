"""Background preparation of device batches (device-resident, prefetched data path).

Collating a batch (CSR concatenation, computed-node lists, pinned host->device copies of ~0.5 MB) costs the host
1-2 ms at batch 32; the training step itself is enqueued asynchronously, so one worker thread preparing batch k+1
(and k+2) while the GPU runs batch k keeps the host off the critical path.  The reference does this work -- plus a
108 MB dense float64 adjacency per 32 commits -- synchronously inside the step loop (run_model.py:87-103).
"""
from __future__ import annotations

import queue
import threading
from typing import Callable, Iterable, Iterator, TypeVar

T = TypeVar("T")
U = TypeVar("U")


def prefetch(items: Iterable[T], prepare: Callable[[T], U], depth: int = 2) -> Iterator[U]:
    """Yield ``prepare(item)`` for every item, computed up to ``depth`` items ahead on a worker thread.
    Exceptions raised by ``prepare`` (or by the iterable) surface at the consumer, in order."""
    q: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
    done = object()

    def work():
        try:
            for it in items:
                q.put((prepare(it), None))
        except BaseException as e:        # noqa: BLE001 - forwarded to the consumer
            q.put((None, e))
            return
        q.put((done, None))

    th = threading.Thread(target=work, daemon=True)
    th.start()
    while True:
        val, err = q.get()
        if err is not None:
            raise err
        if val is done:
            break
        yield val
    th.join()
