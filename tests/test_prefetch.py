import threading
import time

import pytest

from fira_icse_amd.prefetch import prefetch


def test_order_and_overlap():
    seen = []

    def prepare(i):
        time.sleep(0.02)
        seen.append((i, threading.current_thread().name))
        return i * i

    t0 = time.time()
    out = []
    for v in prefetch(range(6), prepare, depth=2):
        time.sleep(0.02)                      # the consumer's own work overlaps the worker's
        out.append(v)
    assert out == [i * i for i in range(6)]
    assert all(name != threading.current_thread().name for _, name in seen)
    assert time.time() - t0 < 6 * 0.04 - 0.03


def test_errors_surface_at_the_consumer():
    def prepare(i):
        if i == 2:
            raise ValueError("bad batch")
        return i

    it = prefetch(range(5), prepare)
    assert next(it) == 0 and next(it) == 1
    with pytest.raises(ValueError):
        next(it)


def test_close_stops_the_worker_after_an_early_break():
    made = []

    def prepare(i):
        made.append(i)
        return i

    it = prefetch(range(1000), prepare, depth=2)
    for v in it:
        if v == 3:
            break
    it.close()
    n = len(made)
    time.sleep(0.2)
    assert len(made) == n and n < 20          # nothing is prepared after close(); the worker is not blocked on put()
    assert not any(th.is_alive() for th in it._threads)


@pytest.mark.parametrize("workers", [1, 2, 3])
def test_several_workers_keep_the_order_and_share_the_work(workers):
    names, spans = set(), []

    def prepare(i):
        t0 = time.monotonic()
        time.sleep(0.03 if i % 2 else 0.01)   # uneven preparation times: later items finish first
        names.add(threading.current_thread().name)
        spans.append((t0, time.monotonic()))
        return i

    out = list(prefetch(range(12), prepare, depth=4, workers=workers))
    assert out == list(range(12))
    assert len(names) == workers
    # the preparations of two workers really overlap in time (a structural check: wall-clock bounds are flaky on a
    # loaded test host); one worker never overlaps itself
    spans.sort()
    overlaps = sum(1 for (a0, a1), (b0, b1) in zip(spans, spans[1:]) if b0 < a1)
    assert (overlaps > 0) == (workers >= 2), (workers, overlaps)


def test_errors_keep_their_place_with_two_workers():
    def prepare(i):
        if i == 3:
            raise ValueError("bad batch")
        time.sleep(0.005)
        return i

    it = prefetch(range(8), prepare, depth=4, workers=2)
    assert [next(it) for _ in range(3)] == [0, 1, 2]
    with pytest.raises(ValueError):
        next(it)
    assert not any(th.is_alive() for th in it._threads)


def test_an_exhausted_empty_iterable():
    assert list(prefetch([], lambda x: x, workers=2)) == []
