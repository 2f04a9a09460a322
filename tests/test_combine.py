"""fourq_amd.combine.Combiner: concurrent single calls leave as one batch (host logic, no GPU).

The runner here is a stand-in that records the batches it is given; the GPU side of the same mechanism is
tests/test_gpu_multi.py::test_concurrent_single_calls_are_combined."""
import threading
import time

import pytest

from fourq_amd.combine import Combiner


def test_lone_calls_are_batches_of_one_and_keep_order_of_results():
    seen = []

    def run(items):
        seen.append(list(items))
        return [a + b for a, b in items]

    cb = Combiner(run)
    assert [cb(i, 10 * i) for i in range(5)] == [11 * i for i in range(5)]
    assert seen == [[(i, 10 * i)] for i in range(5)]
    assert cb.stats() == {"calls": 5, "batches": 5, "largest_batch": 1}


def test_calls_arriving_while_a_batch_runs_leave_together():
    gate, first_in = threading.Event(), threading.Event()
    sizes = []

    def run(items):
        sizes.append(len(items))
        if len(sizes) == 1:
            first_in.set()
            assert gate.wait(30)                      # the first batch "is on the GPU" until every other caller has queued
        return [x * x for (x,) in items]

    cb = Combiner(run)
    results = {}

    def call(x):
        results[x] = cb(x)

    t0 = threading.Thread(target=call, args=(0,))
    t0.start()
    assert first_in.wait(30)
    others = [threading.Thread(target=call, args=(x,)) for x in range(1, 41)]
    for t in others:
        t.start()
    deadline = time.time() + 30
    while len(cb._queue) < 40 and time.time() < deadline:
        time.sleep(0.001)
    assert len(cb._queue) == 40
    gate.set()
    for t in [t0] + others:
        t.join(30)
    assert results == {x: x * x for x in range(41)}
    assert sizes == [1, 40] and cb.stats() == {"calls": 41, "batches": 2, "largest_batch": 40}
    assert not cb._busy and not cb._queue


def test_max_batch_and_handing_on_of_the_lead():
    gate, first_in = threading.Event(), threading.Event()
    sizes, leaders = [], []

    def run(items):
        sizes.append(len(items))
        leaders.append(threading.current_thread().name)
        if len(sizes) == 1:
            first_in.set()
            assert gate.wait(30)
        return [x for (x,) in items]

    cb = Combiner(run, max_batch=16)
    out = []
    threads = [threading.Thread(target=lambda x=x: out.append(cb(x)), name="t%d" % x) for x in range(41)]
    threads[0].start()
    assert first_in.wait(30)
    for t in threads[1:]:
        t.start()
    deadline = time.time() + 30
    while len(cb._queue) < 40 and time.time() < deadline:
        time.sleep(0.001)
    gate.set()
    for t in threads:
        t.join(30)
    assert sorted(out) == list(range(41)) and sizes == [1, 16, 16, 8]
    assert len(set(leaders)) == 4                       # four batches, four different leaders: nobody works for others twice
    assert not cb._busy and not cb._queue


def test_errors_reach_the_right_callers():
    def run(items):
        if any(x == 13 for (x,) in items):
            raise RuntimeError("whole batch failed")
        return [ValueError("odd %d" % x) if x & 1 else x for (x,) in items]

    cb = Combiner(run)
    assert cb(4) == 4
    with pytest.raises(ValueError, match="odd 7"):
        cb(7)
    with pytest.raises(RuntimeError, match="whole batch"):
        cb(13)
    assert cb(6) == 6 and not cb._busy                  # the combiner is usable after either kind of failure

    def bad(items):
        return []

    with pytest.raises(RuntimeError, match="0 results for 1 calls"):
        Combiner(bad)(1)


def test_many_threads_many_calls():
    def run(items):
        time.sleep(0.0005)
        return [a * b for a, b in items]

    cb = Combiner(run)
    bad = []

    def work(t):
        for i in range(200):
            if cb(t, i) != t * i:
                bad.append((t, i))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(16)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    st = cb.stats()
    assert not bad and st["calls"] == 3200 and st["batches"] < 3200 and 1 < st["largest_batch"] <= 16


def test_a_waiter_that_leaves_by_an_exception_does_not_block_later_callers(monkeypatch):
    """a waiter interrupted inside event.wait() (KeyboardInterrupt in the main thread) used to stay in the queue; picked
    as the next leader later, nobody ran its batch and every following call blocked.  Both cases: interrupted while still queued
    behind a running batch, and interrupted after having been promoted to lead the next one -- and interrupted in the
    WINDOW between the leader choosing it as heir under the lock and setting its event after releasing the lock."""
    import fourq_amd.combine as combine

    class Interrupted(BaseException):
        pass

    for when in ("queued", "promoted", "window"):
        gate, first_in, waiting = threading.Event(), threading.Event(), threading.Event()
        chosen, abandoned = threading.Event(), threading.Event()

        class Slot(combine._Slot):
            def __init__(self, args):
                super().__init__(args)
                if args == ("victim",):
                    real_wait = self.event.wait

                    def wait(timeout=None):
                        waiting.set()
                        if when == "promoted":
                            real_wait(30)                 # woken as the next leader ... and interrupted right then
                        if when == "window":
                            assert chosen.wait(30)        # the leader has picked this slot and is about to set its event
                        raise Interrupted()
                    self.event.wait = wait
                    if when == "window":
                        real_set = self.event.set

                        def set_():                       # the leader, past its lock: hold it here until the victim has left
                            chosen.set()
                            assert abandoned.wait(30)
                            real_set()
                        self.event.set = set_

        monkeypatch.setattr(combine, "_Slot", Slot)
        sizes = []

        def run(items):
            sizes.append(len(items))
            if len(sizes) == 1:
                first_in.set()
                assert gate.wait(30)
            return [x for (x,) in items]

        cb = Combiner(run)
        out = {}
        t0 = threading.Thread(target=lambda: out.setdefault("first", cb("first")))
        t0.start()
        assert first_in.wait(30)

        def victim():
            try:
                cb("victim")
            except Interrupted:
                out["victim"] = "interrupted"
            abandoned.set()
        tv = threading.Thread(target=victim)
        tv.start()
        assert waiting.wait(30)
        if when == "queued":
            tv.join(30)                                   # it has left the queue before the first batch ends
        t1 = threading.Thread(target=lambda: out.setdefault("late", cb("late")))
        t1.start()
        time.sleep(0.05)
        gate.set()
        for t in (t0, tv, t1):
            t.join(30)
            assert not t.is_alive(), "a caller is blocked (%s)" % when
        assert out == {"first": "first", "victim": "interrupted", "late": "late"}
        assert cb("after") == "after"                     # and the combiner is idle again: a lone call leads at once
        assert not cb._busy and not cb._queue


def test_suite_is_clean_in_python_development_mode():
    """this file's threads, locks and events under `python -X dev` with faulthandler (unjoined threads, unclosed
    resources and misuse of the threading primitives become visible there) -- run as a child process, warnings as errors."""
    import os
    import subprocess
    import sys
    if os.environ.get("FOURQ_COMBINE_DEV_CHILD"):
        return
    env = dict(os.environ, PYTHONDEVMODE="1", PYTHONFAULTHANDLER="1", FOURQ_COMBINE_DEV_CHILD="1")
    proc = subprocess.run([sys.executable, "-X", "dev", "-X", "faulthandler", "-W", "error::ResourceWarning", "-m", "pytest", "-q", "-x",
                           "-p", "no:cacheprovider", os.path.abspath(__file__)], capture_output=True, text=True, env=env, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    assert "Fatal Python error" not in proc.stderr
