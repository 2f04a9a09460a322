"""Concurrent single calls become one batch.

The reference's functions take ONE scalar and ONE point (`MUL_endo(m, P)`, curve4q.py:405); on the GPU such a call is a
launch of a batch of one, 0.25 ms however little it computes, and calls on one context take turns (include/fourq_amd.h,
"Threads").  A threaded caller of the drop-in module -- a server doing one DH per connection -- would therefore get 4 000
calls a second from a chip that does 2 x 10^8.  `Combiner` removes that without changing the call: while one batch is on
the GPU the calls that arrive are queued, and the next thread to get the GPU takes ALL of them along as one batch.  Nothing
waits for company: a lone call goes out at once (cost: one uncontended lock), and the batch size adapts to the load.

Leadership is handed on rather than kept: the thread that ran a batch returns with its own result and wakes the oldest
waiter, which runs the next batch -- so no caller is held back working for others.

Invariants (all under `_lock`): `_busy` is False only while `_queue` is empty; the queue holds exactly the calls no batch has
taken yet, oldest first; the leader's own call is `_queue[0]` when it takes its batch (it either found the queue empty or
was promoted as the oldest waiter), so a leader is always served by its own batch; every slot's event is set exactly once,
either with `done` (served) or without (promoted); `promoted` is set UNDER THE LOCK when an heir is chosen -- the event is set after
the lock is released, so the flag, not the event, is what says "leadership was handed to this slot"; a waiter that leaves by an
exception takes its slot out of the queue and, if it had been promoted, hands the leadership on (`_abandon`).
"""
import threading


class _Slot:
    __slots__ = ("args", "result", "error", "done", "promoted", "event")

    def __init__(self, args):
        self.args, self.result, self.error, self.done, self.promoted = args, None, None, False, False
        self.event = threading.Event()


class Combiner:
    """`run(list_of_args) -> list_of_results` is called by one thread at a time with the calls queued so far (at most
    `max_batch`); an item of the returned list that is an exception instance is raised in the thread that made that call, and
    an exception raised by `run` itself is raised in every thread of that batch."""

    def __init__(self, run, max_batch=1 << 16):
        self._run, self._max = run, max_batch
        self._lock = threading.Lock()
        self._queue = []
        self._busy = False
        self.calls = self.batches = self.largest = 0          # statistics, updated under the lock

    def __call__(self, *args):
        slot = _Slot(args)
        with self._lock:
            self._queue.append(slot)
            lead = not self._busy
            if lead:
                self._busy = True
        if not lead:
            try:
                slot.event.wait()                             # woken with its result, or as the next leader
            except BaseException:                             # e.g. KeyboardInterrupt in the main thread: leave the queue in order
                self._abandon(slot)
                raise
            if slot.done:
                return self._finish(slot)
        with self._lock:                                      # this thread has the GPU: everything queued up to now is one batch
            batch = self._queue[: self._max]
            del self._queue[: self._max]
        try:
            try:
                results = self._run([s.args for s in batch])
                if len(results) != len(batch):
                    raise RuntimeError("combined call returned %d results for %d calls" % (len(results), len(batch)))
                for s, r in zip(batch, results):
                    if isinstance(r, BaseException):
                        s.error = r
                    else:
                        s.result = r
            except BaseException as e:                        # noqa: BLE001 -- every caller of the batch gets it
                for s in batch:
                    s.result, s.error = None, e
        finally:
            with self._lock:
                self.calls += len(batch)
                self.batches += 1
                self.largest = max(self.largest, len(batch))
                heir = self._pick_heir()
            for s in batch:
                s.done = True
                if s is not slot:
                    s.event.set()
            if heir is not None:
                heir.event.set()                              # not done: it leads the next batch
        return self._finish(slot)

    def _abandon(self, slot):
        """A waiter leaves without its result.  Still queued: it is taken out, so that it can never be picked as the next leader (a
        batch nobody runs would leave `_busy` set and block every later caller).  Already promoted (chosen as heir, not served): the
        leadership it was handed goes on to the oldest waiter, or is given up.  Already in a running batch: nothing to undo."""
        heir = None
        with self._lock:
            if slot in self._queue:
                # `promoted`, not `event.is_set()`: the leader picks its heir under the lock and sets the event after releasing it, and
                # an interrupt in that window must still pass the leadership on
                self._queue.remove(slot)
                if slot.promoted:
                    heir = self._pick_heir()
        if heir is not None:
            heir.event.set()

    def _pick_heir(self):
        """Under the lock: the oldest waiter becomes the next leader (marked `promoted`), or, with nobody waiting, the GPU is free."""
        if self._queue:
            self._queue[0].promoted = True
            return self._queue[0]
        self._busy = False
        return None

    @staticmethod
    def _finish(slot):
        if slot.error is not None:
            raise slot.error
        return slot.result

    def stats(self):
        with self._lock:
            return {"calls": self.calls, "batches": self.batches, "largest_batch": self.largest}
