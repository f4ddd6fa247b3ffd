"""Wall time per stage of the scoring drivers (diagnostics; off unless PK_STAGE_TIMES=1).

`with stage("name"):` around a step adds its wall time to a process-wide table; `report()`
returns {name: (seconds, calls)}.  Steps that run on the reader threads of
score_genome.prefetched overlap the main thread's work: their names carry the thread's role, and
what the main thread loses to them shows up as its own "wait for the reader" stage.  With the
switch off `stage()` hands out one shared do-nothing context (no clock is read).
"""
import contextlib
import os
import threading
import time

ENABLED = os.environ.get("PK_STAGE_TIMES") == "1"
_acc = {}
_lock = threading.Lock()
_null = contextlib.nullcontext()


class _Stage:
    __slots__ = ("name", "t0")

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        dt = time.perf_counter() - self.t0
        with _lock:
            s, c = _acc.get(self.name, (0.0, 0))
            _acc[self.name] = (s + dt, c + 1)
        return False


def stage(name):
    if not ENABLED:
        return _null
    if threading.current_thread() is not threading.main_thread():
        name += " [reader thread]"
    return _Stage(name)


def enable(on=True):
    global ENABLED
    ENABLED = bool(on)


def reset():
    with _lock:
        _acc.clear()


def report():
    with _lock:
        return dict(_acc)
