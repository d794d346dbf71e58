"""Where a prune's wall-clock goes ("report capture / replay / stat / select sub-totals").

Off by default and free then.  With `VLMC_PHASE_TIMERS=1` (bench.py sets it for ONE extra, untimed prune) every phase
boundary synchronises the device and the host clock is read: exclusive times, a nested phase pauses its parent.  The
synchronisations cost a few percent, which is why the timed steps of the bench never run with them."""
from __future__ import annotations

import contextlib
import os
import time

times: dict = {}
_stack: list = []


def enabled():
    return os.environ.get("VLMC_PHASE_TIMERS", "0") == "1"


def reset():
    times.clear()
    _stack.clear()


def _now():
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.perf_counter()


@contextlib.contextmanager
def phase(name):
    if not enabled():
        yield
        return
    now = _now()
    if _stack:
        times[_stack[-1][0]] = times.get(_stack[-1][0], 0.0) + now - _stack[-1][1]
    _stack.append([name, now])
    try:
        yield
    finally:
        now = _now()
        n, t0 = _stack.pop()
        times[n] = times.get(n, 0.0) + now - t0
        if _stack:
            _stack[-1][1] = now
