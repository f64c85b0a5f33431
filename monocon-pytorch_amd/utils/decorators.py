"""reference utils/decorators.py:4-11: wall-clock wrapper returning (result, seconds)."""
import functools
import time


def decorator_timer(fn):
    @functools.wraps(fn)
    def wrapper(*a, **k):
        t0 = time.time()
        out = fn(*a, **k)
        return out, time.time() - t0
    return wrapper
