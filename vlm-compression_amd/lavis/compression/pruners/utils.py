"""`print_time` -- the reference's only timer (lavis/compression/pruners/utils.py:6-18)."""
import functools
from time import time


def print_time(func):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        start = time()
        ret = func(*args, **kwargs)
        print(f"{func.__name__} spent {time() - start:.3f} s")
        return ret
    return wrapper
