"""No-op `njit` that emulates numba's *argument typing* — test infrastructure, see ../README.md."""
import functools
import numpy as np


def _coerce(a):
    if isinstance(a, float) and not isinstance(a, np.floating):
        return np.float64(a)
    if isinstance(a, (list, tuple)) and len(a) > 0 and all(
            isinstance(x, float) or isinstance(x, np.floating) for x in a):
        return type(a)(np.float64(x) if not isinstance(x, np.floating) else x for x in a)
    return a


def njit(*dargs, **dkw):
    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            return fn(*[_coerce(a) for a in args], **{k: _coerce(v) for k, v in kwargs.items()})
        return wrapper
    if len(dargs) == 1 and callable(dargs[0]) and not dkw:
        return deco(dargs[0])
    return deco


jit = njit
