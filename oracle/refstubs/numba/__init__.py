"""No-op `njit` that emulates numba's *typing* where it changes results — test infrastructure,
see ../README.md.

The reference's hot maze functions are `@njit(cache=True)`; numba is not installed, so the Python
bodies run under NumPy-2 (NEP 50) rules instead of numba's. Two differences alter numerics and are
emulated here, nothing else is touched:

 1. Argument typing. Inside an njit function a python float argument is a strong float64. Under
    NEP 50 a python float is *weak* (`python_float / np.float32 -> float32`). The wrapper therefore
    coerces python-float scalars / lists / tuples of python floats to np.float64 on entry.
 2. `min` / `max`. numba types `min(1.0, max(x, 0.0))` as float64; CPython returns whichever
    *object* wins, possibly the python literal, which then turns the following
    `f32_scalar * (alpha * f32_array ...)` into float32 arithmetic. The wrapper installs typed
    `min`/`max` in the decorated function's module globals: same value, np.float64 when any
    argument is a float, plain int when all are ints.
"""
import builtins
import functools

import numpy as np


def _coerce(a):
    if isinstance(a, float) and not isinstance(a, np.floating):
        return np.float64(a)
    if isinstance(a, (list, tuple)) and len(a) > 0 and all(
            isinstance(x, float) or isinstance(x, np.floating) for x in a):
        return type(a)(np.float64(x) if not isinstance(x, np.floating) else x for x in a)
    return a


def _typed(fn):
    def wrapper(*args):
        r = fn(*args)
        if any(isinstance(a, (float, np.floating)) for a in args):
            return np.float64(r)
        return r
    return wrapper


_typed_min = _typed(builtins.min)
_typed_max = _typed(builtins.max)


def njit(*dargs, **dkw):
    def deco(fn):
        fn.__globals__.setdefault("min", _typed_min)
        fn.__globals__.setdefault("max", _typed_max)

        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            return fn(*[_coerce(a) for a in args], **{k: _coerce(v) for k, v in kwargs.items()})
        return wrapper
    if len(dargs) == 1 and callable(dargs[0]) and not dkw:
        return deco(dargs[0])
    return deco


jit = njit
