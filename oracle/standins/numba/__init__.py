"""Stand-in for numba (absent in the build container).

``njit`` returns the plain-Python function, wrapped so that ndarray *subclass* arguments (AgentState,
WorldObj) arrive as base-class ndarray views -- which is how numba's unboxing presents them to compiled
code.  Without this, the reference's AgentState.__getitem__ hook fires inside obs.py on boolean-mask
results whose length happens to equal AgentState.dim (9) and raises AttributeError; that is an artefact of
interpreting obs.py, not behaviour of the reference under numba.
"""
import functools

import numpy as np


def _plain(a):
    if isinstance(a, np.ndarray) and type(a) is not np.ndarray:
        return a.view(np.ndarray)
    return a


def _wrap(fn):
    @functools.wraps(fn)
    def inner(*args, **kwargs):
        return fn(*[_plain(a) for a in args], **{k: _plain(v) for k, v in kwargs.items()})
    return inner


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return _wrap(args[0])
    return _wrap


jit = njit
