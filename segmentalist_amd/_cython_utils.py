"""
Drop-in for segmentalist/_cython_utils.pyx (the reference's only native module): the six
`cpdef` functions, bound to the A9 host shims of libsegk.so (include/segk.h).
"""
import ctypes as C
import random

import numpy as np

from . import _abi


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.c_void_p)


def logsumexp(a):                       # _cython_utils.pyx:13-25
    a, p = _d(a)
    return _abi.lib().segk_logsumexp(p, a.size)


def sum_doubles(y):                     # :30-36
    y, p = _d(y)
    return _abi.lib().segk_sum_doubles(p, y.size)


def sum_ints(y):                        # :41-47
    y = np.ascontiguousarray(y, dtype=np.int64)
    return int(_abi.lib().segk_sum_ints(y.ctypes.data_as(C.c_void_p), y.size))


def sum_log(y):                         # :52-58
    y, p = _d(y)
    return _abi.lib().segk_sum_log(p, y.size)


def sum_square_a_times_b(a, b):         # :63-70
    a, pa = _d(a)
    b, pb = _d(b)
    return _abi.lib().segk_sum_square_a_times_b(pa, pb, a.size)


def draw(p_k):                          # :75-89
    p, pp = _d(p_k)
    return int(_abi.lib().segk_draw(pp, p.size, random.random()))
