"""Drop-in for segmentalist/utils.py: `draw` (utils.py:10-21) through the C ABI host shim."""
import ctypes as C
import random

import numpy as np

from . import _abi


def draw(p_k):
    """Draw from a discrete distribution; consumes one random.random() like the reference."""
    p = np.ascontiguousarray(p_k, dtype=np.float64)
    return int(_abi.lib().segk_draw(p.ctypes.data_as(C.c_void_p), p.size, random.random()))
