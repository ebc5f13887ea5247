"""
CPU-only checks of the drop-in boundary: libsegk.so (cross-compiled for gfx950) loads, exports
every symbol include/segk.h declares, the ctypes table covers them all, the pure-host A9 shims
work, and the product refuses to run its hot path without a GPU (no CPU fallback).
"""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "segk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(segk_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def built():
    from segmentalist_amd import _abi
    if not os.path.exists(_abi.LIB_PATH):
        _abi.build()
    return _abi


def test_library_exports_every_header_symbol(built):
    lib = ctypes.CDLL(built.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), s
    assert set(syms) == set(built.SIGNATURES), set(syms) ^ set(built.SIGNATURES)
    assert built.lib().segk_abi_version() == 1


def test_struct_layout_matches_header(built):
    # field order/size of the ctypes mirrors (LP64): segk_corpus 16 fields, segk_kmeans 10
    assert ctypes.sizeof(built.Corpus) == 8 + 8 + 4 + 4 + 8 * 4 + 8 * 3 + 4 + 4 + 8 + 4 + 4 + 8 + 8      # + band_ids, band_dur
    assert ctypes.sizeof(built.KMeansDev) == 8 * 6 + 8 + 8 + 8 + 8     # K_max padded to 8


def test_host_shims_match_oracle(built, golden):
    from oracle import c_oracle as co
    from segmentalist_amd import _cython_utils as cu
    g = golden("kernels")
    off = 0
    for n, want in zip(g["lse_n"], g["lse_out"]):
        a = g["lse_in"][off:off + n]
        off += n
        assert cu.logsumexp(a) == co.logsumexp(a)
        assert np.isclose(cu.logsumexp(a), want, rtol=1e-15)
    L = built.lib()
    p = np.ascontiguousarray(g["draw_p"])
    for u, k in zip(g["draw_u"], g["draw_k"]):
        assert L.segk_draw(p.ctypes.data_as(ctypes.c_void_p), p.size, float(u)) == k
    y = np.array([3.0, 0.5, 2.0])
    assert cu.sum_doubles(y) == 5.5
    assert cu.sum_ints(np.array([3, 4, 5])) == 12
    assert np.isclose(cu.sum_log(y), np.log(3.0) + np.log(0.5) + np.log(2.0))
    assert cu.sum_square_a_times_b(y, y) == 3.0 ** 3 + 0.5 ** 3 + 2.0 ** 3


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from segmentalist_amd.kmeans_components import KMeansComponents
    with pytest.raises(built.SegkError):
        KMeansComponents(np.zeros((4, 2), np.float32), np.zeros(4, int), 2)
    h = ctypes.c_void_p()
    assert built.lib().segk_create(0, ctypes.byref(h)) < 0
    assert b"no" in built.lib().segk_last_error().lower()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "segmentalist_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "segk_oracle" not in txt, f
