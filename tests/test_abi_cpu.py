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
    # the version the header describes == the one the library reports == the one the binding was written against
    # (built.lib() itself refuses a library that reports another one)
    hdr = int(re.search(r"#define\s+SEGK_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "segk.h")).read()).group(1))
    assert lib.segk_abi_version() == hdr == built.ABI_VERSION


def test_binding_refuses_a_library_of_another_abi_version(built, monkeypatch):
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "ABI_VERSION", built.ABI_VERSION + 1)
    with pytest.raises(built.SegkError, match="ABI version"):
        built.lib()
    monkeypatch.setattr(built, "_lib", None)


def test_integration_stub_structs_match_the_binding(built):
    """INTEGRATION.md shows the ctypes stub a maintainer of the reference would add; its structures are passed to the
    library BY VALUE inside the calls, so a stub that lags behind include/segk.h makes the library read past its end
    (VERDICT r02).  The structures of the document are executed here and compared with the binding's field by field."""
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", txt, flags=re.S)
    stub = [b for b in blocks if "class Corpus(C.Structure)" in b]
    assert len(stub) == 1
    src = stub[0]
    # only the structure definitions: everything up to the first function
    src = src[:src.index("def check(rc)")]
    src = "\n".join(l for l in src.splitlines() if not l.startswith(("import ", "lib = ", "# ", "assert lib.")))
    ns = {"C": ctypes}
    exec(src, ns)
    for name, mine in (("Corpus", built.Corpus), ("KMeansDev", built.KMeansDev), ("Cand", built.CandDev)):
        doc = ns[name]
        assert ctypes.sizeof(doc) == ctypes.sizeof(mine), name
        assert [(f[0], f[1]) for f in doc._fields_] == [(f[0], f[1]) for f in mine._fields_], name


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
