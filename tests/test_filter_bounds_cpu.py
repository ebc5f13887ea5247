"""
The error bound of the one-product fp16 pre-filter (segk_score_h1.hip, k_kmeans_score_h1), checked on a numpy
model of its operands: rows and means are scaled by a power of two so that the largest element lies in
[2^12, 2^13) and rounded once to fp16 (np.float16: round-to-nearest-even with gradual underflow, as
v_cvt_f16_f32); the products are exact in fp32 and their sum is taken here in float64, so what is measured
is the operand rounding alone -- the term 1.01 * 2^-10 |x| M of tau_A.  (The fp32 accumulation error is
covered by the split-precision margin E1', tested on the device in tests/test_gpu_kmeans.py.)
"""
import numpy as np
import pytest


def _scaled_f16(A):
    vmax = np.abs(A).max()
    e = 13 - np.frexp(np.float32(vmax))[1] if vmax > 0 else 0
    return np.ldexp(A.astype(np.float32), e).astype(np.float16).astype(np.float64), e


def _worst_ratio(X, Mn):
    x1, ea = _scaled_f16(X)
    m1, eb = _scaled_f16(Mn)
    approx = np.ldexp(x1 @ m1.T, -(ea + eb))
    true = X.astype(np.float64) @ Mn.astype(np.float64).T
    xn = np.linalg.norm(X.astype(np.float64), axis=1)
    Mmax = np.linalg.norm(Mn.astype(np.float64), axis=1).max()
    bound = 1.01 * 2.0 ** -10 * xn * Mmax
    return (np.abs(approx - true) / bound[:, None]).max()


@pytest.mark.parametrize("D", [8, 40, 100, 128])
def test_operand_rounding_bound_random(D):
    rs = np.random.RandomState(D)
    for scale in (1.0, 1e-3, 3e4):
        X = (rs.randn(600, D) * scale).astype(np.float32)
        Mn = (rs.randn(300, D) * scale).astype(np.float32)
        assert _worst_ratio(X, Mn) < 0.5


def test_operand_rounding_bound_adversarial():
    """Same-sign operands sitting just above fp16 rounding midpoints (every product errs the same way), a
    wide dynamic range that flushes most elements, and rows equal to a mean."""
    rs = np.random.RandomState(1)
    D = 128
    base = 1.0 + (2.0 ** -11) * (1 + 2.0 ** -9)               # just past the midpoint between two fp16 values
    X = (base * 2.0 ** rs.randint(0, 3, size=(200, D))).astype(np.float32)
    Mn = (base * 2.0 ** rs.randint(0, 3, size=(100, D))).astype(np.float32)
    r = _worst_ratio(X, Mn)
    assert 0.2 < r < 1.0, r                                   # the bound is nearly attained, never exceeded
    X = (rs.randn(400, D) * 10.0 ** rs.uniform(-9, 0, size=(400, D))).astype(np.float32)
    Mn = (rs.randn(150, D) * 10.0 ** rs.uniform(-9, 0, size=(150, D))).astype(np.float32)
    Mn[3] = X[5]
    assert _worst_ratio(X, Mn) < 1.0


def _resid_norms(A):
    """|a - a1| per row as k_corpus_resid_sp / k_kmeans_prepare_sp compute it: an element whose piece is zero
    or subnormal in fp16 counts with its full magnitude."""
    a1, e = _scaled_f16(A)
    sc = np.ldexp(A.astype(np.float64), e)
    r = np.where(np.abs(a1) < 2.0 ** -14, np.abs(sc), np.abs(sc - a1))
    return np.ldexp(np.sqrt((r * r).sum(1)), -e), a1, e


@pytest.mark.parametrize("flush", [False, True], ids=["gradual-underflow", "flush-to-zero"])
def test_measured_residual_bound(flush):
    """The margin the kernel uses: (|x| + e_x) E_m + e_x M with the measured residual norms -- holds whether
    the matrix pipe keeps fp16 subnormal inputs or flushes them, and is several times tighter than the
    a-priori 2^-10 |x| M on ordinary data."""
    rs = np.random.RandomState(5)
    for D, scale, wide in ((100, 1.0, False), (128, 30.0, False), (64, 1.0, True), (16, 1e-3, False)):
        X = (rs.randn(500, D) * scale).astype(np.float32)
        Mn = (rs.randn(200, D) * scale).astype(np.float32)
        if wide:
            X = (X * 10.0 ** rs.uniform(-9, 0, size=X.shape)).astype(np.float32)
            Mn = (Mn * 10.0 ** rs.uniform(-9, 0, size=Mn.shape)).astype(np.float32)
        ex, x1, ea = _resid_norms(X)
        em, m1, eb = _resid_norms(Mn)
        if flush:
            x1 = np.where(np.abs(x1) < 2.0 ** -14, 0.0, x1)
            m1 = np.where(np.abs(m1) < 2.0 ** -14, 0.0, m1)
        approx = np.ldexp(x1 @ m1.T, -(ea + eb))
        true = X.astype(np.float64) @ Mn.astype(np.float64).T
        xn = np.linalg.norm(X.astype(np.float64), axis=1)
        Mmax = np.linalg.norm(Mn.astype(np.float64), axis=1).max()
        bound = (xn + ex) * em.max() + ex * Mmax
        assert (np.abs(approx - true) <= bound[:, None] * (1 + 1e-9)).all()
        if not wide:
            assert np.median(bound / (2.0 ** -10 * xn * Mmax)) < 0.5
